"""The whole training run of experiments/baseline.yaml's schedule (200 epochs, 5 views of 128 x 128, progressive stages,
Adam 5e-4 with MultiStepLR, validation every 10 epochs, checkpoints) through train_cli on a synthetic Blender-format scene:
wall time of the run on one MI355X.  The images are smooth colour ramps, not a NeRF dataset -- the point is the clock.

    python tools/run_baseline_schedule.py [--mode bf16] [--epochs 200]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_scene(root, size=128, n_train=5, n_test=2):
    from PIL import Image
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / (size - 1)
    base = np.array([[-0.9999, 0.0042, -0.0133, -0.0538], [-0.0140, -0.2997, 0.9539, 3.8455], [0.0, 0.9540, 0.2997, 1.2081], [0, 0, 0, 1]], np.float32)
    for split, count in (("train", n_train), ("test", n_test)):
        os.makedirs(os.path.join(root, split))
        frames = []
        for i in range(count):
            img = np.stack([0.6 + 0.4 * xx, 0.6 + 0.4 * yy, np.full_like(xx, 0.7 + 0.05 * i), np.ones_like(xx)], -1)
            Image.fromarray((img * 255).astype(np.uint8), "RGBA").save(os.path.join(root, split, f"r_{i}.png"))
            pose = base.copy()
            pose[0, 3] += 0.05 * i
            frames.append({"file_path": f"./{split}/r_{i}", "transform_matrix": pose.tolist()})
        json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, open(os.path.join(root, f"transforms_{split}.json"), "w"))


CFG = """experiment: {name: schedule}
data: {near: 2.0, far: 6.0, resolution: 128, num_views: 5}
rendering: {near: 2.0, far: 6.0, chunk_size: 2048, white_bkgd: false}
model: {use_dino: false}
nerf_model: {pos_freq: 10, dir_freq: 4, hidden_dim: 256, num_layers: 8}
training: {epochs: 200, batch_size: 1024, progressive_schedule: {epochs_0_50: [32, 32, 32], epochs_50_100: [64, 64, 48], epochs_100_plus: [128, 128, 64]}}
optimizer: {lr: 5.0e-4, weight_decay: 1.0e-6, lr_milestones: [100, 150], lr_gamma: 0.5}
loss: {rgb_weight: 1.0, depth_weight: 0.0, reg_weight: 0.0}
output: {save_dir: unused, val_freq: 10, save_freq: 50}
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="bf16")
    ap.add_argument("--epochs", type=int, default=200)
    args = ap.parse_args()
    from nerf_few_shot_limitations_amd import train_cli
    with tempfile.TemporaryDirectory() as tmp:
        write_scene(os.path.join(tmp, "scene"))
        cfg = os.path.join(tmp, "cfg.yaml")
        open(cfg, "w").write(CFG)
        t0 = time.perf_counter()
        log = train_cli.main(["--config", cfg, "--data", os.path.join(tmp, "scene"), "--out", os.path.join(tmp, "run"), "--mode", args.mode,
                              "--epochs", str(args.epochs), "--seed", "0"])
        wall = time.perf_counter() - t0
    train_s = sum(r["seconds"] for r in log)
    samples = 0
    print(json.dumps({"epochs": len(log), "mode": args.mode, "wall_s_incl_loading_validation_checkpoints": round(wall, 2), "training_epochs_s": round(train_s, 2),
                      "first_loss": log[0]["loss"], "last_loss": log[-1]["loss"], "last_val_psnr": [r["psnr"] for r in log if "psnr" in r][-1]}))


if __name__ == "__main__":
    main()
