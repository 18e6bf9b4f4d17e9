"""Wall time of one training epoch of the reference's progressive schedule (experiments/baseline.yaml values: 5 views of
128 x 128, batch 1024, stages [32,32,32] / [64,64,48] / [128,128,64]) through train_cli.train_epoch on a synthetic scene:
everything the loop does per ray batch -- shuffle, gather, stratified sampling, FusedStep -- on one MI355X.

    python tools/bench_epoch.py [--mode bf16] [--net v2|v3]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import nerf_few_shot_limitations_amd as N                              # noqa: E402
from nerf_few_shot_limitations_amd import train_cli                    # noqa: E402
from nerf_few_shot_limitations_amd.training import FusedStep           # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="bf16")
    ap.add_argument("--net", default="v2", choices=["v2", "v3"])
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cfg = {"training": {"batch_size": 1024, "progressive_schedule": {"epochs_0_50": [32, 32, 32], "epochs_50_100": [64, 64, 48],
                                                                      "epochs_100_plus": [128, 128, 64]}},
           "optimizer": {"lr": 5e-4, "weight_decay": 1e-6, "lr_milestones": [100, 150], "lr_gamma": 0.5}}
    H = W = 128
    V = 5
    focal = 0.5 * W / 0.36
    images = [torch.rand(H, W, 3, device=dev) for _ in range(V)]
    poses = []
    for v in range(V):
        p = torch.eye(4)
        p[:3, 3] = torch.tensor([0.1 * v, 0.0, 4.0])
        poses.append(p)
    v3 = args.net == "v3"
    model = N.NeRFMLP(pos_freq=12 if v3 else 10, dir_freq=4, hidden_dim=256, num_density_layers=8, use_dino=v3, dino_dim=64 if v3 else 0,
                      mma_mode=args.mode).to(dev).train()
    maps = torch.rand(V, 9, 9, 64, device=dev) * 2 - 1 if v3 else None
    step = FusedStep(model, lr=5e-4, weight_decay=1e-6)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    for epoch in (0, 50, 100):
        train_cli.train_epoch(step, cfg, epoch, images, poses, H, W, focal, 2.0, 6.0, gen, maps)      # warm-up (buffers, kernels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            loss, samples = train_cli.train_epoch(step, cfg, epoch, images, poses, H, W, focal, 2.0, 6.0, gen, maps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        Ht, Wt, S, batch = train_cli.schedule_for(cfg, epoch)
        print(json.dumps({"net": args.net, "mode": args.mode, "epoch": epoch, "stage": [Ht, Wt, S], "batch_rays": batch, "ray_samples_per_epoch": samples,
                          "epoch_ms": round(dt * 1e3, 2), "Msamples_per_s": round(samples / dt / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
