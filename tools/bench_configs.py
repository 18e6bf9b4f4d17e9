"""The five workloads BASELINE.json names, timed on ONE MI355X (synthetic weights, scene "solid", camera of SURVEY.md 8d):

  C1  100 x 100 x 32, baseline model (the reference's CPU-runnable case)
  C2  400 x 400 x 64, baseline model
  C3  800 x 800, 128 coarse + 64 fine (hierarchical: coarse pass, inverse-cdf resampling, fine pass on 192 sorted depths)
  C4  400 x 400 x 64, DINO-conditioned model (feature map 28 x 28 x 64)
  C5  800 x 800 x 128, baseline model -- the whole frame on one GPU (the 8-GPU tile-sharded run is bench.py --gpus 8)

    python tools/bench_configs.py [--mode bf16] [--reps 5]
Prints one JSON line per configuration: ms per frame, M ray-samples/s, fraction of the dense MFMA peak.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import nerf_few_shot_limitations_amd as N                       # noqa: E402
from oracle import nerf_oracle as O                             # noqa: E402  (synthetic weights / camera generators only)

PEAK = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3, "f16x3": 2500.0}


def timed(fn, reps):
    for _ in range(3):          # warm-up: the first launches of a process run at an idle chip's clocks
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="bf16", choices=["bf16", "f16", "f32", "f16x3"])
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    c2w = torch.from_numpy(O.LEGO_LIKE_C2W.copy())
    v2 = N.NeRFMLP(pos_freq=10, dir_freq=4, hidden_dim=256, num_density_layers=8, use_dino=False, mma_mode=args.mode)
    v2.load_state_dict(O.make_weights("v2", 1, "solid"), strict=False)
    v2 = v2.to(dev).eval()
    v3 = N.NeRFMLP(pos_freq=12, dir_freq=4, hidden_dim=256, num_density_layers=8, use_dino=True, dino_dim=64, mma_mode=args.mode)
    v3.load_state_dict(O.make_weights("v3", 2, "solid"), strict=False)
    v3 = v3.to(dev).eval()
    fm = torch.from_numpy(O.uniform01(7, 28 * 28 * 64).reshape(1, 28, 28, 64) * 2 - 1)

    def camera(model, H, S, dino=None):
        return lambda: N.render_camera(model, H, H, O.focal_for(H), c2w, 2.0, 6.0, S, dino=dino)

    ro8, rd8 = N.get_rays(800, 800, O.focal_for(800), c2w)
    ro8, rd8 = ro8.reshape(-1, 3), rd8.reshape(-1, 3)
    cases = [
        ("C1 100x100x32 baseline", camera(v2, 100, 32), 100 * 100 * 32, v2),
        ("C2 400x400x64 baseline", camera(v2, 400, 64), 400 * 400 * 64, v2),
        ("C3 800x800 128 coarse + 64 fine (hierarchical)", lambda: N.render_hierarchical(v2, ro8, rd8, 2.0, 6.0, 128, 64), 800 * 800 * (128 + 192), v2),
        ("C4 400x400x64 DINO-conditioned", camera(v3, 400, 64, dict(features=fm, pose=c2w, focal=O.focal_for(400), H=400, W=400)), 400 * 400 * 64, v3),
        ("C5 800x800x128 baseline, one GPU", camera(v2, 800, 128), 800 * 800 * 128, v2),
        ("C5 shard: 100 rows (80 000 rays) x 128 of the 800x800 frame = one GPU's share of 8", lambda: N.render_camera(v2, 800, 800, O.focal_for(800), c2w, 2.0, 6.0, 128, ray_begin=0, ray_end=80000), 80000 * 128, v2),
        ("headline shape 800x800x64 baseline", camera(v2, 800, 64), 800 * 800 * 64, v2),
    ]
    with torch.no_grad():
        for _ in range(10):          # bring the chip to its working clocks: a cold first case reads 15-20 % slow
            camera(v2, 400, 64)()
        torch.cuda.synchronize()
        for name, fn, samples, model in cases:
            dt = timed(fn, args.reps)
            tflops = samples * model.flops_per_sample() / dt / 1e12
            print(json.dumps({"config": name, "mode": args.mode, "ms_per_frame": round(dt * 1e3, 3), "M_ray_samples_per_s": round(samples / dt / 1e6, 1),
                              "ray_samples": samples, "TFLOP_per_s": round(tflops, 1), "frac_of_mfma_peak": round(tflops / PEAK[args.mode], 4)}), flush=True)


if __name__ == "__main__":
    main()
