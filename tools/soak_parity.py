"""One-off soak of the randomized parity sweep (tests/test_gpu_parity.py:random_render_case): N random render configurations against the
CPU oracle, worst errors per (mode, sample count) instead of a pass / fail.   python tools/soak_parity.py [--count 600] [--seed 777]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--count", type=int, default=600)
    ap.add_argument("--seed", type=int, default=777)
    a = ap.parse_args()
    import nerf_few_shot_limitations_amd as N
    import test_gpu_parity as Tp
    rng = np.random.RandomState(a.seed)
    models, worst, over, cam_bad, flat_over = {}, {}, [], 0, 0
    for it in range(a.count):
        tag, err, cam_equal = Tp.random_render_case(N, rng, models, it)
        key = (tag[2], tag[5])
        w = worst.setdefault(key, {"n": 0, "z": 0.0, "rgb": 0.0, "depth": 0.0, "weights": 0.0})
        w["n"] += 1
        for k in ("z", "rgb", "depth", "weights"):
            w[k] = max(w[k], err[k])
        tol, tol_depth = Tp.parity_tol(tag[2], tag[5])
        if max(err["rgb"], err["weights"]) > tol or err["depth"] > tol_depth or err["z"] > 2e-6:
            over.append((tag, err))
        flat_over += max(err["rgb"], err["depth"], err["weights"]) > Tp.TOL
        cam_bad += cam_equal is False
    print(f"{a.count} random configurations, seed {a.seed}: {len(over)} over parity_tol ({flat_over} over a flat 1e-4 on everything), "
          f"{cam_bad} camera-route mismatches")
    print("mode   S    n   max|z|      max|rgb|    max|depth|  max|weights|  tol (rgb, weights | depth)")
    for (mode, S) in sorted(worst):
        w = worst[(mode, S)]
        print(f"{mode:6s} {S:3d} {w['n']:4d}  {w['z']:.2e}   {w['rgb']:.2e}   {w['depth']:.2e}   {w['weights']:.2e}   {Tp.parity_tol(mode, S)[0]:.1e} | {Tp.parity_tol(mode, S)[1]:.1e}")
    for tag, err in over[:10]:
        print("OVER", tag, json.dumps(err))
    sys.exit(1 if over or cam_bad else 0)


if __name__ == "__main__":
    main()
