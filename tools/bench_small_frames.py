"""Small frames (BASELINE config 1 and the validation frames of the progressive schedule): ms per frame for every pinned
samples-per-pass split (env NRF_SPW, read per launch) and for the launcher's own choice (fused_impl.hpp:pick_spw_log2).

    python tools/bench_small_frames.py [--mode f16] [--net v2]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf_few_shot_limitations_amd as N                       # noqa: E402
from oracle import nerf_oracle as O                             # noqa: E402  (synthetic weights / camera only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="f16")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    c2w = torch.from_numpy(O.LEGO_LIKE_C2W.copy())
    m = N.NeRFMLP(pos_freq=10, dir_freq=4, hidden_dim=256, num_density_layers=8, use_dino=False, mma_mode=a.mode)
    m.load_state_dict(O.make_weights("v2", 1, "solid"), strict=False)
    m = m.cuda().eval()
    fl = m.flops_per_sample()
    with torch.no_grad():
        for _ in range(10):              # bring the chip to its working clocks first
            N.render_camera(m, 400, 400, O.focal_for(400), c2w, 2.0, 6.0, 64)
    torch.cuda.synchronize()
    for (H, S) in ((100, 32), (32, 32), (64, 48), (128, 64), (200, 64), (400, 64)):
        row = {"frame": f"{H}x{H}x{S}", "mode": a.mode}
        for pin in (None, 0, 1, 2, 3, 4, 5, 6):
            if pin is None:
                os.environ.pop("NRF_SPW", None)
            else:
                os.environ["NRF_SPW"] = str(pin)
            fn = lambda: N.render_camera(m, H, H, O.focal_for(H), c2w, 2.0, 6.0, S)
            with torch.no_grad():
                for _ in range(5 if pin is None else 2):          # the first timing of a process also pays the clock ramp
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.reps):
                    fn()
                torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.reps * 1e3
            row["auto" if pin is None else f"spw{1 << pin}"] = round(ms, 4)
        os.environ.pop("NRF_SPW", None)
        row["auto_frac_of_peak"] = round(H * H * S * fl / (row["auto"] * 1e-3) / 2.5e15, 4)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
