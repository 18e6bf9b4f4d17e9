"""Where a step of the reference's unmodified train_step body (train.py:280-287) spends its host time on the drop-in surface:
segment timers (host time without synchronisation = what the Python costs; with synchronisation = what the GPU costs) and a
cProfile of 200 steps.  python tools/profile_autograd_step.py [--net v1|v2|v3] [--mode bf16] [--fused-adam]"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf_few_shot_limitations_amd as N                                      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="v1")
    ap.add_argument("--mode", default="bf16")
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--samples", type=int, default=32)
    ap.add_argument("--fused-adam", action="store_true")
    ap.add_argument("--nrf-adam", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    if a.net == "v1":
        m = N.NeRFMLP(pos_dim=63, hidden_dim=256, n_layers=8, mma_mode=a.mode)
    else:
        m = N.NeRFMLP(pos_freq=12 if a.net == "v3" else 10, dir_freq=4, hidden_dim=256, num_density_layers=8, use_dino=a.net == "v3", dino_dim=64,
                      mma_mode=a.mode)
    m = m.to(dev).train()
    R, S = a.rays, a.samples
    rend = N.NeRFRenderer(m, 2.0, 6.0, dino_features=[torch.rand(1, 9, 9, 64, device=dev) * 2 - 1] if a.net == "v3" else None,
                          poses=[torch.eye(4)], focal=100.0, H=128, W=128)
    ro = torch.rand(R, 3, device=dev) - 0.5 + torch.tensor([0.0, 0.0, 4.0], device=dev)
    d = torch.rand(R, 3, device=dev) - 0.5
    tgt = torch.rand(R, 3, device=dev)
    if a.nrf_adam:
        opt = N.training.Adam(m, lr=5e-4, weight_decay=1e-6)
    else:
        opt = torch.optim.Adam(m.parameters(), lr=5e-4, weight_decay=1e-6, fused=True if a.fused_adam else None)
    seg = {k: 0.0 for k in ("render", "loss", "zero_grad", "backward", "opt.step")}

    def body(sync=False, timed=False):
        t = [time.perf_counter()]

        def mark():
            if sync:
                torch.cuda.synchronize()
            t.append(time.perf_counter())
        pred = rend.render_rays(ro, d, 0, S); mark()
        ls = torch.nn.functional.mse_loss(pred["rgb"], tgt); mark()
        opt.zero_grad(); mark()
        ls.backward(); mark()
        opt.step(); mark()
        if timed:
            for k, (x, y) in zip(seg, zip(t[:-1], t[1:])):
                seg[k] += y - x
    for _ in range(5):
        body()
    torch.cuda.synchronize()
    k = 200
    t0 = time.perf_counter()
    for _ in range(k):
        body()
    torch.cuda.synchronize()
    print(f"{a.net} {a.mode} {R}x{S}: {(time.perf_counter() - t0) / k * 1e3:.4f} ms per step (free running)")
    for sync in (False, True):
        for key in seg:
            seg[key] = 0.0
        torch.cuda.synchronize()
        for _ in range(k):
            body(sync=sync, timed=True)
        torch.cuda.synchronize()
        print(("synchronised after each segment (GPU + host): " if sync else "host time per segment (no synchronisation):   ")
              + "  ".join(f"{key} {v / k * 1e3:.3f}" for key, v in seg.items()) + f"  | sum {sum(seg.values()) / k * 1e3:.3f} ms")
    if a.no_profile:
        return
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(k):
        body()
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print(s.getvalue()[:6000])


if __name__ == "__main__":
    main()
