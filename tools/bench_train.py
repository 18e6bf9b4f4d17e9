"""Timing of the training path on one MI355X.

    python tools/bench_train.py [--mode bf16] [--samples 65536 1048576] [--steps 20]

Prints, per batch size, the time of the saving forward, of the backward (dZ chain + weight gradients) and of a
whole optimisation step of the reference's loop shape (train_minimal.py:97-123: encode -> NeRFMLP -> composite ->
mse -> backward -> Adam) through the drop-in surface.  FLOP accounting: forward = flops_per_sample, backward =
2 x forward minus the first layer's dX (not needed), all as dense MAC counts of the Linear layers.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import nerf_few_shot_limitations_amd as N                     # noqa: E402
from nerf_few_shot_limitations_amd import _lib as L           # noqa: E402
from nerf_few_shot_limitations_amd.training import Adam, _train_handle   # noqa: E402


def timed(fn, steps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--samples", type=int, nargs="+", default=[65536, 1048576])
    ap.add_argument("--rays-samples", type=int, default=32, help="samples per ray of the whole-step timing")
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = N.NeRFMLP(pos_dim=63, hidden_dim=256, n_layers=8, mma_mode=args.mode).to(dev).train()
    fwd_flops = model.flops_per_sample()
    first_dx = 2 * 63 * 256
    bwd_flops = 2 * fwd_flops - first_dx
    out_lines = []
    for n in args.samples:
        x = torch.rand(n, 63, device=dev) * 2 - 1
        g = torch.rand(n, 4, device=dev) - 0.5
        h, mode = _train_handle(model, dev)
        nbytes = L.lib().nrf_train_context_bytes(h, mode, n)
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = torch.empty(n, 4, device=dev)
        grad = torch.zeros(model.flat_params().flat.numel(), device=dev)

        def fwd():
            L.check(L.lib().nrf_mlp_forward_train_v1(h, mode, L.ptr(x), n, L.ptr(out), C.c_void_p(buf.data_ptr()), nbytes, L.stream_ptr()))

        def bwd():
            L.check(L.lib().nrf_mlp_backward_v1(h, mode, L.ptr(out), L.ptr(g), n, C.c_void_p(buf.data_ptr()), nbytes, L.ptr(grad), L.stream_ptr()))

        t_f = timed(fwd, args.steps)
        t_b = timed(bwd, args.steps)
        # whole step through autograd + the flat Adam kernel
        S = args.rays_samples
        R = n // S
        z = torch.sort(torch.rand(R, S, device=dev) * 4 + 2, dim=-1).values
        d = torch.rand(R, 3, device=dev) - 0.5
        tgt = torch.rand(R, 3, device=dev)
        xs = x[:R * S]
        opt = Adam(model, lr=5e-4)

        def step():
            opt.zero_grad()
            pred = N.volume_render_radiance(model(xs).view(R, 1, S, 4), z.view(R, 1, S), d.view(R, 1, 3))
            torch.nn.functional.mse_loss(pred.view(R, 3), tgt).backward()
            opt.step()

        t_s = timed(step, args.steps)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        t_wall = (time.perf_counter() - t0) * 1e3 / args.steps
        line = {
            "samples": n, "mode": args.mode, "ctx_MB": round(nbytes / 2 ** 20, 1),
            "forward_ms": round(t_f, 4), "forward_TFLOPs": round(n * fwd_flops / t_f / 1e9, 1),
            "backward_ms": round(t_b, 4), "backward_TFLOPs": round(n * bwd_flops / t_b / 1e9, 1),
            "step_ms": round(t_s, 4), "step_wall_ms": round(t_wall, 4),
            "train_Msamples_per_s": round(R * S / t_s / 1e3, 1),
            "fwd_bwd_frac_of_2.5PF": round(n * (fwd_flops + bwd_flops) / (t_f + t_b) / 1e9 / 2500.0, 4),
        }
        print(json.dumps(line), flush=True)
        out_lines.append(line)
    return out_lines


if __name__ == "__main__":
    main()
