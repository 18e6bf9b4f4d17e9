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
from nerf_few_shot_limitations_amd.training import Adam, FusedStep, _train_handle   # noqa: E402


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
    ap.add_argument("--net", default="v1", choices=["v1", "v2", "v3"])
    ap.add_argument("--samples", type=int, nargs="+", default=[65536, 1048576])
    ap.add_argument("--rays-samples", type=int, default=32, help="samples per ray of the whole-step timing")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--cpu-samples", type=int, default=0, help="also time the CPU oracle's step on this many samples (0 = skip)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    v2 = args.net != "v1"
    v3 = args.net == "v3"
    if v3:
        model = N.NeRFMLP(pos_freq=12, dir_freq=4, hidden_dim=256, num_density_layers=8, use_dino=True, dino_dim=64, mma_mode=args.mode).to(dev).train()
    elif v2:
        model = N.NeRFMLP(pos_freq=10, dir_freq=4, hidden_dim=256, num_density_layers=8, use_dino=False, mma_mode=args.mode).to(dev).train()
    else:
        model = N.NeRFMLP(pos_dim=63, hidden_dim=256, n_layers=8, mma_mode=args.mode).to(dev).train()
    fwd_flops = model.flops_per_sample()
    first_dx = 2 * 63 * 256 + (2 * 27 * 128 if v2 else 0)          # no gradient with respect to the encoded inputs
    if v3:
        first_dx = 2 * 139 * 256 + 2 * 27 * 128                    # first fusion pass only; the second pass needs d inputs for the gate
    bwd_flops = 2 * fwd_flops - first_dx
    out_lines = []
    for n in args.samples:
        x = torch.rand(n, 3 if v2 else 63, device=dev) * 2 - 1
        dirs = torch.rand(n, 3, device=dev) * 2 - 1
        dino = torch.rand(n, 64, device=dev) * 2 - 1 if v3 else None
        g = torch.rand(n, 4, device=dev) - 0.5
        g3, g1 = g[:, :3].contiguous(), g[:, 3:].contiguous()
        h, mode = _train_handle(model, dev)
        nbytes = L.lib().nrf_train_context_bytes(h, mode, n)
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = torch.empty(n, 4, device=dev)
        rgb, den = torch.empty(n, 3, device=dev), torch.empty(n, 1, device=dev)
        grad = torch.zeros(model.flat_params().flat.numel(), device=dev)
        ctx = C.c_void_p(buf.data_ptr())

        def fwd():
            if v2:
                L.check(L.lib().nrf_mlp_forward_train(h, mode, L.ptr(x), L.ptr(dirs), L.ptr(dino), n, L.ptr(rgb), L.ptr(den), ctx, nbytes, L.stream_ptr()))
            else:
                L.check(L.lib().nrf_mlp_forward_train_v1(h, mode, L.ptr(x), n, L.ptr(out), ctx, nbytes, L.stream_ptr()))

        def bwd():
            if v2:
                L.check(L.lib().nrf_mlp_backward(h, mode, L.ptr(rgb), L.ptr(den), L.ptr(g3), L.ptr(g1), n, ctx, nbytes, L.ptr(grad), L.stream_ptr()))
            else:
                L.check(L.lib().nrf_mlp_backward_v1(h, mode, L.ptr(out), L.ptr(g), n, ctx, nbytes, L.ptr(grad), L.stream_ptr()))

        t_f = timed(fwd, args.steps)
        t_b = timed(bwd, args.steps)
        # whole step through autograd + the flat Adam kernel
        S = args.rays_samples
        R = n // S
        z = torch.sort(torch.rand(R, S, device=dev) * 4 + 2, dim=-1).values
        d = torch.rand(R, 3, device=dev) - 0.5
        tgt = torch.rand(R, 3, device=dev)
        xs, ds = x[:R * S], dirs[:R * S]
        dn = dino[:R * S] if v3 else None
        opt = Adam(model, lr=5e-4)
        vr = N.VolumeRenderer()

        def step():
            opt.zero_grad()
            if v2:      # train.py:229-236
                c, sg = model(xs, ds, dn)
                pred = vr(c.view(R, S, 3), sg.view(R, S, 1), z, d)[0]
            else:       # train_minimal.py:102-120
                pred = N.volume_render_radiance(model(xs).view(R, 1, S, 4), z.view(R, 1, S), d.view(R, 1, 3)).view(R, 3)
            torch.nn.functional.mse_loss(pred, tgt).backward()
            opt.step()

        t_s = timed(step, args.steps)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        t_wall = (time.perf_counter() - t0) * 1e3 / args.steps
        fused = FusedStep(model, lr=5e-4)
        zs = z.contiguous()
        t_fs = timed(lambda: fused(xs, zs, d, tgt, dirs=ds if v2 else None, dino=dn), args.steps)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fused(xs, zs, d, tgt, dirs=ds if v2 else None, dino=dn)
        torch.cuda.synchronize()
        t_fs_wall = (time.perf_counter() - t0) * 1e3 / args.steps
        line = {
            "net": args.net, "samples": n, "mode": args.mode, "ctx_MB": round(nbytes / 2 ** 20, 1),
            "forward_ms": round(t_f, 4), "forward_TFLOPs": round(n * fwd_flops / t_f / 1e9, 1),
            "backward_ms": round(t_b, 4), "backward_TFLOPs": round(n * bwd_flops / t_b / 1e9, 1),
            "autograd_step_ms": round(t_s, 4), "autograd_step_wall_ms": round(t_wall, 4),
            "fused_step_ms": round(t_fs, 4), "fused_step_wall_ms": round(t_fs_wall, 4),
            "train_Msamples_per_s": round(R * S / max(t_fs, t_fs_wall) / 1e3, 1),
            "fwd_bwd_frac_of_2.5PF": round(n * (fwd_flops + bwd_flops) / (t_f + t_b) / 1e9 / 2500.0, 4),
        }
        print(json.dumps(line), flush=True)
        out_lines.append(line)
    if args.cpu_samples > 0:
        out_lines.append(cpu_baseline(args))
        print(json.dumps(out_lines[-1]), flush=True)
    return out_lines


def cpu_baseline(args):
    """The same optimisation step on the host cores with the CPU oracle + torch autograd + torch.optim.Adam (the
    reference's own arithmetic, train_minimal.py:97-123 / train.py:280-288), on a bounded batch."""
    from oracle import nerf_oracle as O
    v2 = args.net != "v1"
    v3 = args.net == "v3"
    S = args.rays_samples
    R = max(args.cpu_samples // S, 1)
    threads = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(threads)
    p = O.make_weights(args.net, 0, "fog")
    pp = {k: torch.nn.Parameter(v.clone()) for k, v in p.items()}
    opt = torch.optim.Adam(list(pp.values()), lr=5e-4)
    pos = torch.rand(R * S, 3) * 4 - 2
    dirs = torch.rand(R * S, 3) * 2 - 1
    dino = torch.rand(R * S, 64) * 2 - 1
    z = torch.sort(torch.rand(R, S) * 4 + 2, dim=-1).values
    d = torch.rand(R, 3) - 0.5
    tgt = torch.rand(R, 3)
    xenc = O.positional_encoding(pos, 10)

    def step():
        opt.zero_grad()
        if v3:
            c, sg = O.mlp_v3(pp, pos, dirs, dino)
            pred = O.volume_render(c.reshape(R, S, 3), sg.reshape(R, S, 1), z, d)[0]
        elif v2:
            c, sg = O.mlp_v2(pp, pos, dirs)
            pred = O.volume_render(c.reshape(R, S, 3), sg.reshape(R, S, 1), z, d)[0]
        else:
            pred = O.volume_render_radiance(O.mlp_v1(pp, xenc).reshape(R, 1, S, 4), z.reshape(R, 1, S), d.reshape(R, 1, 3)).reshape(R, 3)
        torch.nn.functional.mse_loss(pred, tgt).backward()
        opt.step()

    step()
    reps, t0 = 0, time.perf_counter()
    while reps < 3 or time.perf_counter() - t0 < 5.0:
        step()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return {"cpu_baseline": {"value": round(R * S / dt / 1e6, 4), "unit": "M ray-samples/s per optimisation step", "cores": threads,
                             "kind": "port", "sample": f"{R} rays x {S} samples, {reps} steps, net {args.net}, fp32"}}


if __name__ == "__main__":
    main()
