"""One kernel family under rocprofv3 (tools/pmc_profile.sh puts this program directly after `--`):

    python3 tools/profile_target.py render --net v1|v2|v3|v3w --mode bf16|f16|f16x3|f32 [--reps 5]     800x800x64 frames, ERT off
    python3 tools/profile_target.py queue  --net v1 --mode f16 [--eps 1e-2]                               ray-queue kernel, "smooth" scene
    python3 tools/profile_target.py train  --net v1|v2|v3 --mode bf16 [--rays 2048 --samples 32]         FusedStep: forward / dZ chain / dW / reduce

Prints one JSON line (HIP-event mean ms of the launches it made) so that the profiler's kernel durations can be compared."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf_few_shot_limitations_amd as N                       # noqa: E402
from oracle import nerf_oracle as O                             # noqa: E402  (deterministic synthetic weights / camera only)


def model(net, mode, scene):
    if net == "v1":
        m = N.NeRFMLP(pos_dim=63, hidden_dim=256, n_layers=8, mma_mode=mode)
        m.load_state_dict(O.make_weights("v1", 0, scene))
    elif net == "v2":
        m = N.NeRFMLP(pos_freq=10, dir_freq=4, hidden_dim=256, num_density_layers=8, use_dino=False, mma_mode=mode)
        m.load_state_dict(O.make_weights("v2", 1, scene), strict=False)
    else:
        dd = 128 if net == "v3w" else 64
        m = N.NeRFMLP(pos_freq=12, dir_freq=4, hidden_dim=256, num_density_layers=8, use_dino=True, dino_dim=dd, mma_mode=mode)
        m.load_state_dict(O.make_weights("v3", 2 if dd == 64 else 3, scene, dino_dim=dd), strict=False)
    return m.cuda()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("target", choices=["render", "queue", "train"])
    ap.add_argument("--net", default="v1")
    ap.add_argument("--mode", default="f16")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--eps", type=float, default=1e-2)
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--samples", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ev = []

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            ev.append((e0, e1))
        torch.cuda.synchronize()
        return sum(x.elapsed_time(y) for x, y in ev) / len(ev)

    if a.target in ("render", "queue"):
        H = W = 800
        S = 64
        c2w = torch.from_numpy(O.LEGO_LIKE_C2W.copy())
        m = model(a.net, a.mode, "smooth" if a.target == "queue" else "solid").eval()
        dino = None
        if a.net in ("v3", "v3w"):
            dd = 128 if a.net == "v3w" else 64
            fm = torch.from_numpy(O.uniform01(7, 28 * 28 * dd).reshape(1, 28, 28, dd) * 2 - 1)
            dino = dict(features=fm, pose=c2w, focal=O.focal_for(W), H=H, W=W)
        eps = a.eps if a.target == "queue" else 0.0
        with torch.no_grad():
            ms = timed(lambda: N.render_camera(m, H, W, O.focal_for(W), c2w, 2.0, 6.0, S, ert_eps=eps, dino=dino), a.reps)
        out = {"target": a.target, "net": a.net, "mode": a.mode, "ms_per_launch": round(ms, 4), "ray_samples": H * W * S,
               "flops_per_sample": m.flops_per_sample(), "TFLOP_per_s_credited": round(H * W * S * m.flops_per_sample() / ms / 1e9, 1), "ert_eps": eps}
    else:
        from nerf_few_shot_limitations_amd.training import FusedStep
        R, S = a.rays, a.samples
        m = model(a.net, a.mode, "fog").train()
        torch.manual_seed(0)
        pts = torch.rand(R * S, 63 if a.net == "v1" else 3, device=dev) * 2 - 1
        dirs = torch.rand(R * S, 3, device=dev) * 2 - 1
        z = torch.sort(torch.rand(R, S, device=dev) * 4 + 2, dim=-1).values.contiguous()
        d = torch.rand(R, 3, device=dev) - 0.5
        tgt = torch.rand(R, 3, device=dev)
        kw = {} if a.net == "v1" else {"dirs": dirs}
        if a.net == "v3":
            kw["dino"] = torch.rand(R * S, 64, device=dev) * 2 - 1
        step = FusedStep(m, lr=5e-4, weight_decay=1e-6)
        for _ in range(3):
            step(pts, z, d, tgt, **kw)
        ms = timed(lambda: step(pts, z, d, tgt, **kw), a.reps * 4)
        out = {"target": "train", "net": a.net, "mode": a.mode, "ms_per_step": round(ms, 4), "rays": R, "samples_per_ray": S,
               "M_ray_samples_per_s": round(R * S / ms / 1e3, 1)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
