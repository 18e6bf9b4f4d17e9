"""Prints the measured error of every MFMA mode against the golden vectors / the oracle
(run on the GPU box; the numbers quoted in DESIGN.md and the bounds in test_gpu_parity.py
come from this report)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf_few_shot_limitations_amd as N
from oracle import nerf_oracle as O

def T(a): return torch.from_numpy(np.asarray(a))
def g(name):
    with np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz")) as z:
        return {k: z[k] for k in z.files}
def md(a, b): return float((a.detach().cpu().double() - T(b).double()).abs().max())

def mk(variant, scene, mode):
    p = O.make_weights(variant, {"v1": 0, "v2": 1}[variant], scene)
    if variant == "v1":
        m = N.NeRFMLP(pos_dim=63, mma_mode=mode); m.load_state_dict(p)
    else:
        m = N.NeRFMLP(pos_freq=10, dir_freq=4, hidden_dim=256, num_density_layers=8, use_dino=False, mma_mode=mode); m.load_state_dict(p, strict=False)
    return m.cuda().eval(), p

e = g("end_to_end")
H, W, S = int(e["H"]), int(e["W"]), int(e["S"])
ro, rd = N.get_rays(H, W, float(e["focal"]), T(e["c2w"]))
for variant in ("v1", "v2"):
    for scene in ("fog", "solid"):
        for mode in ("f32", "f16x3", "f16", "bf16"):
            m, p = mk(variant, scene, mode)
            out = N.render_rays(m, ro, rd, 2.0, 6.0, S)
            print(f"e2e {variant} {scene} {mode}: rgb {md(out['rgb'], e[f'{variant}_{scene}_plain_rgb']):.2e} "
                  f"depth {md(out['depth'], e[f'{variant}_{scene}_plain_depth']):.2e} w {md(out['weights'], e[f'{variant}_{scene}_plain_w']):.2e}")
for scene in ("fog", "solid"):
    gm = g(f"mlp_v1_{scene}")
    for mode in ("f32", "f16x3", "f16", "bf16"):
        m, _ = mk("v1", scene, mode)
        with torch.no_grad():
            out = m(T(gm["x_enc"]))
        print(f"mlp_v1 {scene} {mode}: rgb {md(out[:, :3], gm['out'][:, :3]):.2e} sigma {md(out[:, 3], gm['out'][:, 3]):.2e} (|sigma|max {np.abs(gm['out'][:,3]).max():.1f})")
gm = g("mlp_v2")
for mode in ("f32", "f16x3", "f16", "bf16"):
    m, _ = mk("v2", "fog", mode)
    with torch.no_grad():
        rgb, dens = m(T(gm["pos"]), T(gm["dirs"]), None)
    print(f"mlp_v2 {mode}: rgb {md(rgb, gm['rgb']):.2e} density {md(dens, gm['density']):.2e} (max {gm['density'].max():.1f})")
Hc = Wc = 100; Sc = 32
c2w = T(O.LEGO_LIKE_C2W)
for scene in ("fog", "solid"):
    p = O.make_weights("v1", 0, scene)
    roo, rdo = O.get_rays(Hc, Wc, O.focal_for(Wc), c2w)
    ref = O.render_rays(p, "v1", roo, rdo, 2.0, 6.0, Sc)
    for mode in ("f32", "f16x3", "f16", "bf16"):
        m, _ = mk("v1", scene, mode)
        rgb, depth = N.render_camera(m, Hc, Wc, O.focal_for(Wc), c2w, 2.0, 6.0, Sc)
        print(f"100x100x32 v1 {scene} {mode}: rgb {md(rgb, ref['rgb'].numpy()):.2e} depth {md(depth, ref['depth'].numpy()):.2e} "
              f"psnr_vs_oracle {O.psnr(rgb.cpu(), ref['rgb']):.1f} dB")

# ---- numbers behind the 16-bit bounds of tests/test_gpu_parity.py (stable rays = |sigma_last| > 0.5 in the oracle: the
#      reference's tail rule dists[-1]=1e10 makes alpha_last a step function of sigma_last, nerf_mlp.py:182)
p = O.make_weights("v1", 0, "solid")
roo, rdo = O.get_rays(Hc, Wc, O.focal_for(Wc), c2w)
ref = O.render_rays(p, "v1", roo, rdo, 2.0, 6.0, Sc)
gt = O.render_rays(p, "v1", roo, rdo, 2.0, 6.0, 2 * Sc)["rgb"]
sig_last = O.mlp_v1(p, O.positional_encoding(roo.reshape(-1, 3) + rdo.reshape(-1, 3) * 6.0, 10))[:, 3]
stable = sig_last.abs() > 0.5
ps_ref = O.psnr(ref["rgb"], gt)
for mode in ("f16x3", "f16", "bf16"):
    m, _ = mk("v1", "solid", mode)
    rgb, depth = N.render_camera(m, Hc, Wc, O.focal_for(Wc), c2w, 2.0, 6.0, Sc)
    err = (rgb.cpu() - ref["rgb"]).abs().max(-1).values
    derr = (depth.cpu() - ref["depth"]).abs()
    print(f"stable-ray bounds 100x100x32 solid {mode}: rgb max {float(err[stable].max()):.3e} median {float(err.median()):.3e} "
          f"depth max {float(derr[stable].max()):.3e} psnr {O.psnr(rgb.cpu(), ref['rgb']):.2f} dB  psnr-delta vs 2S ground truth {abs(O.psnr(rgb.cpu(), gt) - ps_ref):.5f} dB")
H8 = W8 = 800
b0, b1 = 400 * W8, 402 * W8
ro8, rd8 = O.get_rays(H8, W8, O.focal_for(W8), c2w)
ref8 = O.render_rays(p, "v1", ro8.reshape(-1, 3)[b0:b1], rd8.reshape(-1, 3)[b0:b1], 2.0, 6.0, 64)
for mode in ("f16", "bf16"):
    m, _ = mk("v1", "solid", mode)
    rgb, _ = N.render_camera(m, H8, W8, O.focal_for(W8), c2w, 2.0, 6.0, 64, ray_begin=b0, ray_end=b1)
    print(f"800x800x64 band solid {mode}: psnr vs oracle {O.psnr(rgb.cpu(), ref8['rgb']):.2f} dB")
e = g("end_to_end")
p3 = O.make_weights("v3", 2, "fog")
m3 = N.NeRFMLP(pos_freq=12, dir_freq=4, hidden_dim=256, num_density_layers=8, use_dino=True, dino_dim=64, mma_mode="bf16")
m3.load_state_dict(p3, strict=False); m3 = m3.cuda().eval()
fm = torch.from_numpy(O.uniform01(7, 28 * 28 * 64).reshape(1, 28, 28, 64) * 2 - 1)
dino = dict(features=fm, pose=T(e["c2w"]), focal=float(e["focal"]), H=H, W=W)
for mode in ("f16", "bf16"):
    out = N.render_rays(m3, ro, rd, 2.0, 6.0, S, dino=dino, mma_mode=mode)
    print(f"e2e v3 fog {mode}: rgb {md(out['rgb'], e['v3_fog_plain_rgb']):.2e} depth {md(out['depth'], e['v3_fog_plain_depth']):.2e} psnr {O.psnr(out['rgb'].cpu(), T(e['v3_fog_plain_rgb'])):.1f} dB")
