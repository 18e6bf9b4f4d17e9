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
