#!/usr/bin/env python3
"""Capture golden vectors by running the REFERENCE's own leaf modules.

Run in the build container only (needs /root/reference; the GPU box has no
reference):      PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
Writes tests/golden/*.npz -- inputs and the reference's outputs, nothing else.

Inputs come from the deterministic generators in oracle/nerf_oracle.py (camera,
weights, uniform01), so the fixtures are reproducible.  The reference draws its
stratified jitter with torch.rand inside sample_points_along_rays
(src/utils/ray_utils.py:78); to make the jitter an input, torch.rand is
replaced for the duration of that one call by a function returning our t_rand.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("NERF_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, os.path.join(REF, "src", "models"))

from oracle import nerf_oracle as O  # noqa: E402  (only its input generators are used here)

from models.ray_sampler import get_rays as ref_get_rays_img, sample_points_along_rays as ref_sample_img  # noqa: E402
from utils.ray_utils import get_rays as ref_get_rays_flat, sample_points_along_rays as ref_sample_flat  # noqa: E402
from utils.ray_utils import project_points_to_image as ref_project  # noqa: E402
from models.positional_encoding import PositionalEncoding as RefPE  # noqa: E402
from models.nerf_model import NeRFMLP as RefNeRFMLP  # noqa: E402
import models.nerf_mlp as ref_mlp  # noqa: E402
from models.volume_renderer import volume_render_radiance as ref_vrr  # noqa: E402
import models.dino_feature_model as ref_dfm  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(4)


def npf(t):
    return np.ascontiguousarray(t.detach().cpu().numpy())


class _patched_rand:
    def __init__(self, value):
        self.value = value

    def __enter__(self):
        self._orig = torch.rand
        v = self.value
        torch.rand = lambda *a, **k: v.clone()

    def __exit__(self, *exc):
        torch.rand = self._orig


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def cam(H, W):
    return H, W, O.focal_for(W), torch.from_numpy(O.LEGO_LIKE_C2W.copy())


def load_ref_v1(p):
    m = RefNeRFMLP(pos_dim=63, hidden_dim=256, n_layers=8)
    m.load_state_dict(p)
    return m.eval()


def load_ref_v2(p):
    dm = ref_mlp.DensityMLP(63, 256, 8)
    cm = ref_mlp.ColorMLP(256, 27, 128)
    dm.load_state_dict({k[len("density_mlp."):]: v for k, v in p.items() if k.startswith("density_mlp.")})
    cm.load_state_dict({k[len("color_mlp."):]: v for k, v in p.items() if k.startswith("color_mlp.")})
    return dm.eval(), cm.eval()


def load_ref_v3(p, dino_dim=64):
    m = ref_mlp.NeRFWithDINO(pos_freq=12, dir_freq=4, dino_dim=dino_dim, hidden_dim=256, num_density_layers=8)
    sd = m.state_dict()
    for k in sd:
        if k in p:
            sd[k] = p[k]
    m.load_state_dict(sd)
    return m.eval()


def main():
    # ---------------- K1..K6 known answers, produced by the reference ------------
    ro, rd = ref_get_rays_img(2, 3, 2.0, torch.eye(4))
    k1 = dict(rays_o=npf(ro), rays_d=npf(rd))
    pts, z = ref_sample_flat(torch.zeros(1, 3), torch.tensor([[0., 0., -1.]]), 2.0, 6.0, 5, perturb=False)
    k2 = dict(z=npf(z), pts=npf(pts))
    k3 = dict(enc=npf(RefPE(2)(torch.tensor([.5, -1., 2.]))))
    vr = ref_mlp.VolumeRenderer().eval()
    zz = torch.tensor([[2., 4., 6.]])
    I3 = torch.eye(3)[None]
    outs = {}
    for tag, sig, d, wb in [("k4", [.5, 1., 0.], [0., 0., -1.], False), ("k4w", [.5, 1., 0.], [0., 0., -1.], True),
                            ("k5", [.5, 1., 1e-3], [0., 0., -1.], False), ("k6", [.5, 1., 0.], [0., 0., -2.], False)]:
        c, dep, w = vr(I3, torch.tensor(sig)[None, :, None], zz, torch.tensor([d]), white_bkgd=wb)
        outs[tag + "_rgb"], outs[tag + "_depth"], outs[tag + "_w"] = npf(c), npf(dep), npf(w)
    save("kat", **{f"k1_{k}": v for k, v in k1.items()}, **{f"k2_{k}": v for k, v in k2.items()},
         **{f"k3_{k}": v for k, v in k3.items()}, **outs)

    # ---------------- a1 rays: both reference variants, odd sizes -----------------
    H, W, f, c2w = cam(5, 7)
    ro_i, rd_i = ref_get_rays_img(H, W, f, c2w)
    ro_f, rd_f = ref_get_rays_flat(H, W, f, c2w)
    assert torch.equal(rd_i, rd_f) and torch.equal(ro_i, ro_f)
    H2, W2, f2, _ = cam(16, 12)
    ro2, rd2 = ref_get_rays_img(H2, W2, f2, c2w[:3, :4])          # 3x4 pose
    save("rays", H=H, W=W, focal=np.float32(f), c2w=npf(c2w), rays_o=npf(ro_i), rays_d=npf(rd_i),
         H2=H2, W2=W2, focal2=np.float32(f2), rays_o2=npf(ro2), rays_d2=npf(rd2))

    # ---------------- a2 samples: flat + image layout, jitter, lindisp ------------
    S = 24
    o_flat, d_flat = ro_i.reshape(-1, 3).contiguous(), rd_i.reshape(-1, 3).contiguous()
    R = o_flat.shape[0]
    tr = torch.from_numpy(O.uniform01(11, R * S).reshape(R, S))
    pts0, z0 = ref_sample_flat(o_flat, d_flat, 2.0, 6.0, S, perturb=False)
    with _patched_rand(tr):
        pts1, z1 = ref_sample_flat(o_flat, d_flat, 2.0, 6.0, S, perturb=True)
    pts2, z2 = ref_sample_flat(o_flat, d_flat, 2.0, 6.0, S, perturb=False, lindisp=True)
    with _patched_rand(tr.reshape(H, W, S)):
        pts3, z3 = ref_sample_img(ro_i, rd_i, 2.0, 6.0, S, perturb=True)
    assert torch.equal(z3.reshape(R, S), z1)
    zs = {f"z_S{s}": npf(ref_sample_flat(o_flat[:1], d_flat[:1], 2.0, 6.0, s, perturb=False)[1][0])
          for s in (2, 3, 32, 64, 128, 192)}
    save("samples", rays_o=npf(o_flat), rays_d=npf(d_flat), near=np.float32(2), far=np.float32(6), S=S,
         t_rand=npf(tr), pts_plain=npf(pts0), z_plain=npf(z0), pts_jit=npf(pts1), z_jit=npf(z1),
         pts_lindisp=npf(pts2), z_lindisp=npf(z2), pts_img_jit=npf(pts3), **zs)

    # ---------------- a4 encoding ------------------------------------------------
    x = torch.from_numpy((O.uniform01(21, 300 * 3).reshape(300, 3) * 2 - 1) * 6.0)   # |x| up to 6 (far bound)
    enc = {f"enc_L{L}": npf(RefPE(L)(x)) for L in (4, 10, 12)}
    enc_b = npf(ref_mlp.PositionalEncoding(10)(x))
    assert np.array_equal(enc_b, enc["enc_L10"])
    save("encoding", x=npf(x), **enc)
    # log_sampling=False (positional_encoding.py:17-18; no caller of the reference uses it), with and without the raw input
    save("encoding_linear", x=npf(x), enc_L6=npf(RefPE(6, log_sampling=False)(x)),
         enc_L10_noinput=npf(RefPE(10, include_input=False, log_sampling=False)(x)))

    # ---------------- a5 V1 MLP ----------------------------------------------------
    P = 384
    for scene in ("fog", "solid"):
        p1 = O.make_weights("v1", seed=0, scene=scene)
        xe = RefPE(10)(torch.from_numpy((O.uniform01(31, P * 3).reshape(P, 3) * 2 - 1) * 4.0))
        out = load_ref_v1(p1)(xe)
        save(f"mlp_v1_{scene}", x_enc=npf(xe), out=npf(out))

    # ---------------- a6 V2 (DensityMLP + ColorMLP) --------------------------------
    p2 = O.make_weights("v2", seed=1)
    pos = torch.from_numpy((O.uniform01(41, P * 3).reshape(P, 3) * 2 - 1) * 4.0)
    dirs = torch.from_numpy((O.uniform01(42, P * 3).reshape(P, 3) * 2 - 1))
    dm, cm = load_ref_v2(p2)
    pe10, pe4 = ref_mlp.PositionalEncoding(10), ref_mlp.PositionalEncoding(4)
    dens, feat = dm(pe10(pos))
    rgb = cm(feat, pe4(dirs))
    save("mlp_v2", pos=npf(pos), dirs=npf(dirs), rgb=npf(rgb), density=npf(dens), feature=npf(feat))

    # ---------------- a7 V3 (NeRFWithDINO) -----------------------------------------
    p3 = O.make_weights("v3", seed=2)
    dino = torch.from_numpy(O.uniform01(43, P * 64).reshape(P, 64) * 2 - 1)
    m3 = load_ref_v3(p3)
    rgb3, dens3 = m3(pos, dirs, dino)
    fused = m3.dino_fusion(m3.pos_encoder(pos), dino)
    save("mlp_v3", pos=npf(pos), dirs=npf(dirs), dino=npf(dino), rgb=npf(rgb3), density=npf(dens3), fused=npf(fused))
    # multi-scale feature width (multi_scale_dino.py:50 output_dim = 128, experiments/multiscale.yaml)
    p3w = O.make_weights("v3", seed=3, dino_dim=128)
    dino_w = torch.from_numpy(O.uniform01(44, 128 * 128).reshape(128, 128) * 2 - 1)
    rgbw, densw = load_ref_v3(p3w, 128)(pos[:128], dirs[:128], dino_w)
    save("mlp_v3_d128", pos=npf(pos[:128]), dirs=npf(dirs[:128]), dino=npf(dino_w), rgb=npf(rgbw), density=npf(densw))

    # ---------------- a8 projection + bilinear fetch --------------------------------
    Hc = Wc = 400
    fm = torch.from_numpy(O.uniform01(7, 28 * 28 * 64).reshape(1, 28, 28, 64) * 2 - 1)
    pose = torch.from_numpy(O.LEGO_LIKE_C2W.copy())
    # points spread in front of / around the camera so taps fall inside, on the border and outside the map
    _, rd_c = ref_get_rays_flat(20, 20, O.focal_for(20), pose)
    pts_c = (pose[:3, 3] + rd_c.reshape(-1, 3) * torch.from_numpy(O.uniform01(8, 400) * 5 + 1.5)[:, None])
    extra = torch.from_numpy((O.uniform01(9, 112 * 3).reshape(112, 3) * 2 - 1) * 4.0)
    pts_c = torch.cat([pts_c, extra], 0)
    xy, dep, mask = ref_project(pts_c, pose, O.focal_for(Wc), Hc, Wc)
    sampled = ref_dfm.SpatialDINOFeatures.sample_features_at_points(types.SimpleNamespace(), fm, xy)
    # D8: the reference's projection puts visible points at Z<0 -> x,y mirrored; also exercise in-range taps directly
    xy_in = torch.from_numpy(O.uniform01(10, 256 * 2).reshape(256, 2) * 2.4 - 1.2)
    sampled_in = ref_dfm.SpatialDINOFeatures.sample_features_at_points(types.SimpleNamespace(), fm, xy_in)
    save("dino_fetch", features=npf(fm), pose=npf(pose), focal=np.float32(O.focal_for(Wc)), H=Hc, W=Wc,
         points=npf(pts_c), xy=npf(xy), depth=npf(dep), mask=npf(mask), sampled=npf(sampled),
         xy_in=npf(xy_in), sampled_in=npf(sampled_in))

    # ---------------- a9 / a10 compositor ------------------------------------------
    Rr, Ss = 96, 32
    rgb_in = torch.from_numpy(O.uniform01(51, Rr * Ss * 3).reshape(Rr, Ss, 3))
    sig_in = torch.from_numpy((O.uniform01(52, Rr * Ss).reshape(Rr, Ss, 1) * 2 - 0.6) * 3.0)   # some negative
    sig_in[::7] *= 30.0                                                                              # saturating rays
    tr2 = torch.from_numpy(O.uniform01(53, Rr * Ss).reshape(Rr, Ss))
    dd = torch.from_numpy((O.uniform01(54, Rr * 3).reshape(Rr, 3) * 2 - 1))
    with _patched_rand(tr2):
        _, zc = ref_sample_flat(torch.zeros(Rr, 3), dd, 2.0, 6.0, Ss, perturb=True)
    c0, d0, w0 = vr(rgb_in, sig_in, zc, dd)
    c1, d1, w1 = vr(rgb_in, sig_in, zc, dd, white_bkgd=True)
    img = ref_vrr(torch.cat([rgb_in, sig_in], -1).reshape(8, 12, Ss, 4).clone(), zc.reshape(8, 12, Ss), dd.reshape(8, 12, 3))
    save("composite", rgb_in=npf(rgb_in), sigma_in=npf(sig_in), z=npf(zc), rays_d=npf(dd),
         rgb=npf(c0), depth=npf(d0), weights=npf(w0), rgb_white=npf(c1), radiance=npf(img))

    # ---------------- a11 end to end: camera -> image, three variants ---------------
    He, We, Se = 12, 16, 32
    _, _, fe, c2we = cam(He, We)
    roe, rde = ref_get_rays_flat(He, We, fe, c2we)
    roe, rde = roe.reshape(-1, 3).contiguous(), rde.reshape(-1, 3).contiguous()
    Re = roe.shape[0]
    tre = torch.from_numpy(O.uniform01(61, Re * Se).reshape(Re, Se))
    e2e = dict(H=He, W=We, S=Se, focal=np.float32(fe), c2w=npf(c2we), t_rand=npf(tre))
    for jit in (False, True):
        tag = "jit" if jit else "plain"
        if jit:
            with _patched_rand(tre):
                pts_e, z_e = ref_sample_flat(roe, rde, 2.0, 6.0, Se, perturb=True)
        else:
            pts_e, z_e = ref_sample_flat(roe, rde, 2.0, 6.0, Se, perturb=False)
        pf = pts_e.reshape(-1, 3)
        df = rde.unsqueeze(1).expand(-1, Se, -1).reshape(-1, 3)
        for scene in ("fog", "solid"):
            # V1
            out = load_ref_v1(O.make_weights("v1", 0, scene))(RefPE(10)(pf))
            c, dpt, w = vr(out[:, :3].reshape(Re, Se, 3), out[:, 3:4].reshape(Re, Se, 1), z_e, rde)
            e2e[f"v1_{scene}_{tag}_rgb"], e2e[f"v1_{scene}_{tag}_depth"], e2e[f"v1_{scene}_{tag}_w"] = npf(c), npf(dpt), npf(w)
            # V2
            dm, cm = load_ref_v2(O.make_weights("v2", 1, scene))
            dn, ft = dm(pe10(pf))
            col = cm(ft, pe4(df))
            c, dpt, w = vr(col.reshape(Re, Se, 3), dn.reshape(Re, Se, 1), z_e, rde)
            e2e[f"v2_{scene}_{tag}_rgb"], e2e[f"v2_{scene}_{tag}_depth"], e2e[f"v2_{scene}_{tag}_w"] = npf(c), npf(dpt), npf(w)
        # V3 (fog only): project into the source view (same pose), fetch, fuse  (train.py:203-229)
        m3 = load_ref_v3(O.make_weights("v3", 2))
        xy3, _, _ = ref_project(pf, c2we, fe, He, We)
        ft3 = ref_dfm.SpatialDINOFeatures.sample_features_at_points(types.SimpleNamespace(), fm, xy3)
        col3, dn3 = m3(pf, df, ft3)
        c, dpt, w = vr(col3.reshape(Re, Se, 3), dn3.reshape(Re, Se, 1), z_e, rde)
        e2e[f"v3_fog_{tag}_rgb"], e2e[f"v3_fog_{tag}_depth"], e2e[f"v3_fog_{tag}_w"] = npf(c), npf(dpt), npf(w)
    save("end_to_end", **e2e)


def select_rays(margin_per_sample, R, S, need, margin=1e-4):
    """Indices of the first `need` rays all of whose samples keep every ReLU pre-activation away from 0 (oracle.relu_margin):
    for the others the ReLU mask -- hence the gradient -- is decided by summation order, in the reference as anywhere."""
    ok = (margin_per_sample.reshape(R, S) > margin).all(dim=1).nonzero().flatten()
    assert ok.numel() >= need, (ok.numel(), need)
    return ok[:need]


def thin(t):
    """Large gradient matrices are stored as every 4th row (the fixture stays small; tests slice the same way)."""
    return t[::4] if t.ndim == 2 and t.numel() > 20000 else t


def training():
    """One loss.backward() of the reference's own modules (src/training/train_minimal.py:97-122 for V1; train.py:229-287
    with nerf_mlp.NeRFLoss for V2): inputs, loss and every parameter gradient."""
    out = {}
    S, cand, R = 8, 400, 96
    c2w = torch.from_numpy(O.LEGO_LIKE_C2W.copy())
    ro, rd = ref_get_rays_flat(20, 20, O.focal_for(20), c2w)
    ro, rd = ro.reshape(-1, 3)[:cand], rd.reshape(-1, 3)[:cand]
    tr = torch.from_numpy(O.uniform01(301, cand * S).reshape(cand, S)).float()
    with _patched_rand(tr):
        pts, z = ref_sample_flat(ro, rd, 2.0, 6.0, S, perturb=True)
    dirs = rd.unsqueeze(1).expand(-1, S, -1)
    tgt_all = torch.from_numpy(O.uniform01(302, cand * 3).reshape(cand, 3)).float()
    # ---- V1: NeRFMLP(pos_dim=63, n_layers=3) -> volume_render_radiance -> mse
    p1 = O.make_weights("v1", 0, "solid", n_layers=3)
    keep = select_rays(O.relu_margin(p1, "v1", O.positional_encoding(pts.reshape(-1, 3), 10)), cand, S, R)
    m1 = RefNeRFMLP(pos_dim=63, hidden_dim=256, n_layers=3)
    m1.load_state_dict(p1)
    with torch.enable_grad():
        x = RefPE(10)(pts[keep].reshape(-1, 3))
        pred = ref_vrr(m1(x).view(R, 1, S, 4), z[keep].view(R, 1, S), rd[keep].view(R, 1, 3)).view(R, 3)
        loss = torch.nn.functional.mse_loss(pred, tgt_all[keep])
        loss.backward()
    out.update(v1_pts=npf(pts[keep]), v1_z=npf(z[keep]), v1_rays_d=npf(rd[keep]), v1_target=npf(tgt_all[keep]), v1_pred=npf(pred),
               v1_loss=np.float32(loss.item()))
    for k, q in m1.named_parameters():
        out["v1_grad_" + k] = npf(thin(q.grad))
    # ---- V2: DensityMLP(63,256,3) + ColorMLP(256,27,128) -> VolumeRenderer -> NeRFLoss (rgb + 0.01 * mean(weights^2))
    p2 = O.make_weights("v2", 1, "solid", n_layers=3)
    keep = select_rays(O.relu_margin(p2, "v2", pts.reshape(-1, 3), dirs.reshape(-1, 3)), cand, S, R)
    dm, cm = ref_mlp.DensityMLP(63, 256, 3), ref_mlp.ColorMLP(256, 27, 128)
    dm.load_state_dict({k[len("density_mlp."):]: v for k, v in p2.items() if k.startswith("density_mlp.")})
    cm.load_state_dict({k[len("color_mlp."):]: v for k, v in p2.items() if k.startswith("color_mlp.")})
    vr, crit = ref_mlp.VolumeRenderer(), ref_mlp.NeRFLoss()
    pe10, pe4 = ref_mlp.PositionalEncoding(10), ref_mlp.PositionalEncoding(4)
    with torch.enable_grad():
        dn, ft = dm(pe10(pts[keep].reshape(-1, 3)))
        col = cm(ft, pe4(dirs[keep].reshape(-1, 3)))
        rgb_map, depth_map, w = vr(col.reshape(R, S, 3), dn.reshape(R, S, 1), z[keep], rd[keep])
        losses = crit({"rgb": rgb_map, "depth": depth_map, "weights": w}, {"rgb": tgt_all[keep]})
        losses["total"].backward()
    out.update(v2_pts=npf(pts[keep]), v2_dirs=npf(dirs[keep]), v2_z=npf(z[keep]), v2_rays_d=npf(rd[keep]), v2_target=npf(tgt_all[keep]),
               v2_pred=npf(rgb_map), v2_loss=np.float32(losses["total"].item()))
    for k, q in dm.named_parameters():
        out["v2_grad_density_mlp." + k] = npf(thin(q.grad))
    for k, q in cm.named_parameters():
        out["v2_grad_color_mlp." + k] = npf(thin(q.grad))
    # ---- V3: NeRFWithDINO(pos_freq=12, dino_dim=64, num_density_layers=3) -> VolumeRenderer -> mse (train.py:229-287, use_dino=True)
    p3 = O.make_weights("v3", 2, "solid", n_layers=3)
    dino_all = torch.from_numpy(O.uniform01(303, cand * S * 64).reshape(cand, S, 64) * 2 - 1).float()
    keep = select_rays(O.relu_margin(p3, "v3", pts.reshape(-1, 3), dirs.reshape(-1, 3), dino_all.reshape(-1, 64)), cand, S, R, margin=4e-5)
    m3 = ref_mlp.NeRFWithDINO(pos_freq=12, dir_freq=4, dino_dim=64, hidden_dim=256, num_density_layers=3)
    sd = m3.state_dict()
    for k in sd:
        if k in p3:
            sd[k] = p3[k]
    m3.load_state_dict(sd)
    with torch.enable_grad():
        col, dn = m3(pts[keep].reshape(-1, 3), dirs[keep].reshape(-1, 3), dino_all[keep].reshape(-1, 64))
        rgb_map, _, _ = vr(col.reshape(R, S, 3), dn.reshape(R, S, 1), z[keep], rd[keep])
        loss = torch.nn.functional.mse_loss(rgb_map, tgt_all[keep])
        loss.backward()
    out.update(v3_pts=npf(pts[keep]), v3_dirs=npf(dirs[keep]), v3_dino=npf(dino_all[keep]), v3_z=npf(z[keep]), v3_rays_d=npf(rd[keep]),
               v3_target=npf(tgt_all[keep]), v3_pred=npf(rgb_map), v3_loss=np.float32(loss.item()))
    for k, q in m3.named_parameters():
        if k in p3:
            out["v3_grad_" + k] = npf(thin(q.grad))
    save("train_grads", **out)


if __name__ == "__main__":
    if "--training-only" not in sys.argv:
        main()
    training()
