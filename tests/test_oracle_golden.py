"""Pin the CPU oracle (oracle/nerf_oracle.py) against (a) the closed-form known
answers K1..K6 of SURVEY.md section 4 and (b) golden vectors captured from the
reference's own leaf modules (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

TOL = 1e-6      # oracle vs reference on identical torch ops: expect (near) bit equality


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, tol=TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.max(np.abs(a - b), initial=0.0) <= tol, float(np.max(np.abs(a - b)))


# ---- closed-form KATs (values typed from SURVEY.md section 4, not from the fixture) ----

def test_k1_rays_closed_form(golden):
    ro, rd = O.get_rays(2, 3, 2.0, torch.eye(4))
    want = np.array([[-.75, .5, -1], [-.25, .5, -1], [.25, .5, -1], [-.75, 0, -1], [-.25, 0, -1], [.25, 0, -1]], np.float32)
    assert np.array_equal(rd.reshape(-1, 3).numpy(), want)
    assert np.array_equal(ro.numpy(), np.zeros((2, 3, 3), np.float32))
    g = golden("kat")
    assert np.array_equal(g["k1_rays_d"], rd.numpy())


def test_k2_samples_closed_form(golden):
    pts, z = O.sample_points_along_rays(torch.zeros(1, 3), torch.tensor([[0., 0., -1.]]), 2.0, 6.0, 5)
    assert np.array_equal(z.numpy(), np.array([[2, 3, 4, 5, 6]], np.float32))
    assert np.array_equal(pts[0, :, 2].numpy(), -np.array([2, 3, 4, 5, 6], np.float32))
    g = golden("kat")
    assert np.array_equal(g["k2_z"], z.numpy()) and np.array_equal(g["k2_pts"], pts.numpy())


def test_k3_encoding_closed_form(golden):
    enc = O.positional_encoding(torch.tensor([.5, -1., 2.]), 2).numpy()
    want = [.5, -1, 2, .47942555, -.84147096, .90929741, .87758255, .54030234, -.41614684,
            .84147096, -.90929741, -.7568025, .54030234, -.41614684, -.65364361]
    close(enc, np.array(want, np.float32), 1e-7)
    assert np.array_equal(golden("kat")["k3_enc"], enc)
    assert enc.shape[0] == O.encoded_dim(2) == 15


@pytest.mark.parametrize("tag,sigma,d,white,rgb,depth", [
    ("k4", [.5, 1., 0.], [0, 0, -1.], False, [.63212055, .31809238, 0.], 2.5366106),
    ("k4w", [.5, 1., 0.], [0, 0, -1.], True, [.68190759, .36787942, .04978704], 2.5366106),
    ("k5", [.5, 1., 1e-3], [0, 0, -1.], False, [.63212055, .31809238, .04978706], 2.8353329),
    ("k6", [.5, 1., 0.], [0, 0, -2.], False, [.86466473, .13285652, 0.], 2.2607555),
])
def test_k4_k6_compositor_closed_form(golden, tag, sigma, d, white, rgb, depth):
    c, dep, w = O.volume_render(torch.eye(3)[None], torch.tensor(sigma)[None, :, None],
                                torch.tensor([[2., 4., 6.]]), torch.tensor([d]), white_bkgd=white)
    close(c[0], np.array(rgb, np.float32), 2e-7)
    close(dep, np.array([depth], np.float32), 5e-7)
    g = golden("kat")
    close(g[tag + "_rgb"], c.numpy(), 1e-7)
    close(g[tag + "_depth"], dep.numpy(), 1e-7)
    close(g[tag + "_w"], w.numpy(), 1e-7)
    if tag == "k5":
        assert abs(float(w.sum()) - 1.0) < 1e-6
    # a10 on the same ray gives the same rgb when white_bkgd is off
    if not white:
        img = O.volume_render_radiance(torch.cat([torch.eye(3), torch.tensor(sigma)[:, None]], -1)[None, None],
                                       torch.tensor([[[2., 4., 6.]]]), torch.tensor([[d]]))
        close(img[0, 0], c[0], 1e-7)


# ---- golden vectors from the reference's modules ----

def test_rays_golden(golden):
    g = golden("rays")
    ro, rd = O.get_rays(int(g["H"]), int(g["W"]), float(g["focal"]), T(g["c2w"]))
    assert np.array_equal(rd.numpy(), g["rays_d"]) and np.array_equal(ro.numpy(), g["rays_o"])
    ro2, rd2 = O.get_rays(int(g["H2"]), int(g["W2"]), float(g["focal2"]), T(g["c2w"])[:3, :4])
    assert np.array_equal(rd2.numpy(), g["rays_d2"]) and np.array_equal(ro2.numpy(), g["rays_o2"])


def test_samples_golden(golden):
    g = golden("samples")
    o, d, S = T(g["rays_o"]), T(g["rays_d"]), int(g["S"])
    pts, z = O.sample_points_along_rays(o, d, 2.0, 6.0, S)
    close(z, g["z_plain"]); close(pts, g["pts_plain"])
    pts, z = O.sample_points_along_rays(o, d, 2.0, 6.0, S, t_rand=T(g["t_rand"]))
    close(z, g["z_jit"]); close(pts, g["pts_jit"])
    pts, z = O.sample_points_along_rays(o, d, 2.0, 6.0, S, lindisp=True)
    close(z, g["z_lindisp"]); close(pts, g["pts_lindisp"])
    # image layout (ray_sampler.py) == flat layout
    H, W = 5, 7
    pts, z = O.sample_points_along_rays(o.reshape(H, W, 3), d.reshape(H, W, 3), 2.0, 6.0, S, t_rand=T(g["t_rand"]).reshape(H, W, S))
    close(pts, g["pts_img_jit"])
    for s in (2, 3, 32, 64, 128, 192):
        close(O.z_steps(2.0, 6.0, s), g[f"z_S{s}"])


def test_encoding_golden(golden):
    g = golden("encoding")
    for L in (4, 10, 12):
        e = O.positional_encoding(T(g["x"]), L)
        assert e.shape[1] == O.encoded_dim(L)
        close(e, g[f"enc_L{L}"])
    g = golden("encoding_linear")                      # log_sampling=False (positional_encoding.py:17-18)
    close(O.positional_encoding(T(g["x"]), 6, log_sampling=False), g["enc_L6"])
    close(O.positional_encoding(T(g["x"]), 10, include_input=False, log_sampling=False), g["enc_L10_noinput"])


@pytest.mark.parametrize("scene", ["fog", "solid"])
def test_mlp_v1_golden(golden, scene):
    g = golden(f"mlp_v1_{scene}")
    out = O.mlp_v1(O.make_weights("v1", 0, scene), T(g["x_enc"]))
    close(out, g["out"], 2e-5 if scene == "solid" else 2e-6)


def test_mlp_v2_golden(golden):
    g = golden("mlp_v2")
    p = O.make_weights("v2", 1)
    rgb, dens = O.mlp_v2(p, T(g["pos"]), T(g["dirs"]))
    close(rgb, g["rgb"], 2e-6); close(dens, g["density"], 2e-6)
    _, feat = O.density_mlp(p, "density_mlp.", O.positional_encoding(T(g["pos"]), 10))
    close(feat, g["feature"], 2e-6)


def test_mlp_v3_golden(golden):
    g = golden("mlp_v3")
    p = O.make_weights("v3", 2)
    rgb, dens = O.mlp_v3(p, T(g["pos"]), T(g["dirs"]), T(g["dino"]))
    close(rgb, g["rgb"], 2e-6); close(dens, g["density"], 2e-6)
    fused = O.dino_fusion(p, "dino_fusion.", O.positional_encoding(T(g["pos"]), 12), T(g["dino"]))
    close(fused, g["fused"], 2e-6)


def test_mlp_v3_multiscale_width_golden(golden):
    g = golden("mlp_v3_d128")
    p = O.make_weights("v3", 3, dino_dim=128)
    rgb, dens = O.mlp_v3(p, T(g["pos"]), T(g["dirs"]), T(g["dino"]))
    close(rgb, g["rgb"], 2e-6); close(dens, g["density"], 2e-5)


def test_param_counts():
    # 477 956 and 821 190 parameters
    assert sum(v.numel() for v in O.make_weights("v1").values()) == 477956
    assert sum(v.numel() for v in O.make_weights("v3").values()) == 821190


def test_dino_fetch_golden(golden):
    g = golden("dino_fetch")
    xy, dep, mask = O.project_points_to_image(T(g["points"]), T(g["pose"]), float(g["focal"]), int(g["H"]), int(g["W"]))
    close(dep, g["depth"], 1e-5)
    assert np.array_equal(mask.numpy(), g["mask"])
    # projected coordinates can be huge near Z=0; compare relatively
    ref = g["xy"].astype(np.float64)
    assert np.all(np.abs(xy.numpy() - ref) <= 1e-4 * (1 + np.abs(ref)))
    close(O.sample_features_at_points(T(g["features"]), T(g["xy"])), g["sampled"], 2e-6)
    close(O.sample_features_at_points(T(g["features"]), T(g["xy_in"])), g["sampled_in"], 2e-6)
    assert np.abs(g["sampled_in"]).max() > 0.1        # the in-range case really hits the map


def test_composite_golden(golden):
    g = golden("composite")
    c, d, w = O.volume_render(T(g["rgb_in"]), T(g["sigma_in"]), T(g["z"]), T(g["rays_d"]))
    close(c, g["rgb"]); close(d, g["depth"]); close(w, g["weights"])
    c1, _, _ = O.volume_render(T(g["rgb_in"]), T(g["sigma_in"]), T(g["z"]), T(g["rays_d"]), white_bkgd=True)
    close(c1, g["rgb_white"])
    img = O.volume_render_radiance(torch.cat([T(g["rgb_in"]), T(g["sigma_in"])], -1).reshape(8, 12, -1, 4),
                                   T(g["z"]).reshape(8, 12, -1), T(g["rays_d"]).reshape(8, 12, 3))
    close(img, g["radiance"])
    # properties: sum w <= 1 (+eps), w >= 0
    assert w.min() >= 0 and w.sum(-1).max() <= 1 + 1e-5


@pytest.mark.parametrize("variant,scenes", [("v1", ("fog", "solid")), ("v2", ("fog", "solid")), ("v3", ("fog",))])
def test_end_to_end_golden(golden, variant, scenes):
    g = golden("end_to_end")
    H, W, S = int(g["H"]), int(g["W"]), int(g["S"])
    ro, rd = O.get_rays(H, W, float(g["focal"]), T(g["c2w"]))
    seed = {"v1": 0, "v2": 1, "v3": 2}[variant]
    for scene in scenes:
        p = O.make_weights(variant, seed, scene)
        dino = None
        if variant == "v3":
            fm = torch.from_numpy(O.uniform01(7, 28 * 28 * 64).reshape(1, 28, 28, 64) * 2 - 1)
            dino = dict(features=fm, pose=T(g["c2w"]), focal=float(g["focal"]), H=H, W=W)
        for tag, tr in (("plain", None), ("jit", T(g["t_rand"]))):
            out = O.render_rays(p, variant, ro, rd, 2.0, 6.0, S, t_rand=tr, dino=dino, chunk=64)
            tol = 5e-6
            close(out["rgb"], g[f"{variant}_{scene}_{tag}_rgb"], tol)
            close(out["depth"], g[f"{variant}_{scene}_{tag}_depth"], 2e-5)
            close(out["weights"], g[f"{variant}_{scene}_{tag}_w"], tol)


def test_sample_pdf_properties():
    """a3 has no reference output -- parity UNPINNED; check the intent's invariants."""
    R, S, Ni = 7, 16, 8
    z = O.z_steps(2.0, 6.0, S).expand(R, S).contiguous()
    w = torch.from_numpy(O.uniform01(3, R * S).reshape(R, S))
    w[:, 5] += 10.0                                       # a spike: most new samples must land in bin 5
    new, union = O.sample_pdf(z, w, Ni)
    assert union.shape == (R, S + Ni) and new.shape == (R, Ni)
    assert torch.all(union[:, 1:] >= union[:, :-1])
    assert new.min() >= 2.0 and new.max() <= 6.0
    lo, hi = 0.5 * (z[0, 4] + z[0, 5]), 0.5 * (z[0, 5] + z[0, 6])
    assert ((new >= lo) & (new <= hi)).float().mean() >= 0.45


def test_uniform01_is_stable():
    u = O.uniform01(1234, 5)
    assert u.dtype == np.float32 and np.all((u >= 0) & (u < 1))
    # frozen values: the GPU box must regenerate the same inputs
    assert np.array_equal(u, O.uniform01(1234, 8)[:5])
    assert [int(x * 2 ** 24) for x in O.uniform01(0, 3)] == [int(x * 2 ** 24) for x in O.uniform01(0, 3)]


# ---------------------------------------------------------------------------------------------
# training path: the oracle's autograd against loss.backward() of the reference's own modules
# ---------------------------------------------------------------------------------------------
def thin(t):
    return t[::4] if t.ndim == 2 and t.numel() > 20000 else t


def test_training_gradients_v1_match_reference_backward(golden):
    g = golden("train_grads")
    R, S = g["v1_z"].shape
    p = O.make_weights("v1", 0, "solid", n_layers=3)
    pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    x = O.positional_encoding(torch.from_numpy(g["v1_pts"]).reshape(-1, 3), 10)
    pred = O.volume_render_radiance(O.mlp_v1(pp, x).reshape(R, 1, S, 4), torch.from_numpy(g["v1_z"]).reshape(R, 1, S),
                                    torch.from_numpy(g["v1_rays_d"]).reshape(R, 1, 3)).reshape(R, 3)
    loss = torch.nn.functional.mse_loss(pred, torch.from_numpy(g["v1_target"]))
    loss.backward()
    assert abs(loss.item() - float(g["v1_loss"])) < 1e-6
    assert np.abs(pred.detach().numpy() - g["v1_pred"]).max() < 1e-6
    n = 0
    for k, v in pp.items():
        ref = g["v1_grad_" + k]
        assert np.abs(thin(v.grad).numpy() - ref).max() <= 1e-5 * np.abs(ref).max(), k
        n += 1
    assert n == 10


def test_training_gradients_v2_match_reference_backward(golden):
    """train.py's model + VolumeRenderer + nerf_mlp.NeRFLoss (rgb + 0.01 * mean(weights^2)): nerf_mlp.py:217-256."""
    g = golden("train_grads")
    R, S = g["v2_z"].shape
    p = O.make_weights("v2", 1, "solid", n_layers=3)
    pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    rgb, den = O.mlp_v2(pp, torch.from_numpy(g["v2_pts"]).reshape(-1, 3), torch.from_numpy(g["v2_dirs"]).reshape(-1, 3))
    rgb_map, _, w = O.volume_render(rgb.reshape(R, S, 3), den.reshape(R, S, 1), torch.from_numpy(g["v2_z"]), torch.from_numpy(g["v2_rays_d"]))
    loss = torch.nn.functional.mse_loss(rgb_map, torch.from_numpy(g["v2_target"])) + 0.01 * torch.mean(w ** 2)
    loss.backward()
    assert abs(loss.item() - float(g["v2_loss"])) < 1e-6
    for k, v in pp.items():
        ref = g["v2_grad_" + k]
        assert np.abs(thin(v.grad).numpy() - ref).max() <= 1e-5 * np.abs(ref).max(), k


def test_training_gradients_v3_match_reference_backward(golden):
    """NeRFWithDINO (fusion block twice on the same weights, softmax gate) + VolumeRenderer + mse."""
    g = golden("train_grads")
    R, S = g["v3_z"].shape
    p = O.make_weights("v3", 2, "solid", n_layers=3)
    pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    rgb, den = O.mlp_v3(pp, torch.from_numpy(g["v3_pts"]).reshape(-1, 3), torch.from_numpy(g["v3_dirs"]).reshape(-1, 3),
                        torch.from_numpy(g["v3_dino"]).reshape(-1, 64))
    rgb_map = O.volume_render(rgb.reshape(R, S, 3), den.reshape(R, S, 1), torch.from_numpy(g["v3_z"]), torch.from_numpy(g["v3_rays_d"]))[0]
    loss = torch.nn.functional.mse_loss(rgb_map, torch.from_numpy(g["v3_target"]))
    loss.backward()
    assert abs(loss.item() - float(g["v3_loss"])) < 1e-6
    n = 0
    for k, v in pp.items():
        ref = g["v3_grad_" + k]
        assert np.abs(thin(v.grad).numpy() - ref).max() <= 1e-5 * np.abs(ref).max(), k
        n += 1
    assert n == 2 * (3 + 10)


def test_emulated_training_arithmetic_is_autograd_in_fp32():
    """The rounding-aware restatement used to check the 16-bit kernels is, without rounding, exactly autograd."""
    p = O.make_weights("v1", 0, "solid")
    x = torch.from_numpy(O.uniform01(3, 200 * 63).reshape(200, 63) * 2 - 1).float()
    gg = torch.from_numpy(O.uniform01(4, 800).reshape(200, 4) - 0.5).float()
    pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    out = O.mlp_v1(pp, x)
    (out * gg).sum().backward()
    out2, grads, _, _ = O.mlp_v1_train_emulated(p, x, gg, "f32")
    assert torch.equal(out.detach(), out2)
    for k, v in grads.items():
        assert (pp[k].grad - v).abs().max() <= 2e-6 * pp[k].grad.abs().max(), k
