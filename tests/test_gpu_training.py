"""GPU parity tests of the training path, through the C ABI.

Checker: torch autograd on the CPU oracle (fp32) and, for the 16-bit MFMA modes, the oracle's
rounding-aware restatement `mlp_v1_train_emulated` (operands rounded exactly where the kernels
round them, fp32 accumulation).  Parity of the backward is unpinned by the reference (it has no
tests); the anchor is autograd through the oracle's restatement of the reference's forward.

Tolerances: fp32 mode -- every gradient within 2e-4 of its tensor's max |value| (summation order only);
16-bit modes -- every stage within one operand-type ulp of the oracle's arithmetic on the same stage inputs,
weight gradients within 1e-4 of the products of the saved tensors, cosine against fp32 autograd.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    import nerf_few_shot_limitations_amd as N
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from nerf_few_shot_limitations_amd import _lib
    _lib.lib()
    return N


def make_model(N, mode, scene="fog", n_layers=8, seed=0):
    m = N.NeRFMLP(pos_dim=63, hidden_dim=256, n_layers=n_layers, mma_mode=mode)
    p = O.make_weights("v1", seed, scene, n_layers=n_layers)
    m.load_state_dict(p)
    return m.cuda().train(), p


def inputs(n, seed=3):
    pts = torch.from_numpy(O.uniform01(seed, n * 3).reshape(n, 3) * 4 - 2).float()
    x = O.positional_encoding(pts, 10)
    g = torch.from_numpy(O.uniform01(seed + 1, n * 4).reshape(n, 4) - 0.5).float()
    return x, g


# Samples with a ReLU pre-activation closer to 0 than this get no incoming gradient in the fp32-mode comparisons: their
# mask is decided by summation order (and, for V2, by the last bit of sin/cos), and one flipped mask changes the sample's
# whole gradient below it -- on either side of the comparison (oracle.relu_margin).
MARGIN = 2e-5


def rel_to_max(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def cosine(a, b):
    a, b = a.detach().cpu().double().flatten(), b.detach().cpu().double().flatten()
    return float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300))


# ---------------------------------------------------------------------------------------------
# decode the saved tensors (csrc/train_core.hpp layout) for stage-wise checks
# ---------------------------------------------------------------------------------------------
def ctx_slot(buf, mode, n, n_layers, slot):
    """Saved-tensor slot -> (features, padded samples) fp32 matrix."""
    tiles32 = (n + 255) // 256 * 8
    slot_tiles = [2] + [8] * (2 * n_layers) + [1]
    tb = 4096 if mode == "f32" else 2048
    off = sum(slot_tiles[:slot]) * tiles32 * tb
    KT = slot_tiles[slot]
    raw = buf[off:off + tiles32 * KT * tb].cpu().numpy()
    if mode == "f32":
        v = raw.view(np.float32).reshape(tiles32, KT, 4, 64, 4)            # st, t, vec, lane, e
    else:
        u = raw.view(np.uint16).reshape(tiles32, KT, 2, 64, 8)
        if mode == "bf16":
            v = (u.astype(np.uint32) << 16).view(np.float32)
        else:
            v = u.view(np.float16).astype(np.float32)
    nv, ne = v.shape[2], v.shape[4]
    out = np.zeros((32 * KT, 32 * tiles32), np.float32)
    lanes = np.arange(64)
    c, h = lanes & 31, lanes >> 5
    for vec in range(nv):
        for e in range(ne):
            r = ne * vec + e
            row = (r & 3) + 8 * (r >> 2) + 4 * h                           # accumulator row map
            for t in range(KT):
                # out[32t + row[lane], 32 st + c[lane]] = v[st, t, vec, lane, e]
                out[(32 * t + row)[None, :], (32 * np.arange(tiles32)[:, None] + c[None, :])] = v[:, t, vec, :, e]
    return out


def kernel_feature_order(L=10):
    """Row of saved slot 0 -> index into the reference's 63 encoded features (-1 = padding); feature_map.hpp."""
    KT = (3 * L + 2 + 15) // 16
    idx = np.full(32 * KT, -1)
    for u in range(16 * KT):
        for h in range(2):
            t, r = u // 16, u % 16
            k = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h
            if u < 3 * L:
                idx[k] = 3 + 6 * (u // 3) + 3 * h + (u % 3)
            elif u == 3 * L:
                idx[k] = 2 if h else 0
            elif u == 3 * L + 1:
                idx[k] = -1 if h else 1
    return idx


def close_but_for_mask_flips(a, e, bound, allowed=4):
    """A pre-activation within summation-order noise of 0 flips its ReLU mask: a handful of elements may differ by a
    whole gradient value; everything else must be within `bound`."""
    return int((np.abs(a - e) > bound).sum()) <= allowed


def run_raw(N, model, x, g):
    """One forward_train / backward pair through the C ABI; returns out4, flat grad, the context buffer."""
    from nerf_few_shot_limitations_amd import _lib as L
    from nerf_few_shot_limitations_amd.training import _train_handle
    dev = torch.device("cuda", 0)
    h, mode = _train_handle(model, dev)
    n = x.shape[0]
    xd, gd = x.cuda().contiguous(), g.cuda().contiguous()
    out = torch.empty((n, 4), device=dev)
    nbytes = L.lib().nrf_train_context_bytes(h, mode, n)
    assert nbytes > 0
    buf = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    L.check(L.lib().nrf_mlp_forward_train_v1(h, mode, L.ptr(xd), n, L.ptr(out), C.c_void_p(buf.data_ptr()), nbytes, L.stream_ptr()))
    grad = torch.zeros(model.flat_params().flat.numel(), device=dev)
    L.check(L.lib().nrf_mlp_backward_v1(h, mode, L.ptr(out), L.ptr(gd), n, C.c_void_p(buf.data_ptr()), nbytes, L.ptr(grad), L.stream_ptr()))
    torch.cuda.synchronize()
    return out, grad, buf


def named_grads(model, flat_grad):
    fp = model.flat_params()
    names = []
    for i in range(model.n_layers):
        names += [f"layers.{i}.weight", f"layers.{i}.bias"]
    names += ["sigma_out.weight", "sigma_out.bias", "rgb_out.weight", "rgb_out.bias"]
    return dict(zip(names, fp.views(flat_grad)))


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["f32", "bf16", "f16"])
def test_forward_train_equals_inference_forward(N, mode):
    """Same chain, same arithmetic: the saving forward must reproduce nrf_mlp_forward_v1 bit for bit."""
    model, _ = make_model(N, mode)
    x, g = inputs(1000)
    out, _, _ = run_raw(N, model, x, g)
    with torch.no_grad():
        ref = model.eval()(x.cuda())
    assert torch.equal(out, ref)


def test_saved_tensors_match_oracle_fp32(N):
    """Stage-wise, fp32 mode: every saved activation and every saved dZ against the oracle."""
    n, mode, tol = 300, "f32", 2e-5
    model, p = make_model(N, mode)
    x, g = inputs(n)
    out, _, buf = run_raw(N, model, x, g)
    o_out, _, acts, dzs = O.mlp_v1_train_emulated(p, x, g, mode)
    assert rel_to_max(out, o_out) < 1e-5
    order = kernel_feature_order()
    a0 = ctx_slot(buf, mode, n, 8, 0)
    exp0 = np.zeros_like(a0[:, :n])
    for k, src in enumerate(order):
        if src >= 0:
            exp0[k] = acts[0][:, src].numpy()
    assert np.abs(a0[:, :n] - exp0).max() <= 1e-6
    for l in range(1, 9):
        a = ctx_slot(buf, mode, n, 8, l)[:, :n]
        e = acts[l].numpy().T
        assert close_but_for_mask_flips(a, e, tol * max(1.0, np.abs(e).max())), f"activation of layer {l}"
    for l in range(1, 9):
        d = ctx_slot(buf, mode, n, 8, 8 + l)
        e = dzs[l - 1].numpy().T
        assert close_but_for_mask_flips(d[:, :n], e, tol * np.abs(e).max()), f"dZ of layer {l}"
        assert np.abs(d[:, n:]).max() == 0.0, "padding samples must carry no gradient"
    dh = ctx_slot(buf, mode, n, 8, 17)
    assert np.abs(dh[:4, :n] - dzs[8].numpy().T).max() <= tol * np.abs(dzs[8].numpy()).max()
    assert np.abs(dh[4:]).max() == 0.0


@pytest.mark.parametrize("mode,n", [("bf16", 300), ("f16", 300), ("bf16", 33000)])
def test_saved_tensors_stage_consistent_16bit(N, mode, n):
    """16-bit modes: ReLU masks of near-zero pre-activations differ between any two summation orders, and one flipped
    mask changes a sample's whole gradient below it, so tensors deep in the chain cannot be compared element-wise with
    an independent run.  Instead every stage is checked against the oracle's arithmetic applied to the GPU's OWN saved
    inputs of that stage: one operand-type ulp per element (rounding of the last bit), a few mask flips allowed.
    n = 300 runs the small-batch geometry (4 waves per workgroup), n = 33000 the throughput geometry (8 waves)."""
    ulp = 2.0 ** -7 if mode == "bf16" else 2.0 ** -10
    model, p = make_model(N, mode)
    x, g = inputs(n)
    out, grad, buf = run_raw(N, model, x, g)
    q = lambda t: O.quantize(t, mode)
    acts = [torch.from_numpy(ctx_slot(buf, mode, n, 8, l)[:, :n].T.copy()) for l in range(9)]      # (n, features)
    dzs = [torch.from_numpy(ctx_slot(buf, mode, n, 8, 8 + l)[:, :n].T.copy()) for l in range(1, 9)]
    dhead = torch.from_numpy(ctx_slot(buf, mode, n, 8, 17)[:4, :n].T.copy())

    def stage_close(a, e, what):
        bound = ulp * e.abs() + 1e-6 * e.abs().max()
        assert int(((a - e).abs() > bound).sum()) <= 4 + n // 2000, what

    order = kernel_feature_order()
    w0 = torch.zeros(256, 64)
    for k, src in enumerate(order):
        if src >= 0:
            w0[:, k] = q(p["layers.0.weight"])[:, src]
    stage_close(acts[1], q(torch.relu(acts[0] @ w0.T + p["layers.0.bias"])), "layer 0")
    for l in range(1, 8):
        stage_close(acts[l + 1], q(torch.relu(acts[l] @ q(p[f"layers.{l}.weight"]).T + p[f"layers.{l}.bias"])), f"layer {l}")
    rgb = torch.sigmoid(acts[8] @ q(p["rgb_out.weight"]).T + p["rgb_out.bias"])
    sigma = acts[8] @ q(p["sigma_out.weight"]).T + p["sigma_out.bias"]
    assert (out.cpu() - torch.cat([rgb, sigma], -1)).abs().max() < 2e-3        # fast sigmoid in the 16-bit modes
    o = out.cpu()
    stage_close(dhead, q(torch.cat([g[:, :3] * o[:, :3] * (1 - o[:, :3]), g[:, 3:4]], -1)), "head gradient")
    dh = dhead[:, :3] @ q(p["rgb_out.weight"]) + dhead[:, 3:4] @ q(p["sigma_out.weight"])
    for l in range(7, -1, -1):
        stage_close(dzs[l], q(dh * (acts[l + 1] > 0)), f"dZ of layers.{l}")
        if l > 0:
            dh = dzs[l] @ q(p[f"layers.{l}.weight"])
    # weight gradients from the saved tensors: fp32 accumulation on both sides
    got = named_grads(model, grad)
    exp0 = dzs[0].T @ acts[0]                                                  # kernel feature order -> reference columns
    e0 = torch.zeros(256, 63)
    for k, src in enumerate(order):
        if src >= 0:
            e0[:, src] = exp0[:, k]
    assert rel_to_max(got["layers.0.weight"], e0) < 1e-4
    for l in range(8):
        if l > 0:
            assert rel_to_max(got[f"layers.{l}.weight"], dzs[l].T @ acts[l]) < 1e-4, l
        assert rel_to_max(got[f"layers.{l}.bias"], dzs[l].sum(0)) < 1e-4, l
    assert rel_to_max(got["rgb_out.weight"], dhead[:, :3].T @ acts[8]) < 1e-4
    assert rel_to_max(got["sigma_out.weight"], dhead[:, 3:4].T @ acts[8]) < 1e-4
    assert rel_to_max(got["rgb_out.bias"], dhead[:, :3].sum(0)) < 1e-4
    assert rel_to_max(got["sigma_out.bias"], dhead[:, 3:4].sum(0)) < 1e-4


@pytest.mark.parametrize("n", [1, 255, 1000, 4096 + 17])
def test_gradients_fp32_mode_match_autograd(N, n):
    model, p = make_model(N, "f32", scene="solid")
    x, g = inputs(n, seed=11)
    g = g * (O.relu_margin(p, "v1", x) > MARGIN)[:, None]
    _, grad, _ = run_raw(N, model, x, g)
    pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    (O.mlp_v1(pp, x) * g).sum().backward()
    for name, gv in named_grads(model, grad).items():
        assert rel_to_max(gv, pp[name].grad) < 2e-4, name


@pytest.mark.parametrize("mode,cos_min,n", [("bf16", 0.97, 3000), ("f16", 0.995, 3000), ("bf16", 0.97, 40000)])
def test_gradients_16bit_modes_vs_fp32_autograd(N, mode, cos_min, n):
    """End to end against fp32 autograd: direction of every weight gradient (operand rounding + mask flips are the
    difference; the stage-consistent test above bounds the arithmetic itself).  n = 40000: the 8-wave geometry."""
    model, p = make_model(N, mode)
    x, g = inputs(n, seed=21)
    _, grad, _ = run_raw(N, model, x, g)
    pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    (O.mlp_v1(pp, x) * g).sum().backward()
    for name, gv in named_grads(model, grad).items():
        if name.endswith("weight"):
            assert cosine(gv, pp[name].grad) > cos_min, (name, cosine(gv, pp[name].grad))


@pytest.mark.parametrize("mode", ["bf16", "f32"])
def test_gradients_are_bit_reproducible(N, mode):
    """Partial sums are reduced in a fixed order (weight_grad_reduce_kernel): the same inputs give the same bits."""
    model, _ = make_model(N, mode, scene="solid")
    x, g = inputs(5000, seed=41)
    _, a, _ = run_raw(N, model, x, g)
    _, b, _ = run_raw(N, model, x, g)
    assert torch.equal(a, b)
    m3, _ = make_v3(N, mode, scene="solid")                      # the fusion block's weights receive two sums each
    pos, dirs, dino, g_rgb, g_den = v3_inputs(3000)
    outs = []
    for _ in range(2):
        for q in m3.parameters():
            q.grad = None
        rgb, den = m3(pos.cuda(), dirs.cuda(), dino.cuda())
        ((rgb * g_rgb.cuda()).sum() + (den * g_den.cuda()).sum()).backward()
        outs.append(m3._flat_grad.clone())
    assert torch.equal(outs[0], outs[1])


def test_backward_accumulates_into_the_gradient_vector(N):
    """flat_grad += : two backward calls double the gradient."""
    from nerf_few_shot_limitations_amd import _lib as L
    model, _ = make_model(N, "f32")
    x, g = inputs(512)
    out, grad, buf = run_raw(N, model, x, g)
    once = grad.clone()
    h = model._handle
    L.check(L.lib().nrf_mlp_backward_v1(h, 2, L.ptr(out), L.ptr(g.cuda()), 512, C.c_void_p(buf.data_ptr()), buf.numel(), L.ptr(grad), L.stream_ptr()))
    torch.cuda.synchronize()
    assert rel_to_max(grad, 2 * once) < 1e-5


@pytest.mark.parametrize("n_layers", [2, 3, 5])
def test_other_depths(N, n_layers):
    model, p = make_model(N, "f32", n_layers=n_layers)
    x, g = inputs(700, seed=31)
    g = g * (O.relu_margin(p, "v1", x) > MARGIN)[:, None]
    _, grad, _ = run_raw(N, model, x, g)
    pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    (O.mlp_v1(pp, x) * g).sum().backward()
    for name, gv in named_grads(model, grad).items():
        assert rel_to_max(gv, pp[name].grad) < 2e-4, name


# ---------------------------------------------------------------------------------------------
# compositing backward
# ---------------------------------------------------------------------------------------------
def composite_case(R, S, seed, opaque):
    rgb = torch.from_numpy(O.uniform01(seed, R * S * 3).reshape(R, S, 3)).float()
    sig = torch.from_numpy(O.uniform01(seed + 1, R * S).reshape(R, S, 1) * 3 - 1).float()       # a third of the samples have sigma <= 0
    if opaque:
        sig[:, S // 3] = 40.0                                                                    # an opaque sample mid-ray
    z = torch.sort(torch.from_numpy(O.uniform01(seed + 2, R * S).reshape(R, S) * 4 + 2).float(), dim=-1).values
    d = torch.from_numpy(O.uniform01(seed + 3, R * 3).reshape(R, 3) - 0.5).float()
    return rgb, sig, z, d


@pytest.mark.parametrize("S,opaque,white", [(32, False, False), (64, True, False), (100, True, True), (200, False, True)])
def test_composite_backward_matches_autograd(N, S, opaque, white):
    from nerf_few_shot_limitations_amd.training import composite
    R = 257
    rgb, sig, z, d = composite_case(R, S, 41, opaque)
    g_rgb = torch.from_numpy(O.uniform01(51, R * 3).reshape(R, 3) - 0.5).float()
    g_dep = torch.from_numpy(O.uniform01(52, R).reshape(R) - 0.5).float()
    g_w = torch.from_numpy(O.uniform01(53, R * S).reshape(R, S) - 0.5).float()
    # oracle
    r1, s1 = rgb.clone().requires_grad_(True), sig.clone().requires_grad_(True)
    o_rgb, o_dep, o_w = O.volume_render(r1, s1, z, d, white_bkgd=white)
    ((o_rgb * g_rgb).sum() + (o_dep * g_dep).sum() + (o_w * g_w).sum()).backward()
    # HIP
    packed = torch.cat([rgb, sig], -1).cuda().requires_grad_(True)
    h_rgb, h_dep, h_w = composite(packed, z.cuda(), d.cuda(), white)
    assert (h_rgb.cpu() - o_rgb.detach()).abs().max() < 1e-5
    ((h_rgb * g_rgb.cuda()).sum() + (h_dep * g_dep.cuda()).sum() + (h_w * g_w.cuda()).sum()).backward()
    got = packed.grad.cpu()
    assert rel_to_max(got[..., :3], r1.grad) < 1e-4
    # d sigma spans many orders of magnitude (dist = 1e10 on the last sample): compare relative to each ray's largest entry
    ds, es = got[..., 3], s1.grad[..., 0]
    scale = es.abs().amax(dim=1, keepdim=True).clamp_min(1e-20)
    assert float(((ds - es).abs() / scale).max()) < 1e-4


def test_composite_backward_rgb_only(N):
    """train.py's loss uses predictions['rgb'] only: depth / weights gradients are absent (None), not zeros."""
    from nerf_few_shot_limitations_amd.training import composite
    R, S = 100, 48
    rgb, sig, z, d = composite_case(R, S, 61, True)
    tgt = torch.from_numpy(O.uniform01(62, R * 3).reshape(R, 3)).float()
    r1, s1 = rgb.clone().requires_grad_(True), sig.clone().requires_grad_(True)
    torch.nn.functional.mse_loss(O.volume_render(r1, s1, z, d)[0], tgt).backward()
    packed = torch.cat([rgb, sig], -1).cuda().requires_grad_(True)
    torch.nn.functional.mse_loss(composite(packed, z.cuda(), d.cuda())[0], tgt.cuda()).backward()
    assert rel_to_max(packed.grad[..., :3], r1.grad) < 1e-4
    assert rel_to_max(packed.grad[..., 3], s1.grad[..., 0]) < 1e-4


# ---------------------------------------------------------------------------------------------
# optimizer, re-pack, whole steps
# ---------------------------------------------------------------------------------------------
def test_adam_kernel_matches_torch(N):
    from nerf_few_shot_limitations_amd import _lib as L
    n = 100003
    p0 = torch.from_numpy(O.uniform01(71, n) - 0.5).float()
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=5e-4, weight_decay=1e-6)
    p = p0.clone().cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 6):
        g = torch.from_numpy(O.uniform01(72 + step, n) - 0.5).float() * (10.0 ** (step - 3))
        ref.grad = g.clone()
        opt.step()
        L.check(L.lib().nrf_adam_step(L.ptr(p), L.ptr(g.cuda()), L.ptr(m), L.ptr(v), n, 5e-4, 0.9, 0.999, 1e-8, 1e-6, step, L.stream_ptr()))
    torch.cuda.synchronize()
    assert (p.cpu() - ref.detach()).abs().max() < 1e-6


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_device_repack_equals_host_pack(N, mode):
    """nrf_model_update_device must produce the very streams nrf_model_create packs on the host."""
    a, p = make_model(N, mode, scene="solid")
    b, _ = make_model(N, mode, scene="fog")          # other weights first, then the same ones through the device path
    x, g = inputs(777)
    run_raw(N, b, x, g)                              # builds b's flat vector and training state
    with torch.no_grad():
        for name, t in b.state_dict().items():
            t.copy_(p[name].cuda())
    b._gen += 1
    with torch.no_grad():
        ya = a.eval()(x.cuda())
        yb = b.eval()(x.cuda())
    assert torch.equal(ya, yb)
    _, ga, _ = run_raw(N, a.train(), x, g)
    _, gb, _ = run_raw(N, b.train(), x, g)
    assert rel_to_max(ga, gb) < 1e-5                 # atomics: order only


@pytest.mark.parametrize("optim_name", ["adam", "sgd"])
def test_training_steps_match_cpu_reference_loop(N, optim_name):
    """train_minimal.py:97-123 in miniature: encode -> NeRFMLP -> volume_render_radiance -> mse -> optimizer, three steps,
    fp32 mode, against the same loop on the CPU oracle.  With Adam (the reference's optimizer) the losses must agree; the
    parameters are compared under SGD, whose update is linear in the gradient (Adam's first steps are ~lr*sign(g): an
    element whose gradient is summation-order noise around 0 moves by +-lr on either side)."""
    H = W = 12
    S = 16
    steps = 3
    make = (lambda ps: torch.optim.Adam(ps, lr=5e-4)) if optim_name == "adam" else (lambda ps: torch.optim.SGD(ps, lr=1e-2))
    c2w = torch.from_numpy(O.LEGO_LIKE_C2W.copy())
    ro, rd = O.get_rays(H, W, O.focal_for(W), c2w)
    pts, z = O.sample_points_along_rays(ro.reshape(-1, 3), rd.reshape(-1, 3), 2.0, 6.0, S)
    x = O.positional_encoding(pts.reshape(-1, 3), 10)
    target = torch.from_numpy(O.uniform01(81, H * W * 3).reshape(H, W, 3)).float()
    # CPU loop
    p = O.make_weights("v1", 0, "solid")
    pp = {k: torch.nn.Parameter(v.clone()) for k, v in p.items()}
    opt = make(list(pp.values()))
    cpu_losses = []
    for _ in range(steps):
        opt.zero_grad()
        pred = O.volume_render_radiance(O.mlp_v1(pp, x).reshape(H, W, S, 4), z.reshape(H, W, S), rd)
        loss = torch.nn.functional.mse_loss(pred, target)
        loss.backward()
        opt.step()
        cpu_losses.append(loss.item())
    # HIP loop: the reference's own call sequence on the drop-in surface, a torch optimizer on the module's parameters
    model, _ = make_model(N, "f32", scene="solid")
    opt = make(list(model.parameters()))
    gpu_losses = []
    xd, zd, rdd, td = x.cuda(), z.reshape(H, W, S).cuda(), rd.cuda(), target.cuda()
    for _ in range(steps):
        opt.zero_grad()
        pred = N.volume_render_radiance(model(xd).view(H, W, S, 4), zd, rdd)
        loss = torch.nn.functional.mse_loss(pred, td)
        loss.backward()
        opt.step()
        gpu_losses.append(loss.item())
    assert np.allclose(cpu_losses, gpu_losses, rtol=2e-4, atol=1e-6), (cpu_losses, gpu_losses)
    bound = 2.5 * 5e-4 * steps if optim_name == "adam" else 1e-5
    for name, t in model.state_dict().items():
        assert (t.cpu() - pp[name].detach()).abs().max() < bound, name


def test_fused_adam_loop_reduces_loss(N):
    """bf16 training with the flat-vector Adam kernel: the loss on a fixed batch must go down."""
    from nerf_few_shot_limitations_amd.training import Adam
    model, _ = make_model(N, "bf16", scene="fog")
    x, _ = inputs(4096, seed=91)
    target = torch.from_numpy(O.uniform01(92, 4096 * 4).reshape(4096, 4)).float().cuda()
    opt = Adam(model, lr=1e-3)
    xd = x.cuda()
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(model(xd), target)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.7 * losses[0], losses


def test_dino_features_that_require_grad_are_refused(N):
    """The DINO extractor (LoRA included) is outside the HIP path: no gradient flows to the features, and saying so
    beats silently dropping it."""
    m = N.NeRFMLP(pos_freq=12, dir_freq=4, hidden_dim=256, num_density_layers=8, use_dino=True, dino_dim=64).cuda().train()
    with pytest.raises(NotImplementedError):
        m(torch.zeros(4, 3).cuda(), torch.zeros(4, 3).cuda(), torch.zeros(4, 64).cuda().requires_grad_(True))


# ---------------------------------------------------------------------------------------------
# V3: NeRFWithDINO (train.py with use_dino=True): fusion block twice on the same weights, softmax gate
# ---------------------------------------------------------------------------------------------
def make_v3(N, mode, scene="fog", n_layers=8, dino_dim=64, seed=2):
    m = N.NeRFMLP(pos_freq=12, dir_freq=4, hidden_dim=256, num_density_layers=n_layers, use_dino=True, dino_dim=dino_dim, mma_mode=mode)
    p = O.make_weights("v3", seed, scene, n_layers=n_layers, dino_dim=dino_dim)
    m.load_state_dict(p, strict=False)
    return m.cuda().train(), p


def v3_inputs(n, dino_dim=64, seed=25):
    pos, dirs, g_rgb, g_den = v2_inputs(n, seed)
    dino = torch.from_numpy(O.uniform01(seed + 9, n * dino_dim).reshape(n, dino_dim) * 2 - 1).float()
    return pos, dirs, dino, g_rgb, g_den


@pytest.mark.parametrize("n,n_layers,dino_dim", [(1000, 8, 64), (300, 8, 128), (4096 + 17, 8, 64), (500, 3, 64), (500, 2, 64)])
def test_v3_gradients_fp32_mode_match_autograd(N, n, n_layers, dino_dim):
    model, p = make_v3(N, "f32", scene="solid", n_layers=n_layers, dino_dim=dino_dim)
    pos, dirs, dino, g_rgb, g_den = v3_inputs(n, dino_dim)
    keep = (O.relu_margin(p, "v3", pos, dirs, dino) > MARGIN)[:, None]
    assert keep.float().mean() > 0.85
    g_rgb, g_den = g_rgb * keep, g_den * keep
    rgb, den = model(pos.cuda(), dirs.cuda(), dino.cuda())
    ((rgb * g_rgb.cuda()).sum() + (den * g_den.cuda()).sum()).backward()
    pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    o_rgb, o_den = O.mlp_v3(pp, pos, dirs, dino)
    assert (rgb.detach().cpu() - o_rgb.detach()).abs().max() < 1e-4 and rel_to_max(den, o_den) < 1e-4
    ((o_rgb * g_rgb).sum() + (o_den * g_den).sum()).backward()
    checked = 0
    for name, q in model.named_parameters():
        if name in pp:
            assert rel_to_max(q.grad, pp[name].grad) < 2e-4, (name, rel_to_max(q.grad, pp[name].grad))
            checked += 1
    assert checked == 2 * (n_layers + 10)


@pytest.mark.parametrize("mode,cos_min,n", [("bf16", 0.90, 3000), ("f16", 0.99, 3000), ("bf16", 0.90, 40000)])
def test_v3_gradients_16bit_modes_vs_fp32_autograd(N, mode, cos_min, n):
    model, p = make_v3(N, mode)
    pos, dirs, dino, g_rgb, g_den = v3_inputs(n, 64, seed=35)
    rgb, den = model(pos.cuda(), dirs.cuda(), dino.cuda())
    ((rgb * g_rgb.cuda()).sum() + (den * g_den.cuda()).sum()).backward()
    pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    o_rgb, o_den = O.mlp_v3(pp, pos, dirs, dino)
    ((o_rgb * g_rgb).sum() + (o_den * g_den).sum()).backward()
    for name, q in model.named_parameters():
        # density_head's gradient is gated by relu'(density): on this scene the densities hover around 0 and their sign, after
        # a 16-bit forward through the fusion block twice and the trunk, is rounding noise -- a 256-element vector built from
        # that gate is not a meaningful direction test
        if name in pp and name.endswith("weight") and "density_head" not in name:
            assert cosine(q.grad, pp[name].grad) > cos_min, (name, cosine(q.grad, pp[name].grad))


# ---------------------------------------------------------------------------------------------
# V2: the network train.py builds (use_dino=False)
# ---------------------------------------------------------------------------------------------
def make_v2(N, mode, scene="fog", n_layers=8, seed=1):
    m = N.NeRFMLP(pos_freq=10, dir_freq=4, hidden_dim=256, num_density_layers=n_layers, use_dino=False, mma_mode=mode)
    p = O.make_weights("v2", seed, scene, n_layers=n_layers)
    m.load_state_dict(p, strict=False)
    return m.cuda().train(), p


def v2_inputs(n, seed=5):
    pos = torch.from_numpy(O.uniform01(seed, n * 3).reshape(n, 3) * 4 - 2).float()
    dirs = torch.from_numpy(O.uniform01(seed + 1, n * 3).reshape(n, 3) * 2 - 1).float()
    g_rgb = torch.from_numpy(O.uniform01(seed + 2, n * 3).reshape(n, 3) - 0.5).float()
    g_den = torch.from_numpy(O.uniform01(seed + 3, n).reshape(n, 1) - 0.5).float()
    return pos, dirs, g_rgb, g_den


def v2_grads(model):
    return {name: p.grad for name, p in model.named_parameters()}


@pytest.mark.parametrize("n,n_layers", [(1000, 8), (257, 8), (4096 + 17, 8), (600, 3), (600, 2)])
def test_v2_gradients_fp32_mode_match_autograd(N, n, n_layers):
    model, p = make_v2(N, "f32", scene="solid", n_layers=n_layers)
    pos, dirs, g_rgb, g_den = v2_inputs(n)
    keep = (O.relu_margin(p, "v2", pos, dirs) > MARGIN)[:, None]
    assert keep.float().mean() > 0.9
    g_rgb, g_den = g_rgb * keep, g_den * keep
    rgb, den = model(pos.cuda(), dirs.cuda())
    ((rgb * g_rgb.cuda()).sum() + (den * g_den.cuda()).sum()).backward()
    pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    o_rgb, o_den = O.mlp_v2(pp, pos, dirs)
    assert (rgb.detach().cpu() - o_rgb.detach()).abs().max() < 1e-4 and rel_to_max(den, o_den) < 1e-4
    ((o_rgb * g_rgb).sum() + (o_den * g_den).sum()).backward()
    checked = 0
    for name, g in v2_grads(model).items():
        if name in pp:
            assert rel_to_max(g, pp[name].grad) < 2e-4, name
            checked += 1
    assert checked == 2 * (n_layers + 5)


@pytest.mark.parametrize("mode,cos_min,n", [("bf16", 0.97, 3000), ("f16", 0.995, 3000), ("bf16", 0.97, 40000)])
def test_v2_gradients_16bit_modes_vs_fp32_autograd(N, mode, cos_min, n):
    model, p = make_v2(N, mode)
    pos, dirs, g_rgb, g_den = v2_inputs(n, seed=15)
    rgb, den = model(pos.cuda(), dirs.cuda())
    ((rgb * g_rgb.cuda()).sum() + (den * g_den.cuda()).sum()).backward()
    pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    o_rgb, o_den = O.mlp_v2(pp, pos, dirs)
    ((o_rgb * g_rgb).sum() + (o_den * g_den).sum()).backward()
    for name, g in v2_grads(model).items():
        if name in pp and name.endswith("weight"):
            assert cosine(g, pp[name].grad) > cos_min, (name, cosine(g, pp[name].grad))


def test_v2_train_step_as_train_py_runs_it(N):
    """train.py:188-242,280-288 on the drop-in surface: sample -> model(pts, dirs) -> VolumeRenderer -> mse on 'rgb' ->
    backward -> Adam; three steps in fp32 mode against the same loop on the CPU oracle."""
    R, S, steps = 96, 24, 3
    c2w = torch.from_numpy(O.LEGO_LIKE_C2W.copy())
    ro, rd = O.get_rays(16, 16, O.focal_for(16), c2w)
    ro, rd = ro.reshape(-1, 3)[:R], rd.reshape(-1, 3)[:R]
    t_rand = torch.from_numpy(O.uniform01(7, R * S).reshape(R, S)).float()
    pts, z = O.sample_points_along_rays(ro, rd, 2.0, 6.0, S, t_rand)
    dirs = rd[:, None, :].expand(R, S, 3).reshape(-1, 3)                      # train.py:225: raw ray directions per sample
    target = torch.from_numpy(O.uniform01(8, R * 3).reshape(R, 3)).float()
    p = O.make_weights("v2", 1, "solid")
    pp = {k: torch.nn.Parameter(v.clone()) for k, v in p.items()}
    opt = torch.optim.Adam(list(pp.values()), lr=5e-4, weight_decay=1e-6)       # baseline.yaml:39-40
    cpu_losses = []
    for _ in range(steps):
        opt.zero_grad()
        rgb, den = O.mlp_v2(pp, pts.reshape(-1, 3), dirs)
        pred = O.volume_render(rgb.reshape(R, S, 3), den.reshape(R, S, 1), z, rd)[0]
        loss = torch.nn.functional.mse_loss(pred, target)
        loss.backward()
        opt.step()
        cpu_losses.append(loss.item())
    model, _ = make_v2(N, "f32", scene="solid")
    vr = N.VolumeRenderer()
    opt = torch.optim.Adam(model.parameters(), lr=5e-4, weight_decay=1e-6)
    ptsd, dirsd, zd, rdd, td = pts.reshape(-1, 3).cuda(), dirs.cuda(), z.cuda(), rd.cuda(), target.cuda()
    gpu_losses = []
    for _ in range(steps):
        opt.zero_grad()
        rgb, den = model(ptsd, dirsd, None)
        pred, _, _ = vr(rgb.reshape(R, S, 3), den.reshape(R, S, 1), zd, rdd)
        loss = torch.nn.functional.mse_loss(pred, td)
        loss.backward()
        opt.step()
        gpu_losses.append(loss.item())
    assert np.allclose(cpu_losses, gpu_losses, rtol=2e-4, atol=1e-6), (cpu_losses, gpu_losses)


# ---------------------------------------------------------------------------------------------
# golden: loss.backward() of the reference's own modules (tests/golden/make_golden.py: training())
# ---------------------------------------------------------------------------------------------
def thin(t):
    return t[::4] if t.ndim == 2 and t.numel() > 20000 else t


def test_v1_backward_matches_reference_golden(N, golden):
    g = golden("train_grads")
    R, S = g["v1_z"].shape
    model, _ = make_model(N, "f32", scene="solid", n_layers=3)
    x = O.positional_encoding(torch.from_numpy(g["v1_pts"]).reshape(-1, 3), 10).cuda()
    pred = N.volume_render_radiance(model(x).view(R, 1, S, 4), torch.from_numpy(g["v1_z"]).view(R, 1, S).cuda(),
                                    torch.from_numpy(g["v1_rays_d"]).view(R, 1, 3).cuda()).view(R, 3)
    loss = torch.nn.functional.mse_loss(pred, torch.from_numpy(g["v1_target"]).cuda())
    loss.backward()
    assert abs(loss.item() - float(g["v1_loss"])) < 1e-5 * float(g["v1_loss"]) + 1e-7
    assert (pred.detach().cpu().numpy() - g["v1_pred"]).max() < 1e-4
    for name, q in model.named_parameters():
        ref = g["v1_grad_" + name]
        assert np.abs(thin(q.grad).cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max(), name


def test_v2_backward_matches_reference_golden(N, golden):
    g = golden("train_grads")
    R, S = g["v2_z"].shape
    model, _ = make_v2(N, "f32", scene="solid", n_layers=3)
    rgb, den = model(torch.from_numpy(g["v2_pts"]).reshape(-1, 3).cuda(), torch.from_numpy(g["v2_dirs"]).reshape(-1, 3).cuda(), None)
    rgb_map, depth_map, w = N.VolumeRenderer()(rgb.reshape(R, S, 3), den.reshape(R, S, 1), torch.from_numpy(g["v2_z"]).cuda(),
                                              torch.from_numpy(g["v2_rays_d"]).cuda())
    loss = torch.nn.functional.mse_loss(rgb_map, torch.from_numpy(g["v2_target"]).cuda()) + 0.01 * torch.mean(w ** 2)   # nerf_mlp.py:236-252
    loss.backward()
    assert abs(loss.item() - float(g["v2_loss"])) < 1e-5 * float(g["v2_loss"]) + 1e-7
    for name, q in model.named_parameters():
        ref = g["v2_grad_" + name]
        assert np.abs(thin(q.grad).cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max(), name


def test_v3_backward_matches_reference_golden(N, golden):
    g = golden("train_grads")
    R, S = g["v3_z"].shape
    model, _ = make_v3(N, "f32", scene="solid", n_layers=3)
    rgb, den = model(torch.from_numpy(g["v3_pts"]).reshape(-1, 3).cuda(), torch.from_numpy(g["v3_dirs"]).reshape(-1, 3).cuda(),
                     torch.from_numpy(g["v3_dino"]).reshape(-1, 64).cuda())
    rgb_map = N.VolumeRenderer()(rgb.reshape(R, S, 3), den.reshape(R, S, 1), torch.from_numpy(g["v3_z"]).cuda(),
                                 torch.from_numpy(g["v3_rays_d"]).cuda())[0]
    loss = torch.nn.functional.mse_loss(rgb_map, torch.from_numpy(g["v3_target"]).cuda())
    loss.backward()
    assert abs(loss.item() - float(g["v3_loss"])) < 1e-5 * float(g["v3_loss"]) + 1e-7
    checked = 0
    for name, q in model.named_parameters():
        if "v3_grad_" + name in g:
            ref = g["v3_grad_" + name]
            assert np.abs(thin(q.grad).cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max(), name
            checked += 1
    assert checked == 26


# ---------------------------------------------------------------------------------------------
# FusedStep: the same step without autograd in between
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("net,mode", [("v1", "f32"), ("v1", "bf16"), ("v2", "f32"), ("v2", "bf16"), ("v3", "f32"), ("v3", "bf16")])
def test_fused_step_equals_autograd_route(N, net, mode):
    from nerf_few_shot_limitations_amd.training import Adam, FusedStep
    R, S, steps = 160, 32, 4
    z = torch.sort(torch.from_numpy(O.uniform01(101, R * S).reshape(R, S) * 4 + 2).float(), dim=-1).values.cuda()
    rd = torch.from_numpy(O.uniform01(102, R * 3).reshape(R, 3) - 0.5).float().cuda()
    tgt = torch.from_numpy(O.uniform01(103, R * 3).reshape(R, 3)).float().cuda()
    pos = torch.from_numpy(O.uniform01(104, R * S * 3).reshape(R * S, 3) * 4 - 2).float()
    dirs = rd[:, None, :].expand(R, S, 3).reshape(-1, 3).contiguous()
    if net == "v1":
        a, _ = make_model(N, mode, scene="solid")
        b, _ = make_model(N, mode, scene="solid")
        pts = O.positional_encoding(pos, 10).cuda()
    elif net == "v2":
        a, _ = make_v2(N, mode, scene="solid")
        b, _ = make_v2(N, mode, scene="solid")
        pts = pos.cuda()
    else:
        a, _ = make_v3(N, mode, scene="solid")
        b, _ = make_v3(N, mode, scene="solid")
        pts = pos.cuda()
    dino = torch.from_numpy(O.uniform01(105, R * S * 64).reshape(R * S, 64) * 2 - 1).float().cuda() if net == "v3" else None
    opt = Adam(a, lr=5e-4, weight_decay=1e-6)
    vr = N.VolumeRenderer()
    ref_losses = []
    for _ in range(steps):
        opt.zero_grad()
        if net == "v1":
            pred = N.volume_render_radiance(a(pts).view(R, 1, S, 4), z.view(R, 1, S), rd.view(R, 1, 3)).view(R, 3)
        else:
            c, sg = a(pts, dirs, dino)
            pred = vr(c.view(R, S, 3), sg.view(R, S, 1), z, rd)[0]
        loss = torch.nn.functional.mse_loss(pred, tgt)
        loss.backward()
        opt.step()
        ref_losses.append(loss.item())
    step = FusedStep(b, lr=5e-4, weight_decay=1e-6)
    got = [step(pts, z, rd, tgt, dirs=dirs if net != "v1" else None, dino=dino).item() for _ in range(steps)]
    # same kernels on both routes; only the fp32 atomics of the weight gradients are order dependent, and Adam's first
    # steps turn gradient noise around 0 into +-lr -- the 16-bit modes then see different operand roundings
    assert np.allclose(ref_losses, got, rtol=1e-5 if mode == "f32" else 2e-3, atol=1e-7), (ref_losses, got)
    assert got[-1] < got[0]


# ---------------------------------------------------------------------------------------------
# data-parallel training: rays shard over the ranks, ONE all-reduce of the flat gradient vector per step
# ---------------------------------------------------------------------------------------------
def _run_dp(route, steps, tmp_path):
    import os, socket, subprocess, sys
    out = str(tmp_path / f"dp_{route}.npy")
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "tests", "dp_train_worker.py"), route, out, str(steps)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return np.load(out)


@pytest.mark.parametrize("route", ["autograd_sgd", "fused"])
def test_data_parallel_two_ranks_equal_one_rank_on_the_whole_batch(N, route, tmp_path):
    """Two ranks (gloo here: the test box has one GPU, which both ranks share; RCCL on a real node), each on half of the
    rays, averaged gradients: the parameters after 3 steps equal a single process training on the whole batch.  SGD for the
    autograd route (linear in the gradient); Adam for FusedStep, compared with the looser bound its sign-like first
    steps allow (see test_training_steps_match_cpu_reference_loop)."""
    import importlib.util, os
    from nerf_few_shot_limitations_amd.training import Adam, FusedStep
    steps = 3
    got = _run_dp(route, steps, tmp_path)
    spec = importlib.util.spec_from_file_location("dp_worker", os.path.join(os.path.dirname(__file__), "dp_train_worker.py"))
    worker = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(worker)
    R, S = 128, 16
    x, z, rd, tgt = worker.batch(R, S)
    model, _ = make_model(N, "f32", scene="solid")
    xs, zs, rds, tgts = x.reshape(-1, 63).cuda(), z.cuda(), rd.cuda(), tgt.cuda()
    if route == "fused":
        step = FusedStep(model, lr=1e-2)
        for _ in range(steps):
            step(xs, zs, rds, tgts)
        bound = 2.5 * 1e-2 * steps
    else:
        opt = torch.optim.SGD(model.parameters(), lr=1e-2)
        for _ in range(steps):
            opt.zero_grad()
            pred = N.volume_render_radiance(model(xs).view(R, 1, S, 4), zs.view(R, 1, S), rds.view(R, 1, 3)).view(R, 3)
            torch.nn.functional.mse_loss(pred, tgts).backward()
            opt.step()
        bound = 1e-5
    ref = model.flat_params().flat.detach().cpu().numpy()
    assert np.abs(got - ref).max() < bound, np.abs(got - ref).max()
    if route == "fused":
        assert np.mean(np.abs(got - ref) > 1e-4) < 0.02          # all but the elements whose gradient is noise around 0


# ---------------------------------------------------------------------------------------------
# the training run as a command (train.py:244-292,344-389 on the HIP path)
# ---------------------------------------------------------------------------------------------
def _write_scene(root, n_train=2, n_test=2, size=16):
    import json, os
    from PIL import Image
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / (size - 1)
    for split, count in (("train", n_train), ("test", n_test)):
        os.makedirs(os.path.join(root, split))
        frames = []
        for i in range(count):
            # a smooth, bright target: with dark targets the quickest way down is density -> 0 everywhere, after which
            # relu(density_head) passes no gradient at all (nerf_mlp.py:63) and the loss freezes
            img = np.stack([0.6 + 0.4 * xx, 0.6 + 0.4 * yy, np.full_like(xx, 0.7 + 0.2 * i / max(count - 1, 1)), np.ones_like(xx)], -1)
            Image.fromarray((img * 255).astype(np.uint8), "RGBA").save(os.path.join(root, split, f"r_{i}.png"))
            pose = O.LEGO_LIKE_C2W.copy(); pose[0, 3] += 0.05 * i
            frames.append({"file_path": f"./{split}/r_{i}", "transform_matrix": pose.tolist()})
        json.dump({"camera_angle_x": O.CAMERA_ANGLE_X, "frames": frames}, open(os.path.join(root, f"transforms_{split}.json"), "w"))


_CFG = ("experiment: {{name: tiny}}\ndata: {{near: 2.0, far: 6.0, resolution: 16, num_views: 2}}\n"
        "rendering: {{near: 2.0, far: 6.0, chunk_size: 2048, white_bkgd: false}}\nmodel: {{use_dino: {dino}}}\n"
        "nerf_model: {{pos_freq: {pf}, dir_freq: 4, hidden_dim: 256, num_layers: 8}}\n"
        "training: {{epochs: 3, batch_size: 64, progressive_schedule: {{epochs_0_50: [16, 16, 8], epochs_50_100: [16, 16, 12], epochs_100_plus: [16, 16, 16]}}}}\n"
        "optimizer: {{lr: 2.0e-3, weight_decay: 1.0e-6, lr_milestones: [2], lr_gamma: 0.5}}\nloss: {{rgb_weight: 1.0, depth_weight: 0.0, reg_weight: 0.0}}\n"
        "output: {{save_dir: unused, val_freq: 2, save_freq: 3}}\n"
        "dino_model: {{name: facebook/dinov2-small, lora_rank: 4, lora_alpha: 8, use_lora: true}}\n")


@pytest.mark.parametrize("use_dino", [False, True])
def test_train_cli_runs_the_reference_schedule(N, tmp_path, use_dino):
    import json, os
    from nerf_few_shot_limitations_amd import evaluate_cli, train_cli
    root = str(tmp_path / "scene")
    _write_scene(root)
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text(_CFG.format(dino="true" if use_dino else "false", pf=12 if use_dino else 10))
    out = str(tmp_path / "run")
    # start from the synthetic "fog" weights: torch's default init leaves every density at relu(~bias) and, when that bias
    # happens to be negative, the whole network without gradient (the reference shares this trap: nerf_mlp.py:63)
    p = dict(O.make_weights("v3" if use_dino else "v2", 2 if use_dino else 1, "fog"))
    p["pos_encoder.freq_bands"] = 2.0 ** torch.linspace(0., 11 if use_dino else 9, 12 if use_dino else 10)
    p["dir_encoder.freq_bands"] = 2.0 ** torch.linspace(0., 3, 4)
    torch.save({"epoch": 0, "nerf_model_state_dict": p}, str(tmp_path / "init.pth"))
    argv = ["--config", str(cfg), "--data", root, "--out", out, "--mode", "f32", "--epochs", "3", "--checkpoint", str(tmp_path / "init.pth")]
    if use_dino:
        maps = torch.from_numpy(O.uniform01(77, 2 * 4 * 4 * 64).reshape(2, 4, 4, 64) * 2 - 1).float()
        torch.save(maps, str(tmp_path / "maps.pt"))
        argv += ["--dino-maps", str(tmp_path / "maps.pt")]
    log = train_cli.main(argv)
    assert [r["epoch"] for r in log] == [1, 2, 3]
    assert log[0]["lr"] == pytest.approx(2e-3) and log[2]["lr"] == pytest.approx(1e-3)           # MultiStepLR milestone at epoch 2
    # 12 optimiser steps at a test-sized learning rate: the run must move (that it moves the right way is what the gradient
    # and FusedStep tests establish; a dozen noisy steps from a synthetic init need not be monotone)
    assert all(np.isfinite(r["loss"]) for r in log) and log[-1]["loss"] != log[0]["loss"]
    assert "psnr" in log[1] and "psnr" in log[2] and np.isfinite(log[2]["psnr"])
    files = sorted(os.listdir(out))
    assert "best_tiny.pth" in files and "epoch_3.pth" in files and "train_log.json" in files and "val_2" in files
    ck = torch.load(os.path.join(out, "best_tiny.pth"), map_location="cpu", weights_only=True)
    assert "nerf_model_state_dict" in ck and "density_mlp.density_head.weight" in ck["nerf_model_state_dict"]
    moved = max(float((ck["nerf_model_state_dict"][k] - p[k]).abs().max()) for k in p if k.endswith("weight"))
    assert 1e-4 < moved < 0.1                                                                     # Adam at lr 2e-3 / 1e-3 for a handful of steps
    if not use_dino:                                                                              # the checkpoint feeds the evaluation command
        m = evaluate_cli.main(["--config", str(cfg), "--data", root, "--checkpoint", os.path.join(out, "best_tiny.pth"), "--mode", "f32"])
        assert m["views"] == 2 and abs(m["psnr"] - max(r.get("psnr", 0) for r in log)) < 1e-3
        # --checkpoint resumes the run: epoch counter, Adam moments and step count continue (5 epochs = 3 + 2)
        e3 = torch.load(os.path.join(out, "epoch_3.pth"), map_location="cpu", weights_only=True)
        assert e3["epoch"] == 2 and e3["optimizer_state_dict"]["step"] > 0 and e3["scheduler_state_dict"]["last_epoch"] == 3
        out2 = str(tmp_path / "run2")
        log2 = train_cli.main(["--config", str(cfg), "--data", root, "--out", out2, "--mode", "f32", "--epochs", "5",
                               "--checkpoint", os.path.join(out, "epoch_3.pth")])
        assert [r["epoch"] for r in log2] == [4, 5] and log2[0]["lr"] == pytest.approx(1e-3)
        steps_per_epoch = e3["optimizer_state_dict"]["step"] // 3
        one = train_cli.main(["--config", str(cfg), "--data", root, "--out", str(tmp_path / "run3"), "--mode", "f32", "--epochs", "4",
                              "--checkpoint", os.path.join(out, "epoch_3.pth")])
        assert [r["epoch"] for r in one] == [4] and steps_per_epoch > 0
        assert one[0]["loss"] == pytest.approx(log2[0]["loss"], rel=1e-6)                         # same restored state, same --seed: the resumed epoch is reproducible
    else:
        # the extractor route of train.py:57-75,158-169: maps produced once by the config's SpatialDINOFeatures (random init here:
        # the DINOv2 weights are not available offline), then the same training loop
        log3 = train_cli.main(["--config", str(cfg), "--data", root, "--out", str(tmp_path / "run_x"), "--mode", "f32", "--epochs", "1",
                               "--checkpoint", str(tmp_path / "init.pth"), "--dino-random-init"])
        assert [r["epoch"] for r in log3] == [1] and np.isfinite(log3[0]["loss"])


def test_train_cli_schedule_and_lr_rules():
    from nerf_few_shot_limitations_amd import train_cli
    cfg = {"training": {"batch_size": 1024, "progressive_schedule": {"epochs_0_50": [32, 32, 32], "epochs_50_100": [64, 64, 48], "epochs_100_plus": [128, 128, 64]}},
           "optimizer": {"lr": 5e-4, "lr_milestones": [100, 150], "lr_gamma": 0.5}}
    assert train_cli.schedule_for(cfg, 0) == (32, 32, 32, 2048) and train_cli.schedule_for(cfg, 49) == (32, 32, 32, 2048)      # train.py:249-259
    assert train_cli.schedule_for(cfg, 50) == (64, 64, 48, 1024) and train_cli.schedule_for(cfg, 100) == (128, 128, 64, 512)
    assert train_cli.lr_at(cfg, 99) == 5e-4 and train_cli.lr_at(cfg, 100) == 2.5e-4 and train_cli.lr_at(cfg, 150) == 1.25e-4


def test_render_in_another_mode_after_training_sees_the_new_weights(N):
    """The device-side re-pack refreshes only the streams that are asked for: a render in a mode the model was not
    trained in must still use the trained parameters."""
    from nerf_few_shot_limitations_amd.training import FusedStep
    model, _ = make_model(N, "bf16", scene="solid")
    c2w = torch.from_numpy(O.LEGO_LIKE_C2W.copy())
    before, _ = N.render_camera(model.eval(), 16, 16, O.focal_for(16), c2w, 2.0, 6.0, 16, mma_mode="f32")
    R, S = 64, 16
    x = O.positional_encoding(torch.from_numpy(O.uniform01(111, R * S * 3).reshape(R * S, 3) * 4 - 2).float(), 10).cuda()
    z = torch.sort(torch.from_numpy(O.uniform01(112, R * S).reshape(R, S) * 4 + 2).float(), dim=-1).values.cuda()
    rd = torch.from_numpy(O.uniform01(113, R * 3).reshape(R, 3) - 0.5).float().cuda()
    tgt = torch.ones(R, 3).cuda()
    step = FusedStep(model.train(), lr=1e-2)
    for _ in range(3):
        step(x, z, rd, tgt)
    after, _ = N.render_camera(model.eval(), 16, 16, O.focal_for(16), c2w, 2.0, 6.0, 16, mma_mode="f32")
    ref = N.NeRFMLP(pos_dim=63, hidden_dim=256, n_layers=8, mma_mode="f32")
    ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    want, _ = N.render_camera(ref.cuda().eval(), 16, 16, O.focal_for(16), c2w, 2.0, 6.0, 16)
    assert not torch.equal(before, after)
    assert torch.equal(after, want)


def test_training_entry_points_take_empty_batches(N):
    from nerf_few_shot_limitations_amd import _lib as L
    model, _ = make_model(N, "bf16")
    x, g = inputs(8)
    run_raw(N, model, x, g)                                   # builds the training state
    h = model._handle
    assert L.lib().nrf_train_context_bytes(h, 0, 0) == 0
    assert L.lib().nrf_mlp_forward_train_v1(h, 0, None, 0, None, None, 0, None) == 0
    assert L.lib().nrf_mlp_backward_v1(h, 0, None, None, 0, None, 0, None, None) == 0
    assert L.lib().nrf_composite_backward(None, 3, None, 1, None, None, 0, 8, 0, None, None, None, None, 3, None, 1, None) == 0


@pytest.mark.parametrize("n", [3, 6144, 100003])
def test_mse_grad_kernel_matches_torch(N, n):
    from nerf_few_shot_limitations_amd import _lib as L
    pred = torch.from_numpy(O.uniform01(121, n)).float().cuda().requires_grad_(True)
    tgt = torch.from_numpy(O.uniform01(122, n)).float().cuda()
    ref = 0.7 * torch.nn.functional.mse_loss(pred, tgt)
    ref.backward()
    g = torch.empty(n, device="cuda")
    loss = torch.empty((), device="cuda")
    L.check(L.lib().nrf_mse_grad(L.ptr(pred.detach()), L.ptr(tgt), n, 0.7, L.ptr(g), L.ptr(loss), L.stream_ptr()))
    torch.cuda.synchronize()
    assert abs(loss.item() - ref.item()) < 1e-6 * max(1.0, abs(ref.item()))
    assert (g - pred.grad).abs().max() < 1e-7


@pytest.mark.parametrize("R,S,opaque,white,packed", [(257, 32, False, False, True), (64, 64, True, False, False), (100, 100, True, True, True),
                                                     (33, 200, False, True, False), (1, 1, False, False, True), (5000, 8, True, False, False)])
def test_composite_mse_backward_equals_the_three_calls(N, R, S, opaque, white, packed):
    """nrf_composite_mse_backward (FusedStep's one launch between the two network kernels) against nrf_composite -> nrf_mse_grad ->
    nrf_composite_backward: prediction and both gradients bit-equal (same per-ray bodies); the side job clears the caller's gradient
    vector; the loss, added up from ray_loss by nrf_adam_step_loss's side job, to fp32 summation-order noise -- and the Adam update
    of that launch is nrf_adam_step's, bit for bit."""
    from nerf_few_shot_limitations_amd import _lib as L
    import ctypes as C
    rgb, sig, z, d = composite_case(R, S, 71, opaque and S >= 3)
    tgt = torch.from_numpy(O.uniform01(72, R * 3).reshape(R, 3)).float().cuda()
    z, d = z.cuda().contiguous(), d.cuda().contiguous()
    n = R * S
    if packed:                                                 # V1 rows [r, g, b, sigma]
        o4 = torch.cat([rgb, sig], -1).reshape(n, 4).cuda().contiguous()
        heads = (L.ptr(o4), 4, C.c_void_p(o4.data_ptr() + 12), 4)
        new = lambda: torch.full((n, 4), float("nan"), device="cuda")
        d_heads = lambda t: (L.ptr(t), 4, C.c_void_p(t.data_ptr() + 12), 4)
    else:                                                      # V2 / V3: rgb (n,3) | density (n,1)
        c3, s1 = rgb.reshape(n, 3).cuda().contiguous(), sig.reshape(n, 1).cuda().contiguous()
        heads = (L.ptr(c3), 3, L.ptr(s1), 1)
        new = lambda: torch.full((4 * n,), float("nan"), device="cuda")
        d_heads = lambda t: (L.ptr(t[:3 * n]), 3, L.ptr(t[3 * n:]), 1)
    lib, st, w = L.lib(), L.stream_ptr(), 0.7
    pred_a, g_pred, loss_a, da = torch.empty(R, 3, device="cuda"), torch.empty(R, 3, device="cuda"), torch.empty((), device="cuda"), new()
    L.check(lib.nrf_composite(*heads, L.ptr(z), L.ptr(d), R, S, int(white), L.ptr(pred_a), None, None, st))
    L.check(lib.nrf_mse_grad(L.ptr(pred_a), L.ptr(tgt), 3 * R, w, L.ptr(g_pred), L.ptr(loss_a), st))
    L.check(lib.nrf_composite_backward(*heads, L.ptr(z), L.ptr(d), R, S, int(white), L.ptr(g_pred), None, None, *d_heads(da), st))
    ray_loss = torch.empty(R, device="cuda")
    pred_b, db = torch.empty(R, 3, device="cuda"), new()
    junk = torch.full((100003,), 3.0, device="cuda")
    L.check(lib.nrf_composite_mse_backward(*heads, L.ptr(z), L.ptr(d), R, S, int(white), L.ptr(tgt), w, L.ptr(pred_b), *d_heads(db),
                                           L.ptr(ray_loss), L.ptr(junk), junk.numel() - 3, st))
    torch.cuda.synchronize()
    assert torch.equal(pred_a, pred_b)
    assert torch.equal(da.view(torch.int32), db.view(torch.int32))                  # incl. the untouched (NaN) gaps: nothing else is written
    assert float(junk[:-3].abs().max()) == 0.0 and torch.equal(junk[-3:], torch.full((3,), 3.0, device="cuda"))
    assert torch.allclose(ray_loss, (pred_a - tgt).square().sum(-1), rtol=1e-6, atol=1e-12)
    # the loss through Adam's side job; the update itself equals nrf_adam_step's
    npar = 70001
    p0 = torch.from_numpy(O.uniform01(73, npar) - 0.5).float().cuda()
    g0 = torch.from_numpy(O.uniform01(74, npar) - 0.5).float().cuda()
    pa, ma, va = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    pb, mb, vb = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    loss_b = torch.empty((), device="cuda")
    for step in (1, 2):
        L.check(lib.nrf_adam_step(L.ptr(pa), L.ptr(g0), L.ptr(ma), L.ptr(va), npar, 1e-3, 0.9, 0.999, 1e-8, 1e-6, step, st))
        L.check(lib.nrf_adam_step_loss(L.ptr(pb), L.ptr(g0), L.ptr(mb), L.ptr(vb), npar, 1e-3, 0.9, 0.999, 1e-8, 1e-6, step, L.ptr(ray_loss), R, w,
                                       L.ptr(loss_b), st))
    torch.cuda.synchronize()
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert abs(loss_a.item() - loss_b.item()) <= 2e-6 * max(abs(loss_a.item()), 1e-3)
    # pred and the side job may be omitted; bad arguments are refused before any launch
    db2 = new()
    L.check(lib.nrf_composite_mse_backward(*heads, L.ptr(z), L.ptr(d), R, S, int(white), L.ptr(tgt), w, None, *d_heads(db2), L.ptr(ray_loss),
                                           None, 0, st))
    torch.cuda.synchronize()
    assert torch.equal(da.view(torch.int32), db2.view(torch.int32))
    assert lib.nrf_composite_mse_backward(*heads, L.ptr(z), L.ptr(d), R, S, int(white), L.ptr(tgt), w, None, *d_heads(db2), None, None, 0, st) == -1
    assert lib.nrf_composite_mse_backward(*heads, L.ptr(z), L.ptr(d), R, S, int(white), L.ptr(tgt), w, None, *d_heads(db2), L.ptr(ray_loss),
                                          None, 5, st) == -1
    assert lib.nrf_adam_step_loss(L.ptr(pb), L.ptr(g0), L.ptr(mb), L.ptr(vb), npar, 1e-3, 0.9, 0.999, 1e-8, 1e-6, 3, None, R, w, L.ptr(loss_b), st) == -1


@pytest.mark.parametrize("mode", ["bf16", "f32"])
def test_forward_train_input_rows_aligned_or_not_ragged_or_not(N, mode):
    """The input rows of the training forward may start anywhere a float may (a row-offset view of a larger tensor: 4-byte aligned
    only) and the batch may end anywhere in a tile: the same outputs and the same gradients, bit for bit, as from a fresh allocation.
    (Written with a variant of train_forward_kernel that staged whole tiles through LDS with 16-byte loads when it could -- measured
    no faster, not kept; the property is what callers rely on.)"""
    m, _ = make_model(N, mode, scene="solid")
    m.train()
    for n in (1, 31, 32, 33, 255, 256, 257, 1000):
        big = torch.from_numpy(O.uniform01(700 + n, (n + 1) * 63).reshape(n + 1, 63) * 2 - 1).float().cuda()
        xa = big[1:].clone()                                # fresh allocation: 256-byte aligned
        xu = big[1:]                                        # 252 bytes into an allocation: 4-byte aligned only
        assert xa.data_ptr() % 16 == 0 and xu.data_ptr() % 16 != 0 and xu.is_contiguous()
        g = torch.from_numpy(O.uniform01(800 + n, n * 4).reshape(n, 4) - 0.5).float().cuda()
        outs, grads = [], []
        for x in (xa, xu):
            m.zero_grad(set_to_none=True)
            out = m(x)
            (out * g).sum().backward()
            outs.append(out.detach().clone())
            grads.append(torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone())
        assert torch.equal(outs[0], outs[1]), n
        assert torch.equal(grads[0], grads[1]), n


@pytest.mark.parametrize("mode", ["bf16", "f32"])
def test_gradient_is_additive_over_the_batch_at_scale(N, mode):
    """Size-independent property at a batch no CPU check could cover (131 072 samples, 8-wave kernels in bf16): every
    sample's chain is independent of its neighbours, so grad(batch) = grad(first half) + grad(second half) up to fp32
    summation order -- in the 16-bit modes too -- and scaling the incoming gradient scales the result."""
    n = 131072
    model, _ = make_model(N, mode, scene="solid")
    x, g = inputs(n, seed=61)
    _, full, _ = run_raw(N, model, x, g)
    _, a, _ = run_raw(N, model, x[: n // 2], g[: n // 2])
    _, b, _ = run_raw(N, model, x[n // 2:], g[n // 2:])
    assert rel_to_max(a + b, full) < 2e-5
    if mode == "f32":
        _, half, _ = run_raw(N, model, x, 0.5 * g)            # exact in fp32: a power-of-two factor commutes with every rounding
        assert torch.equal(half * 2, full)


def test_train_cli_data_parallel_matches_one_process(N, tmp_path):
    """`--data-parallel` under torch.distributed.run with two ranks (gloo, sharing the test GPU): same shuffles, every batch
    split over the ranks, gradients averaged -- the run lands where the single-process run does (validation PSNR)."""
    import json, os, socket, subprocess, sys
    from nerf_few_shot_limitations_amd import train_cli
    root = str(tmp_path / "scene")
    _write_scene(root)
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text(_CFG.format(dino="false", pf=10))
    p = dict(O.make_weights("v2", 1, "fog"))
    p["pos_encoder.freq_bands"] = 2.0 ** torch.linspace(0., 9, 10)
    p["dir_encoder.freq_bands"] = 2.0 ** torch.linspace(0., 3, 4)
    torch.save({"epoch": 0, "nerf_model_state_dict": p}, str(tmp_path / "init.pth"))
    common = ["--config", str(cfg), "--data", root, "--mode", "f32", "--epochs", "2", "--checkpoint", str(tmp_path / "init.pth")]
    one = train_cli.main(common + ["--out", str(tmp_path / "one")])
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           "-m", "nerf_few_shot_limitations_amd.train_cli"] + common + ["--out", str(tmp_path / "two"), "--data-parallel", "--rehearse"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=repo, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    two = json.load(open(tmp_path / "two" / "train_log.json"))
    assert [e["epoch"] for e in two] == [1, 2]
    # the first epoch's mean loss: rank 0 reports the mean over ITS half of every batch -- compare the validation metric and
    # the trained weights instead, which are common to all ranks
    assert abs(two[-1]["psnr"] - one[-1]["psnr"]) < 0.05
    a = torch.load(tmp_path / "one" / "best_tiny.pth", map_location="cpu", weights_only=True)["nerf_model_state_dict"]
    b = torch.load(tmp_path / "two" / "best_tiny.pth", map_location="cpu", weights_only=True)["nerf_model_state_dict"]
    worst = max(float((a[k] - b[k]).abs().max()) for k in a if k.endswith("weight"))
    # not the same trajectory bit for bit: the stratified jitter is keyed by a ray's position inside the call, which differs
    # between a whole batch and a rank's shard of it; bounded by what Adam can move in 8 steps
    assert worst < 2.5 * 2e-3 * 8


def test_training_survives_moving_the_module_and_loading_weights(N):
    """`.cpu().cuda()` re-allocates every parameter (the flat vector loses them) and load_state_dict overwrites them in
    place: in both cases the next backward must see the current values, forward AND transposed streams."""
    x, g = inputs(2000, seed=71)
    xd, gd = x.cuda(), g.cuda()

    def grads(model):
        for q in model.parameters():
            q.grad = None
        (model(xd) * gd).sum().backward()
        return torch.cat([q.grad.reshape(-1) for q in model.parameters()]).clone()

    ref, _ = make_model(N, "f32", scene="solid")
    want = grads(ref)
    m, _ = make_model(N, "f32", scene="fog")
    grads(m)                                                   # training state built on the fog weights
    m = m.cpu().cuda()                                         # new storages
    m.load_state_dict(O.make_weights("v1", 0, "solid"))        # new values, in place
    got = grads(m.train())
    assert rel_to_max(got, want) < 1e-6
    m.load_state_dict(O.make_weights("v1", 0, "fog"))
    fog, _ = make_model(N, "f32", scene="fog")
    assert rel_to_max(grads(m), grads(fog)) < 1e-6


def test_autograd_route_follows_torch_gradient_conventions(N):
    """What a reference-style loop may do around backward(): accumulate micro-batches, zero gradients in place instead of
    dropping them, clip them -- all on the flat gradient vector behind .grad."""
    x, g = inputs(3000, seed=81)
    xd, gd = x.cuda(), g.cuda()
    model, _ = make_model(N, "f32", scene="solid")

    def flat_grad():
        return torch.cat([q.grad.reshape(-1) for q in model.parameters()]).clone()

    for q in model.parameters():
        q.grad = None
    (model(xd) * gd).sum().backward()
    whole = flat_grad()
    # two micro-batches accumulate (no zero_grad in between)
    for q in model.parameters():
        q.grad = None
    (model(xd[:1000]) * gd[:1000]).sum().backward()
    (model(xd[1000:]) * gd[1000:]).sum().backward()
    assert rel_to_max(flat_grad(), whole) < 1e-5
    # zero_grad(set_to_none=False) keeps the tensors: the next backward adds into zeros
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    opt.zero_grad(set_to_none=False)
    assert float(flat_grad().abs().max()) == 0.0
    (model(xd) * gd).sum().backward()
    assert torch.equal(flat_grad(), whole)                              # deterministic reduction: the same bits
    # clipping scales the flat vector through its views; training.Adam then consumes that very vector
    from nerf_few_shot_limitations_amd.training import Adam
    total = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    assert abs(float(total) - float(whole.norm())) < 1e-3 * float(whole.norm())
    assert abs(float(model._flat_grad.norm()) - 1.0) < 1e-3
    before = model.flat_params().flat.clone()
    Adam(model, lr=1e-3).step()
    moved = (model.flat_params().flat - before).abs()
    assert 0.5e-3 < float(moved.max()) <= 1.01e-3                      # first Adam step: lr * sign(g) where g is not ~0


def test_deepcopy_of_a_trained_module_is_independent(N):
    """copy.deepcopy (EMA / best-model snapshots) must not share the native handle or the flat vectors."""
    import copy, gc
    x, g = inputs(1000, seed=91)
    xd, gd = x.cuda(), g.cuda()
    a, _ = make_model(N, "f32", scene="solid")
    (a(xd) * gd).sum().backward()
    want = torch.cat([q.grad.reshape(-1) for q in a.parameters()]).clone()
    b = copy.deepcopy(a)
    assert b._handle is None and b._flat is None
    with torch.no_grad():
        for q in a.parameters():
            q.mul_(0.5)                                            # a moves on; b keeps the snapshot
    for q in b.parameters():
        q.grad = None
    (b(xd) * gd).sum().backward()
    got = torch.cat([q.grad.reshape(-1) for q in b.parameters()])
    assert rel_to_max(got, want) < 1e-6
    assert a._handle is not None and b._handle is not None and a._handle.value != b._handle.value
    del a, b
    gc.collect()                                                   # two handles, two destroys


def test_fused_step_white_background_and_loss_weight(N):
    """rendering.white_bkgd and loss.rgb_weight of the YAML reach both routes the same way (train.py:236, :41)."""
    from nerf_few_shot_limitations_amd.training import Adam, FusedStep
    R, S, steps = 128, 24, 3
    z = torch.sort(torch.from_numpy(O.uniform01(131, R * S).reshape(R, S) * 4 + 2).float(), dim=-1).values.cuda()
    rd = torch.from_numpy(O.uniform01(132, R * 3).reshape(R, 3) - 0.5).float().cuda()
    tgt = torch.from_numpy(O.uniform01(133, R * 3).reshape(R, 3)).float().cuda()
    pts = torch.from_numpy(O.uniform01(134, R * S * 3).reshape(R * S, 3) * 4 - 2).float().cuda()
    dirs = rd[:, None, :].expand(R, S, 3).reshape(-1, 3).contiguous()
    a, _ = make_v2(N, "f32", scene="fog")
    b, _ = make_v2(N, "f32", scene="fog")
    opt = Adam(a, lr=5e-4)
    vr = N.VolumeRenderer()
    ref = []
    for _ in range(steps):
        opt.zero_grad()
        c, sg = a(pts, dirs, None)
        pred = vr(c.view(R, S, 3), sg.view(R, S, 1), z, rd, white_bkgd=True)[0]
        loss = 0.5 * torch.nn.functional.mse_loss(pred, tgt)
        loss.backward()
        opt.step()
        ref.append(loss.item())
    step = FusedStep(b, lr=5e-4, rgb_weight=0.5, white_bkgd=True)
    got = [step(pts, z, rd, tgt, dirs=dirs).item() for _ in range(steps)]
    assert np.allclose(ref, got, rtol=1e-5, atol=1e-7), (ref, got)
