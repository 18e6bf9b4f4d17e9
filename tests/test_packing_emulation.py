"""CPU tests of the host logic: the C-ABI library loads and exports every symbol the header
declares, argument validation fails loudly, and the packed weight stream -- replayed through a
numpy model of the MFMA lane maps (tests/mfma_emulator.py) -- reproduces the oracle's MLP."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from tests import mfma_emulator as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from nerf_few_shot_limitations_amd import _lib
    _lib.lib()
    return _lib


def test_library_exports_every_declared_symbol(L):
    header = open(os.path.join(ROOT, "include", "nerfhip.h")).read()
    declared = set(re.findall(r"\b(nrf_[a-z0-9_]+)\s*\(", header))
    declared -= {"nrf_model"}
    assert declared, "no declarations parsed"
    lib = L.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in nerfhip.h but not exported"
    assert declared == set(L.SIGNATURES), (declared ^ set(L.SIGNATURES))
    assert lib.nrf_abi_version() == 5
    import ctypes
    for which, st in enumerate((L.nrf_arch, L.nrf_linear, L.nrf_dino, L.nrf_render_opts)):
        assert lib.nrf_abi_sizeof(which) == ctypes.sizeof(st)
    assert lib.nrf_abi_sizeof(99) == -1


def _pack(L, variant, p, mode, **arch_kw):
    names = [n for n, _, _ in O.layer_shapes(variant)]
    arr = (L.nrf_linear * len(names))()
    keep = []
    for i, n in enumerate(names):
        w = np.ascontiguousarray(p[n + ".weight"].numpy()); b = np.ascontiguousarray(p[n + ".bias"].numpy())
        keep += [w, b]
        arr[i] = L.nrf_linear(w.ctypes.data_as(L.c_float_p), b.ctypes.data_as(L.c_float_p), w.shape[0], w.shape[1])
    net = {"v1": 1, "v2": 2, "v3": 3}[variant]
    arch = L.nrf_arch(net, arch_kw.get("pos_freq", 12 if variant == "v3" else 10), 4, 256, arch_kw.get("n_layers", 8), 64 if variant == "v3" else 0)
    nb, ns = C.c_int64(), C.c_int64()
    L.check(L.lib().nrf_debug_pack(C.byref(arch), arr, len(names), L.MMA_MODES[mode], None, 0, C.byref(ns), None, 0, C.byref(nb)))
    raw = (C.c_uint8 * ns.value)()
    bias = np.zeros(nb.value, np.float32)
    L.check(L.lib().nrf_debug_pack(C.byref(arch), arr, len(names), L.MMA_MODES[mode], raw, ns.value, None,
                                   bias.ctypes.data_as(C.c_void_p), nb.value, None))
    return bytes(raw), bias


def _encode_tiles(x, Lf):
    """positions (32,3) -> operand tiles in the kernel's feature order (feature_map.hpp), via the oracle's encoding."""
    enc = O.positional_encoding(torch.from_numpy(x), Lf).numpy()          # (32, 3(2L+1)) reference order
    KT = (3 * Lf + 2 + 15) // 16
    X = np.zeros((32 * KT, 32), np.float32)
    for k in range(32 * KT):
        t, w = k >> 5, k & 31
        h, r = (w >> 2) & 1, (w & 3) | ((w >> 3) << 2)
        u = 16 * t + r
        if u < 3 * Lf:
            idx = 3 + 6 * (u // 3) + 3 * h + (u % 3)
        elif u == 3 * Lf:
            idx = 2 if h else 0
        elif u == 3 * Lf + 1:
            idx = -1 if h else 1
        else:
            idx = -1
        if idx >= 0:
            X[k] = enc[:, idx]
    return E.tiles_from_matrix(X)


@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("f16x3", 2e-5), ("f16", 3e-3), ("bf16", 3e-2)])
def test_v1_stream_replay_matches_oracle(L, mode, tol):
    p = O.make_weights("v1", 0)
    raw, bias = _pack(L, "v1", p, mode)
    assert len(raw) % (16 * 1024) == 0
    st = E.Stream(raw, mode)
    x = ((O.uniform01(5, 96).reshape(32, 3) * 2 - 1) * 3).astype(np.float32)
    act = E.quantize(_encode_tiles(x, 10), mode)
    boff = 0
    for layer in range(8):
        acc = E.dense(st, bias[boff:boff + 256], act, 8, mode)
        act = E.quantize(np.maximum(acc, 0), mode)
        boff += 256
    head = E.dense(st, bias[boff:boff + 32], act, 1, mode)
    assert st.pos <= st.frags.shape[0] and (st.pos + 15) // 16 == st.frags.shape[0] // 16
    got = np.stack([head[0, :32, k] for k in range(4)], -1)               # lanes 0..31, regs 0..3 = rows 0..3
    got_hi = np.stack([head[0, 32:, k] for k in range(4)], -1)            # lanes 32..63 see the duplicated rows 4..7
    ref = O.mlp_v1(p, O.positional_encoding(torch.from_numpy(x), 10)).numpy()
    ref_raw = ref.copy()
    ref_raw[:, :3] = np.log(ref[:, :3] / (1 - ref[:, :3]))                # undo the sigmoid: the head tile holds logits
    scale = max(1.0, float(np.abs(ref_raw).max()))
    assert np.abs(got - ref_raw).max() < tol * scale, (np.abs(got - ref_raw).max(), scale)
    assert np.array_equal(got, got_hi)


@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("f16x3", 2e-5), ("bf16", 3e-2)])
def test_v2_stream_replay_matches_oracle(L, mode, tol):
    p = O.make_weights("v2", 1)
    raw, bias = _pack(L, "v2", p, mode)
    st = E.Stream(raw, mode)
    x = ((O.uniform01(6, 96).reshape(32, 3) * 2 - 1) * 3).astype(np.float32)
    d = (O.uniform01(7, 96).reshape(32, 3) * 2 - 1).astype(np.float32)
    act = E.quantize(_encode_tiles(x, 10), mode)
    dirt = E.quantize(_encode_tiles(d, 4), mode)
    boff = 0
    for layer in range(8):
        acc = E.dense(st, bias[boff:boff + 256], act, 8, mode)
        act = E.quantize(np.maximum(acc, 0), mode)
        boff += 256
    dens = E.dense(st, bias[boff:boff + 32], act, 1, mode); boff += 32
    feat = E.quantize(E.dense(st, bias[boff:boff + 256], act, 8, mode), mode); boff += 256
    in9 = np.concatenate([feat, dirt], -3)                                  # tile axis (the split mode's images carry a leading hi/lo axis)
    c0 = E.quantize(np.maximum(E.dense(st, bias[boff:boff + 128], in9, 4, mode), 0), mode); boff += 128
    c1 = E.quantize(np.maximum(E.dense(st, bias[boff:boff + 64], c0, 2, mode), 0), mode); boff += 64
    rgb = E.dense(st, bias[boff:boff + 32], c1, 1, mode); boff += 32
    assert boff == bias.shape[0]
    assert st.pos <= st.frags.shape[0] and (st.pos + 15) // 16 == st.frags.shape[0] // 16
    ref_rgb, ref_dens = O.mlp_v2(p, torch.from_numpy(x), torch.from_numpy(d))
    got_rgb = 1 / (1 + np.exp(-np.stack([rgb[0, :32, k] for k in range(3)], -1)))
    assert np.abs(got_rgb - ref_rgb.numpy()).max() < tol
    assert np.abs(np.maximum(dens[0, :32, 0], 0) - ref_dens.numpy()[:, 0]).max() < tol * max(1.0, float(ref_dens.max()))
    assert np.array_equal(dens[0, :32, 0], dens[0, 32:, 0])               # both lane halves see the density


def test_v3_stream_replay_matches_oracle(L):
    """NeRFWithDINO: the fusion block's weights appear twice in the stream; the softmax gate rescales the inputs."""
    mode, tol = "f32", 5e-5
    p = O.make_weights("v3", 2)
    raw, bias = _pack(L, "v3", p, mode)
    st = E.Stream(raw, mode)
    x = ((O.uniform01(6, 96).reshape(32, 3) * 2 - 1) * 3).astype(np.float32)
    d = (O.uniform01(7, 96).reshape(32, 3) * 2 - 1).astype(np.float32)
    dino = (O.uniform01(8, 32 * 64).reshape(32, 64) * 2 - 1).astype(np.float32)
    dirt = _encode_tiles(d, 4)

    def inputs(w0, w1):
        pe = _encode_tiles(x, 12)                                            # 3 tiles
        pe_m = E.matrix_from_tiles(pe) * w0[None, :]
        dn = dino.T * w1[None, :]                                            # (64 channels, 32 samples): K index = channel
        return E.tiles_from_matrix(np.concatenate([pe_m, dn], 0).astype(np.float32))

    off = [0]

    def layer(act, MT, relu=True):
        acc = E.dense(st, bias[off[0]:off[0] + 32 * MT], act, MT, mode)
        off[0] += 32 * MT
        return np.maximum(acc, 0) if relu else acc

    one = np.ones(32, np.float32)
    f = layer(layer(inputs(one, one), 8), 8)
    a0 = layer(f, 2)
    lg = layer(a0, 1, relu=False)
    l0, l1 = lg[0, :32, 0], lg[0, :32, 1]
    assert np.array_equal(l0, lg[0, 32:, 0]) and np.array_equal(l1, lg[0, 32:, 1])
    w0 = 1 / (1 + np.exp(l1 - l0)); w1 = 1 - w0
    f2 = layer(layer(inputs(w0.astype(np.float32), w1.astype(np.float32)), 8), 8)
    hcur = layer(f2, 8, relu=False)                                          # output_proj
    fused_ref = O.dino_fusion(p, "dino_fusion.", O.positional_encoding(torch.from_numpy(x), 12), torch.from_numpy(dino)).numpy()
    assert np.abs(E.matrix_from_tiles(hcur).T - fused_ref).max() < tol * max(1.0, np.abs(fused_ref).max())
    for _ in range(8):
        hcur = layer(hcur, 8)
    dens = layer(hcur, 1, relu=False)
    feat = layer(hcur, 8, relu=False)
    c0 = layer(np.concatenate([feat, dirt], 0), 4)
    c1 = layer(c0, 2)
    rgb = layer(c1, 1, relu=False)
    assert off[0] == bias.shape[0] and (st.pos + 15) // 16 == st.frags.shape[0] // 16
    ref_rgb, ref_dens = O.mlp_v3(p, torch.from_numpy(x), torch.from_numpy(d), torch.from_numpy(dino))
    got_rgb = 1 / (1 + np.exp(-np.stack([rgb[0, :32, k] for k in range(3)], -1)))
    assert np.abs(got_rgb - ref_rgb.numpy()).max() < 1e-4
    assert np.abs(np.maximum(dens[0, :32, 0], 0) - ref_dens.numpy()[:, 0]).max() < 1e-4 * max(1.0, float(ref_dens.max()))


def test_model_create_validates_shapes(L):
    p = O.make_weights("v1", 0)
    names = [n for n, _, _ in O.layer_shapes("v1")]
    arr = (L.nrf_linear * len(names))()
    keep = []
    for i, n in enumerate(names):
        w = np.ascontiguousarray(p[n + ".weight"].numpy()); b = np.ascontiguousarray(p[n + ".bias"].numpy())
        keep += [w, b]
        arr[i] = L.nrf_linear(w.ctypes.data_as(L.c_float_p), b.ctypes.data_as(L.c_float_p), w.shape[0], w.shape[1])
    ns = C.c_int64()
    bad = L.nrf_arch(1, 10, 4, 128, 8, 0)                                  # hidden=128 is not built
    assert L.lib().nrf_debug_pack(C.byref(bad), arr, len(names), 0, None, 0, C.byref(ns), None, 0, None) == -1
    assert b"hidden" in L.lib().nrf_last_error()
    bad = L.nrf_arch(1, 10, 4, 256, 7, 0)                                  # wrong layer count
    assert L.lib().nrf_debug_pack(C.byref(bad), arr, len(names), 0, None, 0, C.byref(ns), None, 0, None) == -1
    ok = L.nrf_arch(1, 10, 4, 256, 8, 0)
    assert L.lib().nrf_debug_pack(C.byref(ok), arr, len(names), 9, None, 0, C.byref(ns), None, 0, None) == -1


def test_product_path_fails_loudly_without_gpu(L):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import nerf_few_shot_limitations_amd as N
    with pytest.raises(RuntimeError, match="no CPU path"):
        N.get_rays(4, 4, 10.0, torch.eye(4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        N.PositionalEncoding(4)(torch.zeros(2, 3))


def test_config_yaml_near_far_resolution(tmp_path):
    import nerf_few_shot_limitations_amd as N
    y = tmp_path / "c.yaml"
    y.write_text("data: {near: 2.0, far: 6.0}\nrendering: {chunk_size: 2048}\nmodel: {use_dino: false}\n"
                 "nerf_model: {pos_freq: 10, dir_freq: 4, hidden_dim: 256, num_layers: 8}\n"
                 "training: {progressive_schedule: {epochs_100_plus: [128, 128, 64]}}\n")
    cfg = N.load_config(str(y))
    assert N.resolve_near_far(cfg) == (2.0, 6.0)                           # D3: falls through to data:
    cfg["near"], cfg["far"] = 1.0, 5.0
    assert N.resolve_near_far(cfg) == (1.0, 5.0)                           # top level wins (train.py:192-193)
    m = N.model_from_config(cfg)
    assert m.flops_per_sample() == 1170560
    assert N.render_settings(cfg)["n_samples"] == 64
    assert set(m.state_dict()) >= {"density_mlp.density_layers.0.weight", "density_mlp.feature_head.bias", "color_mlp.color_layers.4.weight"}


def test_out_rgbd_rows_must_be_16_byte_aligned(L):
    """the kernels write an out_rgbd row with one 16-byte store; a float-aligned base is refused with NRF_EINVAL
    before anything else is looked at (no GPU needed: the check precedes the model check)."""
    lib = L.lib()
    o = L.nrf_render_opts()
    o.near, o.far, o.n_samples, o.out_rgbd = 2.0, 6.0, 8, 1
    c2w = (C.c_float * 12)(*([0.0] * 12))
    for bad in (0x1004, 0x1008, 0x100C):
        assert lib.nrf_render_rays(None, C.c_void_p(0x2000), C.c_void_p(0x3000), 4, C.byref(o), C.c_void_p(bad), None, None, None, None) == -1
        assert b"16-byte aligned" in lib.nrf_last_error()
        assert lib.nrf_render_camera(None, 4, 4, 1.0, c2w, 0, 16, C.byref(o), C.c_void_p(bad), None, None, None, None) == -1
        assert b"16-byte aligned" in lib.nrf_last_error()
        assert lib.nrf_render_cameras_tiles(None, 4, 4, 1.0, C.cast(c2w, C.c_void_p), 1, 4, 0, 1, 1, C.byref(o), C.c_void_p(bad), None, None, None, None) == -1
        assert b"16-byte aligned" in lib.nrf_last_error()
    # aligned: the next check answers (model is NULL)
    assert lib.nrf_render_rays(None, C.c_void_p(0x2000), C.c_void_p(0x3000), 4, C.byref(o), C.c_void_p(0x1000), None, None, None, None) == -1
    assert b"model is NULL" in lib.nrf_last_error()
    o.out_rgbd = 0                                               # separate outputs: no alignment requirement
    assert lib.nrf_render_rays(None, C.c_void_p(0x2000), C.c_void_p(0x3000), 4, C.byref(o), C.c_void_p(0x1004), None, None, None, None) == -1
    assert b"model is NULL" in lib.nrf_last_error()


def test_fused_step_entry_points_refuse_bad_arguments_before_any_launch(L):
    """nrf_composite_mse_backward / nrf_adam_step_loss (ABI v5): sizes, strides and the pointers they need are checked on the host
    (no GPU needed: nothing is launched)."""
    lib = L.lib()
    P = lambda a: C.c_void_p(a)
    ok = dict(rgb=P(0x1000), rs=4, sig=P(0x100C), ss=4, z=P(0x2000), d=P(0x3000), R=8, S=16, white=0, tgt=P(0x4000), w=1.0, pred=None,
              drgb=P(0x5000), drs=4, dsig=P(0x500C), dss=4, rl=P(0x6000), zb=None, zn=0)

    def call(**kw):
        a = dict(ok, **kw)
        return lib.nrf_composite_mse_backward(a["rgb"], a["rs"], a["sig"], a["ss"], a["z"], a["d"], a["R"], a["S"], a["white"], a["tgt"], a["w"],
                                              a["pred"], a["drgb"], a["drs"], a["dsig"], a["dss"], a["rl"], a["zb"], a["zn"], None)
    for bad in (dict(R=0), dict(R=-3), dict(S=0), dict(S=5000), dict(rs=2), dict(ss=0), dict(drs=1), dict(dss=0), dict(rgb=None), dict(sig=None),
                dict(z=None), dict(d=None), dict(tgt=None), dict(drgb=None), dict(dsig=None), dict(rl=None), dict(zn=-1), dict(zn=4)):
        assert call(**bad) == -1, bad
        assert lib.nrf_last_error()
    adam = lambda **kw: lib.nrf_adam_step_loss(kw.get("p", P(0x1000)), kw.get("g", P(0x2000)), kw.get("m", P(0x3000)), kw.get("v", P(0x4000)),
                                               kw.get("n", 100), 1e-3, kw.get("b1", 0.9), 0.999, 1e-8, 0.0, kw.get("step", 1), kw.get("rl", P(0x5000)),
                                               kw.get("R", 8), 1.0, kw.get("loss", P(0x6000)), None)
    for bad in (dict(n=0), dict(step=0), dict(p=None), dict(g=None), dict(m=None), dict(v=None), dict(b1=1.0), dict(rl=None), dict(loss=None), dict(R=0)):
        assert adam(**bad) == -1, bad


@pytest.mark.parametrize("mode", ["f16", "f16x3"])
def test_f16_typed_streams_saturate_instead_of_overflowing(L, mode):
    """a weight beyond the f16 range used to pack as inf (split mode: hi = inf, lo = -inf -> NaN products).  The packer
    saturates at +-65504; in-range weights pack exactly as before."""
    p = O.make_weights("v1", 0, "fog")
    ref, _ = _pack(L, "v1", p, mode)
    q = {k: v.clone() for k, v in p.items()}
    q["layers.3.weight"][5, 7] = 1.0e6
    q["layers.3.weight"][6, 8] = -3.0e5
    got, _ = _pack(L, "v1", q, mode)
    a = np.frombuffer(got, np.float16)
    assert np.isfinite(a.astype(np.float32)).all()
    assert (a == np.float16(65504)).sum() >= 1 and (a == np.float16(-65504)).sum() >= 1
    changed = np.frombuffer(got, np.uint16) != np.frombuffer(ref, np.uint16)
    assert 2 <= changed.sum() <= 4                               # only the two edited weights (hi and lo parts in the split mode)
