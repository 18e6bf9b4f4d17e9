"""CPU tests of the training path's host logic (no GPU): the transposed fragment stream replayed through the numpy
model of the MFMA lane maps reproduces the oracle's backward chain, and the weight-gradient plan (saved-tensor slots,
row / column maps, flat parameter layout) turns the oracle's dZ and activations into autograd's parameter gradients."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nerf_oracle as O
from tests import mfma_emulator as E


@pytest.fixture(scope="module")
def L():
    from nerf_few_shot_limitations_amd import _lib
    _lib.lib()
    return _lib


def _linears(L, variant, p, n_layers):
    names = [n for n, _, _ in O.layer_shapes(variant, n_layers=n_layers)]
    arr = (L.nrf_linear * len(names))()
    keep = []
    for i, n in enumerate(names):
        w = np.ascontiguousarray(p[n + ".weight"].numpy()); b = np.ascontiguousarray(p[n + ".bias"].numpy())
        keep += [w, b]
        arr[i] = L.nrf_linear(w.ctypes.data_as(L.c_float_p), b.ctypes.data_as(L.c_float_p), w.shape[0], w.shape[1])
    arch = L.nrf_arch({"v1": 1, "v2": 2}[variant], 10, 4, 256, n_layers, 0)
    return names, arr, arch, keep


def backward_stream(L, variant, p, n_layers, mode="f32"):
    names, arr, arch, keep = _linears(L, variant, p, n_layers)
    ns = C.c_int64()
    L.check(L.lib().nrf_debug_pack_backward(C.byref(arch), arr, len(names), L.MMA_MODES[mode], None, 0, C.byref(ns)))
    raw = (C.c_uint8 * ns.value)()
    L.check(L.lib().nrf_debug_pack_backward(C.byref(arch), arr, len(names), L.MMA_MODES[mode], raw, ns.value, None))
    return bytes(raw)


def train_plan(L, variant, p, n_layers):
    names, arr, arch, keep = _linears(L, variant, p, n_layers)
    n = C.c_int64()
    L.check(L.lib().nrf_debug_train_plan(C.byref(arch), arr, len(names), None, 0, C.byref(n)))
    buf = np.zeros(n.value, np.int32)
    L.check(L.lib().nrf_debug_train_plan(C.byref(arch), arr, len(names), buf.ctypes.data_as(C.c_void_p), n.value, None))
    it = iter(buf.tolist())
    n_slots = next(it)
    slot_tiles = [next(it) for _ in range(n_slots)]
    jobs = []
    for _ in range(next(it)):
        j = dict(x_slot=next(it), dz_slot=next(it), KT=next(it), MT=next(it), x_first=next(it))
        j["row_w"] = [next(it) for _ in range(32 * j["MT"])]
        j["row_b"] = [next(it) for _ in range(32 * j["MT"])]
        j["col"] = [next(it) for _ in range(32 * j["KT"])]
        jobs.append(j)
    n_mask = next(it)
    assert next(it, None) is None
    assert n_mask == (n_layers if variant == "v1" else n_layers + 2)
    return names, slot_tiles, jobs


def tiles(M):
    """(features, 32 samples) -> operand tiles; rows padded to a multiple of 32."""
    rows = (M.shape[0] + 31) // 32 * 32
    X = np.zeros((rows, 32), np.float32)
    X[:M.shape[0]] = M
    return E.tiles_from_matrix(X)


def chain(stream, in_tiles, MT):
    return E.dense(stream, np.zeros(32 * MT, np.float32), in_tiles, MT, "f32")


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_layers", [8, 3, 2])
def test_v1_backward_stream_replay(L, n_layers):
    p = O.make_weights("v1", 0, "solid", n_layers=n_layers)
    x = O.positional_encoding(torch.from_numpy(((O.uniform01(5, 96).reshape(32, 3) * 2 - 1) * 2).astype(np.float32)), 10)
    g = torch.from_numpy((O.uniform01(6, 128).reshape(32, 4) - 0.5).astype(np.float32))
    _, _, acts, dzs = O.mlp_v1_train_emulated(p, x, g, "f32")
    st = E.Stream(backward_stream(L, "v1", p, n_layers), "f32")
    d = tiles(dzs[-1].numpy().T)                                   # head gradient: rows 0..3
    for l in range(n_layers - 1, -1, -1):                          # head^T, then layers l^T: dH of layers.{l}'s output
        dh = E.matrix_from_tiles(chain(st, d, 8))
        dz = dh * (acts[l + 1].numpy().T > 0)
        assert np.abs(dz - dzs[l].numpy().T).max() < 1e-5 * max(1.0, np.abs(dz).max()), l
        d = tiles(dz)
    assert (st.pos + 15) // 16 == st.frags.shape[0] // 16          # the whole stream was consumed (first layer has no dX)


def v2_oracle_backward(p, pos, dirs, g_rgb, g_den, n):
    pe, de = O.positional_encoding(pos, 10), O.positional_encoding(dirs, 4)
    hs = [pe]
    for i in range(n):
        hs.append(F.relu(hs[-1] @ p[f"density_mlp.density_layers.{2 * i}.weight"].T + p[f"density_mlp.density_layers.{2 * i}.bias"]))
    dens_raw = hs[-1] @ p["density_mlp.density_head.weight"].T + p["density_mlp.density_head.bias"]
    feat = hs[-1] @ p["density_mlp.feature_head.weight"].T + p["density_mlp.feature_head.bias"]
    x9 = torch.cat([feat, de], -1)
    c0 = F.relu(x9 @ p["color_mlp.color_layers.0.weight"].T + p["color_mlp.color_layers.0.bias"])
    c1 = F.relu(c0 @ p["color_mlp.color_layers.2.weight"].T + p["color_mlp.color_layers.2.bias"])
    o = torch.sigmoid(c1 @ p["color_mlp.color_layers.4.weight"].T + p["color_mlp.color_layers.4.bias"])
    d4 = g_rgb * o * (1 - o)
    d2 = (d4 @ p["color_mlp.color_layers.4.weight"]) * (c1 > 0)
    d0 = (d2 @ p["color_mlp.color_layers.2.weight"]) * (c0 > 0)
    dfeat = (d0 @ p["color_mlp.color_layers.0.weight"])[:, :256]
    dsig = g_den * (dens_raw > 0)
    dh = dfeat @ p["density_mlp.feature_head.weight"] + dsig @ p["density_mlp.density_head.weight"]
    dz = [None] * n
    for l in range(n - 1, -1, -1):
        dz[l] = dh * (hs[l + 1] > 0)
        if l > 0:
            dh = dz[l] @ p[f"density_mlp.density_layers.{2 * l}.weight"]
    return dict(hs=hs, x9=x9, c0=c0, c1=c1, d4=d4, d2=d2, d0=d0, dfeat=dfeat, dsig=dsig, dz=dz)


def v2_case(n_layers):
    p = O.make_weights("v2", 1, "solid", n_layers=n_layers)
    pos = torch.from_numpy(((O.uniform01(5, 96).reshape(32, 3) * 2 - 1) * 2).astype(np.float32))
    dirs = torch.from_numpy((O.uniform01(6, 96).reshape(32, 3) * 2 - 1).astype(np.float32))
    g_rgb = torch.from_numpy((O.uniform01(7, 96).reshape(32, 3) - 0.5).astype(np.float32))
    g_den = torch.from_numpy((O.uniform01(8, 32).reshape(32, 1) - 0.5).astype(np.float32))
    return p, pos, dirs, g_rgb, g_den


@pytest.mark.parametrize("n_layers", [8, 3])
def test_v2_backward_stream_replay(L, n_layers):
    p, pos, dirs, g_rgb, g_den = v2_case(n_layers)
    o = v2_oracle_backward(p, pos, dirs, g_rgb, g_den, n_layers)
    st = E.Stream(backward_stream(L, "v2", p, n_layers), "f32")
    close = lambda a, b: np.abs(a - b.numpy().T).max() < 1e-5 * max(1.0, np.abs(a).max())
    d2 = E.matrix_from_tiles(chain(st, tiles(o["d4"].numpy().T), 2)) * (o["c1"].numpy().T > 0)
    assert close(d2, o["d2"])
    d0 = E.matrix_from_tiles(chain(st, tiles(d2), 4)) * (o["c0"].numpy().T > 0)
    assert close(d0, o["d0"])
    dfeat = E.matrix_from_tiles(chain(st, tiles(d0), 8))
    assert close(dfeat, o["dfeat"])
    in9 = np.concatenate([tiles(dfeat), tiles(o["dsig"].numpy().T)], 0)            # [d feature_vec (8 tiles) | d sigma (row 0 of tile 8)]
    d = None
    for l in range(n_layers - 1, -1, -1):
        dh = E.matrix_from_tiles(chain(st, in9 if l == n_layers - 1 else d, 8))
        dz = dh * (o["hs"][l + 1].numpy().T > 0)
        assert close(dz, o["dz"][l]), l
        d = tiles(dz)
    assert (st.pos + 15) // 16 == st.frags.shape[0] // 16


# ---------------------------------------------------------------------------------------------
def kernel_order_rows(x_ref, Lf):
    """(samples, 3(2L+1)) reference-order encoding -> (32*KT, samples) rows in the kernel's feature order."""
    KT = (3 * Lf + 2 + 15) // 16
    X = np.zeros((32 * KT, x_ref.shape[0]), np.float32)
    for k in range(32 * KT):
        t, w = k >> 5, k & 31
        h, r = (w >> 2) & 1, (w & 3) | ((w >> 3) << 2)
        u = 16 * t + r
        idx = (3 + 6 * (u // 3) + 3 * h + (u % 3)) if u < 3 * Lf else ((2 if h else 0) if u == 3 * Lf else ((-1 if h else 1) if u == 3 * Lf + 1 else -1))
        if idx >= 0:
            X[k] = x_ref[:, idx]
    return X


def flat_autograd(names, pp):
    return torch.cat([t.grad.reshape(-1) for n in names for t in (pp[n + ".weight"], pp[n + ".bias"])]).numpy()


def run_jobs(slots, jobs, total):
    """What weight_grad_kernel computes, from (features, samples) matrices per slot, scattered through the plan's maps."""
    flat = np.zeros(total, np.float64)
    for j in jobs:
        X = slots[j["x_slot"]][32 * j["x_first"]:32 * (j["x_first"] + j["KT"])]
        Z = slots[j["dz_slot"]][:32 * j["MT"]]
        dW = Z.astype(np.float64) @ X.astype(np.float64).T
        db = Z.astype(np.float64).sum(1)
        for r in range(32 * j["MT"]):
            if j["row_b"][r] >= 0:
                flat[j["row_b"][r]] += db[r]
            if j["row_w"][r] < 0:
                assert np.abs(Z[r]).max() == 0 or j["row_b"][r] < 0          # an unmapped row may only be a duplicate / padding row
                continue
            for k in range(32 * j["KT"]):
                if j["col"][k] >= 0:
                    flat[j["row_w"][r] + j["col"][k]] += dW[r, k]
    return flat


def pad_rows(M, rows):
    out = np.zeros((rows, M.shape[1]), np.float32)
    out[:M.shape[0]] = M
    return out


@pytest.mark.parametrize("n_layers", [8, 2])
def test_v1_weight_gradient_plan(L, n_layers):
    p = O.make_weights("v1", 0, "solid", n_layers=n_layers)
    names, slot_tiles, jobs = train_plan(L, "v1", p, n_layers)
    assert slot_tiles == [2] + [8] * (2 * n_layers) + [1] and len(jobs) == n_layers + 1
    x = O.positional_encoding(torch.from_numpy(((O.uniform01(5, 150).reshape(50, 3) * 2 - 1) * 2).astype(np.float32)), 10)
    g = torch.from_numpy((O.uniform01(6, 200).reshape(50, 4) - 0.5).astype(np.float32))
    pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    (O.mlp_v1(pp, x) * g).sum().backward()
    _, _, acts, dzs = O.mlp_v1_train_emulated(p, x, g, "f32")
    slots = {0: kernel_order_rows(x.numpy(), 10)}
    for l in range(1, n_layers + 1):
        slots[l] = acts[l].numpy().T
        slots[n_layers + l] = dzs[l - 1].numpy().T
    slots[2 * n_layers + 1] = pad_rows(dzs[n_layers].numpy().T, 32)
    ref = flat_autograd(names, pp)
    got = run_jobs(slots, jobs, ref.shape[0])
    assert np.abs(got - ref).max() < 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize("n_layers", [8, 3])
def test_v2_weight_gradient_plan(L, n_layers):
    p, pos, dirs, g_rgb, g_den = v2_case(n_layers)
    names, slot_tiles, jobs = train_plan(L, "v2", p, n_layers)
    n = n_layers
    assert slot_tiles == [2] + [8] * n + [9, 4, 2] + [8] * n + [1, 8, 4, 2, 1] and len(jobs) == n + 6
    o = v2_oracle_backward(p, pos, dirs, g_rgb, g_den, n)
    pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    rgb, den = O.mlp_v2(pp, pos, dirs)
    ((rgb * g_rgb).sum() + (den * g_den).sum()).backward()
    T = lambda t: t.numpy().T
    slots = {0: kernel_order_rows(O.positional_encoding(pos, 10).numpy(), 10)}
    for l in range(1, n + 1):
        slots[l] = T(o["hs"][l])
        slots[n + 3 + l] = T(o["dz"][l - 1])
    slots[n + 1] = np.concatenate([T(o["x9"][:, :256]), kernel_order_rows(O.positional_encoding(dirs, 4).numpy(), 4)], 0)
    slots[n + 2], slots[n + 3] = T(o["c0"]), T(o["c1"])
    slots[2 * n + 4] = pad_rows(T(o["dsig"]), 32)
    slots[2 * n + 5], slots[2 * n + 6], slots[2 * n + 7] = T(o["dfeat"]), T(o["d0"]), T(o["d2"])
    slots[2 * n + 8] = pad_rows(T(o["d4"]), 32)
    ref = flat_autograd(names, pp)
    got = run_jobs(slots, jobs, ref.shape[0])
    assert np.abs(got - ref).max() < 1e-5 * np.abs(ref).max()


def test_training_entry_points_validate_arguments(L):
    lib = L.lib()
    assert lib.nrf_param_count(None) == 0
    assert lib.nrf_model_update_device(None, None, 1, None) == -1 and b"null" in lib.nrf_last_error()
    assert lib.nrf_mlp_forward_train_v1(None, 0, None, 4, None, None, 0, None) == -1
    assert lib.nrf_mlp_backward(None, 0, None, None, None, None, 4, None, 0, None, None) == -1
    assert lib.nrf_mlp_forward_train(None, 0, None, None, None, 4, None, None, None, 0, None) == -1
    assert lib.nrf_adam_step(None, None, None, None, 4, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None) == -1
    assert lib.nrf_adam_step(None, None, None, None, 4, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, None) == -1      # steps count from 1
    assert lib.nrf_adam_step(None, None, None, None, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None) == 0       # nothing to do
    assert lib.nrf_composite_backward(None, 3, None, 1, None, None, 5, 8, 0, None, None, None, None, 3, None, 1, None) == -1
    assert lib.nrf_composite_backward(None, 2, None, 1, None, None, 5, 8, 0, None, None, None, None, 3, None, 1, None) == -1
    p = O.make_weights("v3", 2)
    names = [n for n, _, _ in O.layer_shapes("v3")]
    arr = (L.nrf_linear * len(names))()
    keep = []
    for i, nme in enumerate(names):
        w = np.ascontiguousarray(p[nme + ".weight"].numpy()); b = np.ascontiguousarray(p[nme + ".bias"].numpy())
        keep += [w, b]
        arr[i] = L.nrf_linear(w.ctypes.data_as(L.c_float_p), b.ctypes.data_as(L.c_float_p), w.shape[0], w.shape[1])
    arch = L.nrf_arch(3, 12, 4, 256, 8, 64)
    n = C.c_int64()
    assert lib.nrf_debug_train_plan(C.byref(arch), arr, len(names), None, 0, C.byref(n)) == 0
    buf = np.zeros(n.value, np.int32)
    assert lib.nrf_debug_train_plan(C.byref(arch), arr, len(names), buf.ctypes.data_as(C.c_void_p), n.value, None) == 0
    n_slots = int(buf[0])
    assert n_slots == 23 + 2 * 8 and list(buf[1:1 + n_slots][[0, 3, 4]]) == [5, 2, 5]        # [pe(3)|dino(2)] tiles, attention.0, scaled inputs
    assert int(buf[1 + n_slots]) == 8 + 13                                                  # jobs: both fusion layers twice
    assert int(buf[-1]) == 7 + 8                                                            # ReLU bit planes
    arch_bad = L.nrf_arch(7, 12, 4, 256, 8, 64)
    assert lib.nrf_debug_train_plan(C.byref(arch_bad), arr, len(names), None, 0, C.byref(n)) < 0
