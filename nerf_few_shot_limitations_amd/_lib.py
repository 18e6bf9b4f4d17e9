"""ctypes binding of libnerfhip.so (include/nerfhip.h).

The library is the product: there is no Python / PyTorch fallback.  If it is
missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch  # imported BEFORE the library on purpose: libnerfhip.so then binds to the HIP runtime torch already loaded

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NRF_LIB") or os.path.join(_PKG, "libnerfhip.so")     # NRF_LIB: A/B another build of the same ABI

NRF_NET_V1, NRF_NET_V2, NRF_NET_V3 = 1, 2, 3
MMA_MODES = {"bf16": 0, "f16": 1, "f32": 2, "f16x3": 3}
# the training kernels are built for the first three; a module in the split mode (fp32-class results) trains in exact fp32
TRAIN_MODE = {"bf16": "bf16", "f16": "f16", "f32": "f32", "f16x3": "f32"}
ERRORS = {-1: "NRF_EINVAL", -2: "NRF_EUNSUPPORTED", -3: "NRF_EHIP", -4: "NRF_ENOMEM"}

c_float_p = C.POINTER(C.c_float)


class NrfError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libnerfhip: {ERRORS.get(code, code)}: {msg}")
        self.code = code


class nrf_arch(C.Structure):
    _fields_ = [("net", C.c_int32), ("pos_freq", C.c_int32), ("dir_freq", C.c_int32), ("hidden", C.c_int32),
                ("n_layers", C.c_int32), ("dino_dim", C.c_int32)]


class nrf_linear(C.Structure):
    _fields_ = [("weight", c_float_p), ("bias", c_float_p), ("out_f", C.c_int32), ("in_f", C.c_int32)]


class nrf_dino(C.Structure):
    _fields_ = [("features", C.c_void_p), ("Hp", C.c_int32), ("Wp", C.c_int32), ("C", C.c_int32),
                ("inv_pose", C.c_float * 16), ("focal", C.c_float), ("H", C.c_int32), ("W", C.c_int32)]


class nrf_render_opts(C.Structure):
    _fields_ = [("near", C.c_float), ("far", C.c_float), ("n_samples", C.c_int32), ("lindisp", C.c_int32),
                ("perturb", C.c_int32), ("t_rand", C.c_void_p), ("z_ladder", C.c_void_p), ("z_in", C.c_void_p), ("rng_seed", C.c_uint64), ("ert_eps", C.c_float),
                ("white_bkgd", C.c_int32), ("mma_mode", C.c_int32), ("dino", C.POINTER(nrf_dino)), ("out_rgbd", C.c_int32)]


# name -> (restype, argtypes); tests/test_packing_emulation.py checks this table against include/nerfhip.h
SIGNATURES = {
    "nrf_abi_version": (C.c_int, []),
    "nrf_last_error": (C.c_char_p, []),
    "nrf_abi_sizeof": (C.c_int, [C.c_int]),
    "nrf_model_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(nrf_arch), C.POINTER(nrf_linear), C.c_int]),
    "nrf_model_update": (C.c_int, [C.c_void_p, C.POINTER(nrf_linear), C.c_int, C.c_void_p]),
    "nrf_model_destroy": (None, [C.c_void_p]),
    "nrf_model_flops_per_sample": (C.c_int64, [C.c_void_p]),
    "nrf_render_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(nrf_render_opts),
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_render_camera": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float * 12, C.c_int64, C.c_int64,
                                    C.POINTER(nrf_render_opts), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_render_cameras_tiles": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64,
                                           C.c_int64, C.POINTER(nrf_render_opts), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_get_rays": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_float * 12, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_sample_along_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_encode": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_mlp_forward_v1": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "nrf_mlp_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_composite": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_sample_pdf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_debug_pack": (C.c_int, [C.POINTER(nrf_arch), C.POINTER(nrf_linear), C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                 C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "nrf_project_fetch": (C.c_int, [C.POINTER(nrf_dino), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_sample_features": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "nrf_debug_pack_backward": (C.c_int, [C.POINTER(nrf_arch), C.POINTER(nrf_linear), C.c_int, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "nrf_debug_train_plan": (C.c_int, [C.POINTER(nrf_arch), C.POINTER(nrf_linear), C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    # training path
    "nrf_param_count": (C.c_int64, [C.c_void_p]),
    "nrf_model_update_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "nrf_train_context_bytes": (C.c_int64, [C.c_void_p, C.c_int, C.c_int64]),
    "nrf_mlp_forward_train_v1": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "nrf_mlp_backward_v1": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "nrf_mlp_forward_train": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_void_p]),
    "nrf_mlp_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                   C.c_void_p, C.c_void_p]),
    "nrf_composite_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "nrf_mse_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_composite_mse_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                             C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_int64, C.c_void_p]),
    "nrf_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_float, C.c_int, C.c_void_p]),
    "nrf_adam_step_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_int, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
}

_lib = None
_lock = threading.Lock()


def lib() -> C.CDLL:
    """Load libnerfhip.so once.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m nerf_few_shot_limitations_amd.build` "
                                   "(there is no fallback path)")
            handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(handle, name)          # AttributeError if the symbol is not exported
                fn.restype, fn.argtypes = res, args
            if handle.nrf_abi_version() != 5:
                raise RuntimeError("libnerfhip.so ABI version mismatch")
            for which, st in enumerate((nrf_arch, nrf_linear, nrf_dino, nrf_render_opts)):
                if handle.nrf_abi_sizeof(which) != C.sizeof(st):
                    raise RuntimeError(f"libnerfhip.so: sizeof({st.__name__}) differs from the ctypes declaration")
            _lib = handle
    return _lib


def check(code: int) -> None:
    if code != 0:
        raise NrfError(code, lib().nrf_last_error().decode("utf-8", "replace"))


def ptr(t):
    """Device pointer of a CUDA(HIP) fp32 contiguous tensor (an int: ctypes converts it for a c_void_p parameter), or None."""
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError("expected a contiguous float32 tensor on the GPU")
    return t.data_ptr()


def dev_f32(t, device=None):
    """Bring a tensor-like to contiguous fp32 on the GPU (the reference's callers hand over fp32 tensors)."""
    if (type(t) is torch.Tensor and t.is_cuda and t.dtype == torch.float32 and not t.requires_grad and t.is_contiguous()
            and (device is None or t.device == device)):
        return t                                    # the common case on the training loop's host-bound path: nothing to do
    t = torch.as_tensor(t)
    if device is None:
        device = t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device())
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def refuse_grad(t, what):
    """The staged leaf kernels produce no gradient with respect to their inputs (the reference never needs one: positions,
    directions, rays and depths are data).  Under grad mode a tensor that requires grad is refused -- as NeRFMLP.forward refuses
    DINO features that require grad -- instead of being detached silently."""
    if torch.is_grad_enabled() and isinstance(t, torch.Tensor) and t.requires_grad:
        raise NotImplementedError(f"{what}: no gradient with respect to this input is produced; pass it detached or call under torch.no_grad()")
    return t


def fresh_seed():
    """A new seed for the in-kernel jitter RNG, drawn from torch's default CPU generator: a fresh pattern on every call (the
    reference draws torch.rand per call, ray_utils.py:78) that repeats under torch.manual_seed."""
    return int(torch.randint(0, 2 ** 62, (), dtype=torch.int64).item())


def stream_ptr():
    """The current device's current stream as the `void* stream` argument (torch's raw-stream accessor: no Stream object per call)."""
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


_gpu_seen = False


def require_gpu():
    global _gpu_seen
    if _gpu_seen:
        return
    if not torch.cuda.is_available():
        raise RuntimeError("nerf_few_shot_limitations_amd needs an MI355X (gfx950) GPU: no HIP device is visible and there is no CPU path")
    _gpu_seen = True


_ladders = {}
_u_rows = {}


def u_row(n_importance, device):
    """torch.linspace(0, 1, Ni) exactly as the reference computes it on THIS host (ray_utils.py:115-116), on `device`: the
    un-perturbed inverse-cdf arguments shared by all rays (nrf_sample_pdf with u_ray_stride = 0)."""
    key = (int(n_importance), str(device))
    u = _u_rows.get(key)
    if u is None:
        u = torch.linspace(0., 1., int(n_importance)).to(dtype=torch.float32).to(device).contiguous()
        if len(_u_rows) > 64:
            _u_rows.clear()
        _u_rows[key] = u
    return u



def z_ladder(near, far, n_samples, lindisp, device):
    """The un-jittered depth ladder exactly as the reference computes it on THIS host
    (src/utils/ray_utils.py:58-66: torch.linspace on the CPU, then near*(1-t)+far*t), on `device`."""
    key = (float(near), float(far), int(n_samples), bool(lindisp), str(device))
    z = _ladders.get(key)
    if z is None:
        t = torch.linspace(0., 1., int(n_samples))
        z = 1. / (1. / near * (1. - t) + 1. / far * t) if lindisp else near * (1. - t) + far * t
        z = z.to(dtype=torch.float32).to(device).contiguous()
        if len(_ladders) > 64:
            _ladders.clear()
        _ladders[key] = z
    return z
