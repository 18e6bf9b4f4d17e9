"""NeRFMLP on the GPU: an nn.Module that owns the reference's parameters (same
state_dict names) and evaluates them with libnerfhip's MFMA kernels.

Two constructor forms, as the reference uses them:
  * legacy  NeRFMLP(pos_dim=63, hidden_dim=256, n_layers=8)(x_enc (P,63)) -> (P,4)
            src/models/nerf_model.py:5-24, src/training/train_minimal.py:28,102
  * trainer NeRFMLP(pos_freq=, dir_freq=, hidden_dim=, num_density_layers=, use_dino=, dino_dim=)
            (positions (P,3), directions (P,3), dino (P,C)|None) -> (rgb (P,3), density (P,1))
            src/training/train.py:82-89,229; the module tree mirrors
            src/models/nerf_mlp.py:86-158 (NeRFWithDINO) so checkpoints load by name.
With grad enabled every form runs the training kernels (training.py: saved activations,
transposed-stream backward, MFMA weight gradients); gradients reach the parameters only -- a
dino_features tensor that requires grad is refused instead of silently detached.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L


def _stack(n_in, hidden, n_layers):
    mods = []
    for i in range(n_layers):
        mods += [nn.Linear(n_in if i == 0 else hidden, hidden), nn.ReLU(inplace=True)]
    return nn.Sequential(*mods)


class _DensityMLP(nn.Module):          # parameter container, names of nerf_mlp.py:41-58
    def __init__(self, input_dim, hidden_dim, num_layers):
        super().__init__()
        self.density_layers = _stack(input_dim, hidden_dim, num_layers)
        self.density_head = nn.Linear(hidden_dim, 1)
        self.feature_head = nn.Linear(hidden_dim, hidden_dim)


class _ColorMLP(nn.Module):            # nerf_mlp.py:68-80
    def __init__(self, feature_dim, dir_dim, hidden_dim):
        super().__init__()
        self.color_layers = nn.Sequential(nn.Linear(feature_dim + dir_dim, hidden_dim), nn.ReLU(inplace=True),
                                          nn.Linear(hidden_dim, hidden_dim // 2), nn.ReLU(inplace=True),
                                          nn.Linear(hidden_dim // 2, 3), nn.Sigmoid())


class _Fusion(nn.Module):              # lora_dino.py:146-169
    def __init__(self, pos_dim, dino_dim, hidden_dim):
        super().__init__()
        self.fusion = nn.Sequential(nn.Linear(pos_dim + dino_dim, hidden_dim), nn.ReLU(inplace=True),
                                    nn.Linear(hidden_dim, hidden_dim), nn.ReLU(inplace=True))
        self.attention = nn.Sequential(nn.Linear(hidden_dim, hidden_dim // 4), nn.ReLU(inplace=True),
                                       nn.Linear(hidden_dim // 4, 2), nn.Softmax(dim=-1))
        self.output_proj = nn.Linear(hidden_dim, hidden_dim)


class _FreqBuffer(nn.Module):          # nerf_mlp.py:14-15 registers freq_bands as a buffer
    def __init__(self, num_freqs):
        super().__init__()
        self.num_freqs = num_freqs
        self.register_buffer("freq_bands", 2.0 ** torch.linspace(0.0, num_freqs - 1, num_freqs))

    def get_output_dim(self, input_dim):
        return input_dim * (2 * self.num_freqs + 1)


class NeRFMLP(nn.Module):
    def __init__(self, pos_dim=63, hidden_dim=256, n_layers=8, *, pos_freq=None, dir_freq=4, num_density_layers=None,
                 use_dino=False, dino_dim=0, mma_mode="f32"):
        super().__init__()
        self.mma_mode = mma_mode
        self.hidden_dim = int(hidden_dim)
        if pos_freq is None:
            # legacy form: input is already encoded
            if (pos_dim - 3) % 6 != 0:
                raise ValueError("pos_dim must be 3*(2L+1)")
            self.net = L.NRF_NET_V1
            self.pos_freq, self.dir_freq, self.dino_dim = (pos_dim - 3) // 6, 0, 0
            self.n_layers = int(n_layers)
            self.layers = nn.ModuleList([nn.Linear(pos_dim if i == 0 else hidden_dim, hidden_dim) for i in range(n_layers)])
            self.sigma_out = nn.Linear(hidden_dim, 1)
            self.rgb_out = nn.Linear(hidden_dim, 3)
        else:
            self.pos_freq, self.dir_freq = int(pos_freq), int(dir_freq)
            self.n_layers = int(num_density_layers if num_density_layers is not None else n_layers)
            self.pos_encoder, self.dir_encoder = _FreqBuffer(self.pos_freq), _FreqBuffer(self.dir_freq)
            pe, de = self.pos_encoder.get_output_dim(3), self.dir_encoder.get_output_dim(3)
            if use_dino:
                self.net, self.dino_dim = L.NRF_NET_V3, int(dino_dim)
                self.dino_fusion = _Fusion(pe, self.dino_dim, hidden_dim)
                d_in = hidden_dim
            else:
                self.net, self.dino_dim = L.NRF_NET_V2, 0
                d_in = pe
            self.density_mlp = _DensityMLP(d_in, hidden_dim, self.n_layers)
            self.color_mlp = _ColorMLP(hidden_dim, de, hidden_dim // 2)
        self._handle = None
        self._handle_dev = None
        self._packed = None          # parameter versions the packed streams were built from
        self._packed_modes = None    # modes whose streams match _packed (None = all three)
        self._gen = 0                # bumped by optimizers that update the flat vector behind autograd's back
        self._flat = None
        self._train_ready = False

    # ---- parameters in the order include/nerfhip.h documents ------------------------------------
    def linears(self):
        cached = self.__dict__.get("_linears_cache")        # the module tree is fixed after __init__; this list is asked for several times per step
        if cached is not None:
            return cached
        out = self._build_linears()
        self.__dict__["_linears_cache"] = out
        return out

    def _build_linears(self):
        if self.net == L.NRF_NET_V1:
            return list(self.layers) + [self.sigma_out, self.rgb_out]
        out = []
        if self.net == L.NRF_NET_V3:
            f = self.dino_fusion
            out += [f.fusion[0], f.fusion[2], f.attention[0], f.attention[2], f.output_proj]
        out += [m for m in self.density_mlp.density_layers if isinstance(m, nn.Linear)]
        out += [self.density_mlp.density_head, self.density_mlp.feature_head]
        out += [m for m in self.color_mlp.color_layers if isinstance(m, nn.Linear)]
        return out

    def flops_per_sample(self):
        """2*MAC of the Linear layers per ray-sample; NeRFDINOFusion.fusion runs twice (lora_dino.py:181,191)."""
        mac = sum(m.out_features * m.in_features for m in self.linears())
        if self.net == L.NRF_NET_V3:
            f = self.dino_fusion.fusion
            mac += f[0].out_features * f[0].in_features + f[2].out_features * f[2].in_features
        return 2 * mac

    def _arch(self):
        return L.nrf_arch(self.net, self.pos_freq, self.dir_freq, self.hidden_dim, self.n_layers, self.dino_dim)

    def _host_linears(self):
        lins = self.linears()
        keep = []
        arr = (L.nrf_linear * len(lins))()
        for i, m in enumerate(lins):
            w = np.ascontiguousarray(m.weight.detach().to("cpu", torch.float32).numpy())
            b = np.ascontiguousarray(m.bias.detach().to("cpu", torch.float32).numpy())
            keep += [w, b]
            arr[i] = L.nrf_linear(w.ctypes.data_as(L.c_float_p), b.ctypes.data_as(L.c_float_p), m.out_features, m.in_features)
        return arr, len(lins), keep

    def _versions(self):
        return tuple((p.data_ptr(), p._version) for m in self.linears() for p in (m.weight, m.bias)) + (self._gen,)

    def flat_params(self):
        """training.FlatParams of this module (parameters as views into one flat device vector)."""
        if self._flat is None:
            from .training import FlatParams
            self._flat = FlatParams(self)
        return self._flat

    def _flat_on(self, idx):
        """The flat parameter vector if the parameters currently are views into it on device `idx`, else None."""
        fp = self._flat
        if fp is None or fp.flat is None or not fp.flat.is_cuda or fp.flat.device.index != idx:
            return None
        base = fp.flat.data_ptr()
        ok = all(p.data_ptr() == base + 4 * off for p, off in zip(fp.params(), fp.offsets))
        return fp.flat if ok else None

    def handle(self, device=None, mma_mode=None):
        """The nrf_model* for this module on `device`, (re)packed if the parameters changed.  `mma_mode`: the arithmetic
        mode the caller is about to run (default: the module's own) -- after a device-side re-pack only the modes that
        were asked for hold the current parameters."""
        L.require_gpu()
        if device is None:
            p = next(self.parameters())
            device = p.device if p.is_cuda else torch.device("cuda", torch.cuda.current_device())
        device = torch.device(device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        ver = self._versions()
        mode = L.MMA_MODES[mma_mode or self.mma_mode]
        same = self._handle is not None and self._handle_dev == idx
        if same and self._packed == ver and (self._packed_modes is None or mode in self._packed_modes):
            return self._handle
        flat = self._flat_on(idx) if same else None
        if flat is not None:
            # parameters live in the flat device vector (training): re-pack this mode's streams on the device
            with torch.cuda.device(idx):
                L.check(L.lib().nrf_model_update_device(self._handle, L.ptr(flat), 1 << mode, L.stream_ptr()))
            self._packed_modes = ({mode} if self._packed != ver or self._packed_modes is None else self._packed_modes | {mode})
            self._packed = ver
            return self._handle
        arr, n, keep = self._host_linears()
        if same:
            with torch.cuda.device(idx):
                L.check(L.lib().nrf_model_update(self._handle, arr, n, L.stream_ptr()))
        else:
            self.release()
            h = C.c_void_p()
            arch = self._arch()
            L.check(L.lib().nrf_model_create(C.byref(h), idx, C.byref(arch), arr, n))
            self._handle, self._handle_dev = h, idx
            self._train_ready = False
        del keep
        self._packed, self._packed_modes = ver, None
        return self._handle

    # native state never travels with a copy: copy.deepcopy / pickle of a module that has rendered or trained would otherwise
    # duplicate the nrf_model* (freed twice) and alias the flat vectors
    _NATIVE = ("_handle", "_handle_dev", "_packed", "_packed_modes", "_flat", "_flat_grad", "_flat_grad_views", "_linears_cache", "_train_ready")

    def __getstate__(self):
        st = self.__dict__.copy()
        for k in self._NATIVE:
            st.pop(k, None)
        return st

    def __setstate__(self, st):
        super().__setstate__(st)
        self._handle = self._handle_dev = self._packed = self._packed_modes = self._flat = None
        self._train_ready = False
        self.__dict__.setdefault("_gen", 0)

    def release(self):
        if getattr(self, "_handle", None) is not None:
            L.lib().nrf_model_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def _wants_grad(self):
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def forward(self, positions, directions=None, dino_features=None):
        if self._wants_grad():
            from .training import mlp_v1_train, mlp_v2_train
            if self.net == L.NRF_NET_V1:
                return mlp_v1_train(self, positions)
            return mlp_v2_train(self, positions, directions, dino_features)
        mode = L.MMA_MODES[self.mma_mode]
        x = L.dev_f32(positions)            # reached only when no parameter wants a gradient (inference): inputs carry none either
        h = self.handle(x.device)
        with torch.cuda.device(x.device):
            if self.net == L.NRF_NET_V1:
                pe = 3 * (2 * self.pos_freq + 1)
                flat = x.reshape(-1, pe)
                out = torch.empty((flat.shape[0], 4), dtype=torch.float32, device=x.device)
                L.check(L.lib().nrf_mlp_forward_v1(h, mode, L.ptr(flat), flat.shape[0], L.ptr(out), L.stream_ptr()))
                return out.reshape(*x.shape[:-1], 4)
            pos = x.reshape(-1, 3)
            dirs = L.dev_f32(directions, x.device).reshape(-1, 3)
            dino = None
            if self.net == L.NRF_NET_V3:
                if dino_features is None:
                    raise ValueError("use_dino=True needs dino_features")
                dino = L.dev_f32(dino_features, x.device).reshape(-1, self.dino_dim)
            n = pos.shape[0]
            rgb = torch.empty((n, 3), dtype=torch.float32, device=x.device)
            dens = torch.empty((n, 1), dtype=torch.float32, device=x.device)
            L.check(L.lib().nrf_mlp_forward(h, mode, L.ptr(pos), L.ptr(dirs), L.ptr(dino), n, L.ptr(rgb), L.ptr(dens), L.stream_ptr()))
            return rgb, dens


def load_checkpoint_into(model: NeRFMLP, ckpt: dict):
    """Accept both checkpoint key sets the reference writes: `nerf_model_state_dict`
    (train.py:378) and `nerf_state_dict` (train_multiscale.py:368-376, read by evaluate.py:27)."""
    for key in ("nerf_model_state_dict", "nerf_state_dict"):
        if key in ckpt:
            return model.load_state_dict(ckpt[key])
    return model.load_state_dict(ckpt)
