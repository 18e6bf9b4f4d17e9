"""PositionalEncoding module backed by nrf_encode (drop-in for
src/models/positional_encoding.py:5-33 and src/models/nerf_mlp.py:6-39)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib as L


class PositionalEncoding(nn.Module):
    def __init__(self, num_freqs=10, include_input=True, log_sampling=True):
        super().__init__()
        self.num_freqs = int(num_freqs)
        self.include_input = bool(include_input)
        self.log_sampling = bool(log_sampling)
        if log_sampling:
            bands = 2.0 ** torch.linspace(0.0, num_freqs - 1, num_freqs)                  # positional_encoding.py:14
        else:
            bands = torch.linspace(2.0 ** 0.0, 2.0 ** (num_freqs - 1), num_freqs)         # positional_encoding.py:18
        self.register_buffer("freq_bands", bands)

    def get_output_dim(self, input_dim):
        return input_dim * 2 * self.num_freqs + (input_dim if self.include_input else 0)

    def forward(self, x):
        L.require_gpu()
        xd = L.dev_f32(L.refuse_grad(x, "PositionalEncoding"))
        dim = xd.shape[-1]
        flat = xd.reshape(-1, dim)
        bands = None if self.log_sampling else L.dev_f32(self.freq_bands, xd.device)      # powers of two are generated in-kernel
        with torch.cuda.device(xd.device):
            out = torch.empty((flat.shape[0], self.get_output_dim(dim)), dtype=torch.float32, device=xd.device)
            L.check(L.lib().nrf_encode(L.ptr(flat), flat.shape[0], dim, self.num_freqs, int(self.include_input), L.ptr(bands), L.ptr(out),
                                       L.stream_ptr()))
        return out.reshape(*xd.shape[:-1], out.shape[-1])
