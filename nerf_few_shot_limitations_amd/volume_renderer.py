"""Alpha compositing on the GPU (drop-in for src/models/nerf_mlp.py:160-215
`VolumeRenderer` and src/models/volume_renderer.py:4-43 `volume_render_radiance`)."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L


def _composite(rgb, rgb_stride, sigma, sigma_stride, z, d, R, S, white_bkgd, want_depth, want_w):
    dev = z.device
    with torch.cuda.device(dev):
        out_rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
        out_depth = torch.empty((R,), dtype=torch.float32, device=dev) if want_depth else None
        out_w = torch.empty((R, S), dtype=torch.float32, device=dev) if want_w else None
        L.check(L.lib().nrf_composite(L.ptr(rgb), rgb_stride, L.ptr(sigma), sigma_stride,
                                      L.ptr(z), L.ptr(d), R, S, int(bool(white_bkgd)), L.ptr(out_rgb), L.ptr(out_depth), L.ptr(out_w),
                                      L.stream_ptr()))
    return out_rgb, out_depth, out_w


class VolumeRenderer(nn.Module):
    def forward(self, rgb, density, z_vals, rays_d, noise_std=0.0, white_bkgd=False):
        """rgb (R,S,3), density (R,S,1), z_vals (R,S), rays_d (R,3) -> (rgb (R,3), depth (R,), weights (R,S))."""
        L.require_gpu()
        if noise_std > 0.0 and self.training:
            # nerf_mlp.py:188-190 -- training-only regulariser; no caller in the reference passes it
            density = density + torch.randn_like(density) * noise_std
        z = L.dev_f32(z_vals)
        R, S = z.shape
        if torch.is_grad_enabled() and (getattr(rgb, "requires_grad", False) or getattr(density, "requires_grad", False)):
            # training (train.py:236 inside loss.backward()'s graph): differentiable with respect to rgb and density
            from .training import composite
            packed = torch.cat([rgb.reshape(R, S, 3), density.reshape(R, S, 1)], dim=-1).to(device=z.device, dtype=torch.float32)
            return composite(packed, z, L.dev_f32(rays_d, z.device).reshape(R, 3), white_bkgd)
        c = L.dev_f32(rgb, z.device).reshape(R, S, 3)
        sg = L.dev_f32(density, z.device).reshape(R, S)
        d = L.dev_f32(rays_d, z.device).reshape(R, 3)
        return _composite(c, 3, sg, 1, z, d, R, S, white_bkgd, True, True)


def volume_render_radiance(rgb_sigma, z_vals, rays_d, noise_std=0.0):
    """rgb_sigma (H,W,S,4)=[r,g,b,sigma], z_vals (H,W,S), rays_d (H,W,3) -> rgb (H,W,3)."""
    L.require_gpu()
    if torch.is_grad_enabled() and getattr(rgb_sigma, "requires_grad", False):
        # train_minimal.py:53,120 -- the loss is taken on this output
        from .training import composite
        if noise_std > 0.0:
            noise = torch.zeros_like(rgb_sigma)
            noise[..., 3] = noise_std * torch.randn_like(rgb_sigma[..., 3])
            rgb_sigma = rgb_sigma + noise
        lead = tuple(rgb_sigma.shape[:-2])
        S = rgb_sigma.shape[-2]
        packed = rgb_sigma.reshape(-1, S, 4).to(torch.float32)
        z = L.dev_f32(z_vals, packed.device).reshape(-1, S)
        d = L.dev_f32(rays_d, packed.device).reshape(-1, 3)
        return composite(packed, z, d, False)[0].reshape(*lead, 3)
    rs = L.dev_f32(rgb_sigma)
    if noise_std > 0.0:
        rs = rs.clone()
        rs[..., 3] += noise_std * torch.randn_like(rs[..., 3])       # volume_renderer.py:28-29 (applied out of place here)
    lead = tuple(rs.shape[:-2])
    S = rs.shape[-2]
    R = 1
    for n in lead:
        R *= n
    z = L.dev_f32(z_vals, rs.device).reshape(R, S)
    d = L.dev_f32(rays_d, rs.device).reshape(R, 3)
    flat = rs.reshape(R, S, 4)
    dev = z.device
    with torch.cuda.device(dev):
        out_rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
        L.check(L.lib().nrf_composite(L.ptr(flat), 4, C.c_void_p(flat.data_ptr() + 12), 4, L.ptr(z), L.ptr(d), R, S, 0,
                                      L.ptr(out_rgb), None, None, L.stream_ptr()))
    return out_rgb.reshape(*lead, 3)
