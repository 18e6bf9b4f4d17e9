"""Rays and stratified samples on the GPU (drop-in for the reference's
src/models/ray_sampler.py and src/utils/ray_utils.py:4-143).

Same names, argument order and shapes as the reference; the work is done by
libnerfhip.so (nrf_get_rays / nrf_sample_along_rays / nrf_sample_pdf).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def _c2w12(c2w) -> "C.Array":
    m = torch.as_tensor(c2w).detach().to("cpu", torch.float32)
    if m.shape not in ((4, 4), (3, 4)):
        raise ValueError(f"c2w must be (4,4) or (3,4), got {tuple(m.shape)}")
    return (C.c_float * 12)(*m[:3, :4].reshape(-1).tolist())


def get_rays(H, W, focal, c2w):
    """rays_o, rays_d of shape (H,W,3); ray id y*W+x.  ray_sampler.py:4-30 == ray_utils.py:4-37."""
    L.require_gpu()
    H, W = int(H), int(W)
    dev = c2w.device if isinstance(c2w, torch.Tensor) and c2w.is_cuda else torch.device("cuda", torch.cuda.current_device())
    with torch.cuda.device(dev):
        rays_o = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
        rays_d = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
        L.check(L.lib().nrf_get_rays(H, W, float(focal), _c2w12(c2w), 0, H * W, L.ptr(rays_o), L.ptr(rays_d), L.stream_ptr()))
    return rays_o, rays_d


def sample_points_along_rays(rays_o, rays_d, near, far, N_samples, perturb=True, lindisp=False, t_rand=None, seed=None):
    """pts (...,S,3), z_vals (...,S) for rays of shape (N,3) or (H,W,3).

    ray_utils.py:39-84 (flat) == ray_sampler.py:32-61 (image).  `perturb=True`
    draws the stratified jitter from the kernel's counter RNG unless `t_rand`
    (same shape as z_vals, U[0,1)) is given.  seed=None (default): a NEW seed per
    call taken from torch's CPU generator, so that, as with the reference's
    torch.rand (ray_utils.py:78), consecutive calls jitter differently and a run
    repeats under torch.manual_seed; an int pins the pattern.
    """
    L.require_gpu()
    o = L.dev_f32(L.refuse_grad(rays_o, "sample_points_along_rays(rays_o)"))
    d = L.dev_f32(L.refuse_grad(rays_d, "sample_points_along_rays(rays_d)"), o.device)
    if seed is None:
        seed = L.fresh_seed() if (perturb and t_rand is None) else 0
    lead = tuple(o.shape[:-1])
    if o.shape[-1] != 3 or d.shape != o.shape:
        raise ValueError("rays_o and rays_d must both be (...,3)")
    o2, d2 = o.reshape(-1, 3), d.reshape(-1, 3)
    R, S = o2.shape[0], int(N_samples)
    tr = None
    if t_rand is not None:
        tr = L.dev_f32(t_rand, o.device).reshape(R, S)
    with torch.cuda.device(o.device):
        pts = torch.empty((R, S, 3), dtype=torch.float32, device=o.device)
        z = torch.empty((R, S), dtype=torch.float32, device=o.device)
        lad = L.z_ladder(near, far, S, lindisp, o.device)
        L.check(L.lib().nrf_sample_along_rays(L.ptr(o2), L.ptr(d2), R, float(near), float(far), S, int(bool(lindisp)),
                                              int(bool(perturb) or tr is not None), L.ptr(tr), L.ptr(lad), int(seed), L.ptr(pts), L.ptr(z),
                                              L.stream_ptr()))
    return pts.reshape(*lead, S, 3), z.reshape(*lead, S)


def hierarchical_sampling(rays_o, rays_d, z_vals, weights, N_importance, perturb=True, u=None):
    """Importance resampling: returns (pts (R,S+Ni,3), z_union (R,S+Ni)) like ray_utils.py:86-143.

    The reference function raises on every input; this is its
    intent with bin EDGES (see oracle/nerf_oracle.py:sample_pdf).  `perturb`
    needs explicit `u` (R,Ni) in [0,1); otherwise u = linspace(0,1,Ni).
    """
    L.require_gpu()
    z = L.dev_f32(z_vals)
    w = L.dev_f32(weights, z.device)
    R, S = z.shape
    Ni = int(N_importance)
    if u is not None:
        uu, stride = L.dev_f32(u, z.device).reshape(R, Ni), Ni
    elif perturb:
        uu, stride = torch.rand((R, Ni), dtype=torch.float32, device=z.device), Ni
    else:
        uu, stride = L.u_row(Ni, z.device), 0                        # ray_utils.py:115-116: linspace(0,1,Ni) as this host computes it
    with torch.cuda.device(z.device):
        union = torch.empty((R, S + Ni), dtype=torch.float32, device=z.device)
        L.check(L.lib().nrf_sample_pdf(L.ptr(z), L.ptr(w), R, S, Ni, L.ptr(uu), stride, None, L.ptr(union), L.stream_ptr()))
    o = L.dev_f32(rays_o, z.device).reshape(R, 3)
    d = L.dev_f32(rays_d, z.device).reshape(R, 3)
    pts = o[:, None, :] + d[:, None, :] * union[:, :, None]
    return pts, union


def sample_pdf(z_vals, weights, N_importance, u=None):
    """(new samples (R,Ni), sorted union (R,S+Ni)) -- the staged form of hierarchical_sampling."""
    L.require_gpu()
    z = L.dev_f32(z_vals)
    w = L.dev_f32(weights, z.device)
    R, S = z.shape
    Ni = int(N_importance)
    uu, stride = (L.u_row(Ni, z.device), 0) if u is None else (L.dev_f32(u, z.device).reshape(R, Ni), Ni)
    with torch.cuda.device(z.device):
        smp = torch.empty((R, Ni), dtype=torch.float32, device=z.device)
        union = torch.empty((R, S + Ni), dtype=torch.float32, device=z.device)
        L.check(L.lib().nrf_sample_pdf(L.ptr(z), L.ptr(w), R, S, Ni, L.ptr(uu), stride, L.ptr(smp), L.ptr(union), L.stream_ptr()))
    return smp, union


def get_ray_batch(rays_o, rays_d, batch_size=1024):
    """ray_utils.py:145-174: host-side chunk generator, kept for callers that still chunk
    (the fused renderer does not need it)."""
    H, W = rays_o.shape[:2]
    n = H * W
    o, d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
    idx = torch.arange(n, device=o.device)
    for i in range(0, n, batch_size):
        yield o[i:i + batch_size], d[i:i + batch_size], idx[i:i + batch_size]
