"""The fused render call surface: sample -> encode -> MLP -> composite in ONE
kernel launch (nrf_render_rays / nrf_render_camera).

`render_rays` replaces NeRFDINOTrainer.render_rays (src/training/train.py:188-242);
`render_full_image` replaces NeRFDINOEvaluator.render_full_image
(src/training/evaluate.py:65-81) and the eval chunk loop (train.py:305-319);
`NeRFRenderer` carries the trainer-side state those methods read from `self`
(config near/far, the model, the precomputed DINO map of view 0).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L
from .nerf_model import NeRFMLP
from .ray_sampler import _c2w12


def make_dino(features, pose, focal, H, W):
    """Pack the V3 side channel: feature map (1,Hp,Wp,C) of the source view + its camera (train.py:203-214).
    Returns (nrf_dino struct, tensors that must stay alive)."""
    fm = L.dev_f32(features)
    if fm.dim() != 4 or fm.shape[0] != 1:
        raise ValueError("features must be (1,Hp,Wp,C)")
    inv = torch.inverse(torch.as_tensor(pose).detach().to("cpu", torch.float32))      # ray_utils.py:191
    d = L.nrf_dino()
    d.features = fm.data_ptr()
    d.Hp, d.Wp, d.C = int(fm.shape[1]), int(fm.shape[2]), int(fm.shape[3])
    d.inv_pose = (C.c_float * 16)(*inv.reshape(-1).tolist())
    d.focal, d.H, d.W = float(focal), int(H), int(W)
    return d, fm


def _opts(near, far, n_samples, perturb, t_rand, seed, lindisp, ert_eps, white_bkgd, mma_mode, dino, device, z_in=None):
    o = L.nrf_render_opts()
    o.near, o.far, o.n_samples, o.lindisp = float(near), float(far), int(n_samples), int(bool(lindisp))
    o.perturb = int(bool(perturb) or t_rand is not None)
    o.t_rand = t_rand.data_ptr() if t_rand is not None else None
    o.z_ladder = L.z_ladder(near, far, n_samples, lindisp, device).data_ptr()     # cached per (near, far, S, device)
    o.z_in = z_in.data_ptr() if z_in is not None else None
    if seed is None:                 # a new jitter pattern per call (ray_utils.py:78 draws torch.rand per call); repeats under torch.manual_seed
        seed = L.fresh_seed() if (o.perturb and t_rand is None) else 0
    o.rng_seed = int(seed)
    o.ert_eps, o.white_bkgd, o.mma_mode = float(ert_eps), int(bool(white_bkgd)), L.MMA_MODES[mma_mode]
    o.dino = C.pointer(dino) if dino is not None else None
    return o


def render_rays(model: NeRFMLP, rays_o, rays_d, near, far, N_samples=64, perturb=False, t_rand=None, seed=None, lindisp=False,
                ert_eps=0.0, white_bkgd=False, mma_mode: Optional[str] = None, dino=None, return_weights=True, return_z=False,
                z_in=None):
    """Render explicit rays (R,3)/(H,W,3) -> {'rgb' (R,3), 'depth' (R,), 'weights' (R,S)[, 'z_vals' (R,S)]}.
    Under torch.no_grad() or with the model in eval() mode (evaluate, train.py:294-296) this is ONE fused kernel launch.  With
    grad enabled and the model in train() mode (train_step, train.py:245,280-287) the same call returns tensors with a grad_fn:
    the staged training kernels behind one autograd node (training.render_rays_train), so `loss.backward(); optimizer.step()`
    work on the drop-in unchanged.  (`mma_mode` selects the arithmetic on both routes; the split mode trains in exact fp32.)"""
    L.require_gpu()
    if model.training and model._wants_grad():
        from .training import render_rays_train
        out = render_rays_train(model, rays_o, rays_d, near, far, N_samples, perturb=perturb, t_rand=t_rand, seed=seed, lindisp=lindisp,
                                white_bkgd=white_bkgd, dino=dino, z_in=z_in, mma_mode=mma_mode)
        if not return_weights:
            out.pop("weights")
        if not return_z:
            out.pop("z_vals")
        return out
    o = L.dev_f32(rays_o).reshape(-1, 3)
    d = L.dev_f32(rays_d, o.device).reshape(-1, 3)
    R, S = o.shape[0], int(N_samples)
    tr = L.dev_f32(t_rand, o.device).reshape(R, S) if t_rand is not None else None
    zin = L.dev_f32(z_in, o.device).reshape(R, S) if z_in is not None else None
    keep = None
    dn = None
    if model.net == L.NRF_NET_V3:
        if dino is None:
            raise ValueError("a use_dino model needs dino=dict(features=, pose=, focal=, H=, W=)")
        dn, keep = make_dino(**dino)
    opts = _opts(near, far, S, perturb, tr, seed, lindisp, ert_eps, white_bkgd, mma_mode or model.mma_mode, dn, o.device, zin)
    h = model.handle(o.device, mma_mode or model.mma_mode)
    with torch.cuda.device(o.device):
        rgb = torch.empty((R, 3), dtype=torch.float32, device=o.device)
        depth = torch.empty((R,), dtype=torch.float32, device=o.device)
        w = torch.empty((R, S), dtype=torch.float32, device=o.device) if return_weights else None
        z = torch.empty((R, S), dtype=torch.float32, device=o.device) if return_z else None
        L.check(L.lib().nrf_render_rays(h, L.ptr(o), L.ptr(d), R, C.byref(opts), L.ptr(rgb), L.ptr(depth), L.ptr(w), L.ptr(z), L.stream_ptr()))
    del keep
    out = {"rgb": rgb, "depth": depth}
    if w is not None:
        out["weights"] = w
    if z is not None:
        out["z_vals"] = z
    return out


def render_camera(model: NeRFMLP, H, W, focal, c2w, near, far, N_samples=64, ray_begin=0, ray_end=None, perturb=False, seed=None,
                  lindisp=False, ert_eps=0.0, white_bkgd=False, mma_mode: Optional[str] = None, dino=None, device=None,
                  out_rgb=None, out_depth=None):
    """Render rays [ray_begin, ray_end) of an HxW pinhole camera with in-kernel ray generation
    (get_rays + render_rays of the reference, train.py:179,305-319) -> (rgb (n,3), depth (n,))."""
    L.require_gpu()
    H, W = int(H), int(W)
    ray_end = H * W if ray_end is None else int(ray_end)
    n = ray_end - int(ray_begin)
    if device is None:
        p = next(model.parameters())
        device = p.device if p.is_cuda else torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    keep = None
    dn = None
    if model.net == L.NRF_NET_V3:
        if dino is None:
            raise ValueError("a use_dino model needs dino=...")
        dn, keep = make_dino(**dino)
    opts = _opts(near, far, N_samples, perturb, None, seed, lindisp, ert_eps, white_bkgd, mma_mode or model.mma_mode, dn, device)
    h = model.handle(device, mma_mode or model.mma_mode)
    with torch.cuda.device(device):
        rgb = out_rgb if out_rgb is not None else torch.empty((n, 3), dtype=torch.float32, device=device)
        depth = out_depth if out_depth is not None else torch.empty((n,), dtype=torch.float32, device=device)
        L.check(L.lib().nrf_render_camera(h, H, W, float(focal), _c2w12(c2w), int(ray_begin), ray_end, C.byref(opts),
                                          L.ptr(rgb), L.ptr(depth), None, None, L.stream_ptr()))
    del keep
    return rgb, depth


def render_hierarchical(model: NeRFMLP, rays_o, rays_d, near, far, N_samples=128, N_importance=64, perturb=False, u=None, **kw):
    """Coarse pass -> inverse-cdf resampling of its weights -> fine pass on the sorted union of S+Ni depths
    (BASELINE.json config 3: 128 coarse + 64 fine).  Mirrors the intent of ray_utils.py:86-143, which the
    reference never calls and which raises on every input: parity of the resampling step is
    unpinned, the two render passes are the same kernel as `render_rays`.
    Returns the fine pass' dict plus 'coarse' (the coarse pass' dict) and 'z_vals' (R, S+Ni)."""
    from .ray_sampler import sample_pdf
    coarse = render_rays(model, rays_o, rays_d, near, far, N_samples, perturb=perturb, return_weights=True, return_z=True, **kw)
    if u is None and perturb:
        u = torch.rand((coarse["weights"].shape[0], int(N_importance)), device=coarse["weights"].device)
    _, union = sample_pdf(coarse["z_vals"], coarse["weights"], N_importance, u=u)
    kw.pop("t_rand", None)
    fine = render_rays(model, rays_o, rays_d, near, far, N_samples + int(N_importance), z_in=union, return_weights=True, **kw)
    fine["coarse"] = coarse
    fine["z_vals"] = union
    return fine


class NeRFRenderer:
    """Trainer-shaped wrapper: `render_rays(rays_o, rays_d, view_idx, N_samples=64)` and
    `render_full_image(rays_o, rays_d, closest_view_idx, chunk_size=1024)` with the
    signatures of train.py:188 and evaluate.py:65.  `chunk_size` is accepted and
    ignored: the fused kernel never materialises (rays x samples) intermediates."""

    def __init__(self, model: NeRFMLP, near, far, white_bkgd=False, mma_mode=None, ert_eps=0.0,
                 dino_features=None, poses=None, focal=None, H=None, W=None):
        self.nerf_model, self.near, self.far = model, float(near), float(far)
        self.white_bkgd, self.mma_mode, self.ert_eps = white_bkgd, mma_mode, ert_eps
        self.dino_features_precomputed, self.poses, self.focal, self.H, self.W = dino_features, poses, focal, H, W

    def _dino(self, view_idx):
        if self.nerf_model.net != L.NRF_NET_V3:
            return None
        idx = view_idx if self.nerf_model.training else 0          # train.py:203-208
        return dict(features=self.dino_features_precomputed[idx], pose=self.poses[idx], focal=self.focal, H=self.H, W=self.W)

    def render_rays(self, rays_o, rays_d, view_idx=0, N_samples=64):
        """train.py:188: differentiable under grad mode (train_step), the fused kernel under torch.no_grad() (evaluate)."""
        return render_rays(self.nerf_model, rays_o, rays_d, self.near, self.far, N_samples,
                           perturb=self.nerf_model.training, white_bkgd=self.white_bkgd, mma_mode=self.mma_mode,
                           ert_eps=self.ert_eps, dino=self._dino(view_idx))

    @torch.no_grad()
    def render_full_image(self, rays_o, rays_d, closest_view_idx=0, chunk_size=1024, N_samples=64):
        H, W = rays_o.shape[:2]
        out = self.render_rays(rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), closest_view_idx, N_samples)
        return out["rgb"].reshape(H, W, 3).cpu().numpy()
