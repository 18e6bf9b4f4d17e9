"""The training path: what `loss.backward()` / `optimizer.step()` of the reference's loops
(src/training/train_minimal.py:80-127, src/training/train.py:244-292) run through on the GPU.

  * `mlp_v1_train`, `mlp_v2_train`   NeRFMLP.forward with grad enabled: libnerfhip's forward that saves
                     every layer's operand tiles, and a backward made of the transposed weight-stream
                     chain + MFMA weight-gradient kernel (csrc/train_impl.hpp);
  * `composite`      nerf_mlp.VolumeRenderer / volume_render_radiance with grad enabled;
  * `FlatParams`     the module's parameters as views into one flat fp32 vector, so that an
                     optimizer step is visible to the kernels without leaving the device;
  * `Adam`           torch.optim.Adam's update as one kernel on the flat vectors (train.py:113-118);
  * `FusedStep`      the whole optimisation step of the reference's loop as a fixed sequence of library calls;
  * `all_reduce_gradients`   data-parallel training: ONE collective per step on the flat gradient vector.

There is no PyTorch fallback: without libnerfhip.so / a gfx950 GPU every call raises.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


# ---------------------------------------------------------------------------------------------
# flat parameter storage
# ---------------------------------------------------------------------------------------------
class FlatParams:
    """Parameters of `module.linears()` (weight, bias per Linear, state_dict order == include/nerfhip.h's flat
    layout) re-homed as views into one contiguous fp32 device vector.  The nn.Parameter objects stay the same
    (optimizers keep working); only their storage moves."""

    def __init__(self, module):
        self.module = module
        self.flat = None
        self.offsets = []

    def params(self):
        # looked up on every call: Module.to() across device types REPLACES the nn.Parameter objects (a cached list would keep
        # feeding the old ones to autograd); the registry dicts are read directly, nn.Module.__getattr__ is the slow path
        return [m._parameters[k] for m in self.module.linears() for k in ("weight", "bias")]

    def ensure(self):
        ps = self.params()
        dev = ps[0].device
        if not ps[0].is_cuda:
            raise RuntimeError("training runs on the GPU: move the module with .to('cuda') first")
        if self.flat is not None and self.flat.device == dev:
            base = self.flat.data_ptr()
            if all(p.data_ptr() == base + 4 * off and p.dtype == torch.float32 for p, off in zip(ps, self.offsets)):
                return self.flat
        total = sum(p.numel() for p in ps)
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        offsets, off = [], 0
        with torch.no_grad():
            for p in ps:
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1).to(torch.float32))
                p.data = flat[off:off + n].view(p.shape)
                offsets.append(off)
                off += n
        self.flat, self.offsets = flat, offsets
        return flat

    def views(self, vec):
        """Per-parameter views of another flat vector of the same layout (gradients, Adam moments)."""
        return [vec[off:off + p.numel()].view(p.shape) for p, off in zip(self.params(), self.offsets)]


def _grad_target(module):
    """Where a backward should accumulate: (flat vector, attach).  The gradients live in ONE persistent flat vector whose
    per-parameter views are the parameters' .grad, so the kernels add into it directly and autograd's per-parameter
    AccumulateGrad (20+ tensors cloned and added per step) is bypassed.  Cases:
      * every .grad is None (after zero_grad(set_to_none=True)): zero the vector, attach the views;
      * every .grad already is our view: accumulate (two backward calls before a step add up, as in torch);
      * anything else (someone else produced some .grad): return None -- the caller falls back to handing autograd
        ordinary gradient tensors."""
    fp = module.flat_params()
    ps = fp.params()
    flat = fp.flat
    fg = getattr(module, "_flat_grad", None)
    if fg is None or fg.shape != flat.shape or fg.device != flat.device:
        fg = module._flat_grad = torch.zeros_like(flat)
        module._flat_grad_views = fp.views(fg)
    grads = [p.grad for p in ps]
    base = fg.data_ptr()
    if all(g is None for g in grads):
        fg.zero_()
        # the view objects handed out earlier ARE the old .grad tensors: Module.to() re-homes a parameter's .grad in place
        # (same Python object, new storage), after which they no longer alias the flat vector -- check before re-use
        views = module._flat_grad_views
        if any(v.data_ptr() != base + 4 * off for v, off in zip(views, fp.offsets)):
            views = module._flat_grad_views = fp.views(fg)
        for p, v in zip(ps, views):
            p.grad = v
        return fg
    if all(g is not None and g.data_ptr() == base + 4 * off and g.shape == p.shape for g, p, off in zip(grads, ps, fp.offsets)):
        return fg
    return None


def _train_handle(module, dev, mma_mode=None):
    """nrf_model* with forward AND backward streams matching the current parameter values."""
    module.flat_params().ensure()
    train_mode = L.TRAIN_MODE[mma_mode or module.mma_mode]
    h = module.handle(dev, train_mode)
    mode = L.MMA_MODES[train_mode]
    if not module._train_ready:
        # first use: build the backward plan, then pack both directions from the flat vector
        with torch.cuda.device(dev):
            if L.lib().nrf_train_context_bytes(h, mode, 1) < 0:
                raise L.NrfError(-2, L.lib().nrf_last_error().decode("utf-8", "replace"))
            L.check(L.lib().nrf_model_update_device(h, L.ptr(module.flat_params().flat), 1 << mode, L.stream_ptr()))
        module._train_ready = True
        module._packed, module._packed_modes = module._versions(), {mode}
    return h, mode


class _MLPV1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x_enc, *params):
        dev = x_enc.device
        h, mode = _train_handle(module, dev)
        n = x_enc.shape[0]
        out = torch.empty((n, 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            nbytes = L.lib().nrf_train_context_bytes(h, mode, n)
            if nbytes < 0:
                raise L.NrfError(-2, L.lib().nrf_last_error().decode("utf-8", "replace"))
            buf = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=dev)
            L.check(L.lib().nrf_mlp_forward_train_v1(h, mode, L.ptr(x_enc), n, L.ptr(out), C.c_void_p(buf.data_ptr()), nbytes, L.stream_ptr()))
        ctx.module, ctx.buf, ctx.nbytes, ctx.n, ctx.mode = module, buf, nbytes, n, mode
        ctx.versions = module._versions()
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g_out):
        module = ctx.module
        if module._versions() != ctx.versions:
            raise RuntimeError("NeRFMLP parameters were modified between forward and backward: the saved activations "
                               "no longer match the packed weights")
        (out,) = ctx.saved_tensors
        dev = out.device
        g = g_out.to(torch.float32).contiguous()
        fp = module.flat_params()
        direct = _grad_target(module)
        grad = direct if direct is not None else torch.zeros_like(fp.flat)
        with torch.cuda.device(dev):
            L.check(L.lib().nrf_mlp_backward_v1(module._handle, ctx.mode, L.ptr(out), L.ptr(g), ctx.n, C.c_void_p(ctx.buf.data_ptr()), ctx.nbytes,
                                                L.ptr(grad), L.stream_ptr()))
        ctx.buf = None
        if direct is not None:
            return (None, None) + (None,) * len(fp.offsets)       # already accumulated into the parameters' .grad
        return (None, None, *fp.views(grad))


class _MLPV2Fn(torch.autograd.Function):
    """nerf_mlp.py:134-158: (positions, directions[, dino features]) -> (rgb, density); V2 and V3 models."""

    @staticmethod
    def forward(ctx, module, pos, dirs, dino, *params):
        dev = pos.device
        h, mode = _train_handle(module, dev)
        n = pos.shape[0]
        rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
        dens = torch.empty((n, 1), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            nbytes = L.lib().nrf_train_context_bytes(h, mode, n)
            if nbytes < 0:
                raise L.NrfError(-2, L.lib().nrf_last_error().decode("utf-8", "replace"))
            buf = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=dev)
            L.check(L.lib().nrf_mlp_forward_train(h, mode, L.ptr(pos), L.ptr(dirs), L.ptr(dino), n, L.ptr(rgb), L.ptr(dens),
                                                  C.c_void_p(buf.data_ptr()), nbytes, L.stream_ptr()))
        ctx.module, ctx.buf, ctx.nbytes, ctx.n, ctx.mode = module, buf, nbytes, n, mode
        ctx.versions = module._versions()
        ctx.save_for_backward(rgb, dens)
        return rgb, dens

    @staticmethod
    def backward(ctx, g_rgb, g_dens):
        module = ctx.module
        if module._versions() != ctx.versions:
            raise RuntimeError("NeRFMLP parameters were modified between forward and backward: the saved activations "
                               "no longer match the packed weights")
        rgb, dens = ctx.saved_tensors
        dev = rgb.device
        g_rgb = g_rgb.to(torch.float32).contiguous()
        g_dens = g_dens.to(torch.float32).contiguous()
        fp = module.flat_params()
        direct = _grad_target(module)
        grad = direct if direct is not None else torch.zeros_like(fp.flat)
        with torch.cuda.device(dev):
            L.check(L.lib().nrf_mlp_backward(module._handle, ctx.mode, L.ptr(rgb), L.ptr(dens), L.ptr(g_rgb), L.ptr(g_dens), ctx.n,
                                             C.c_void_p(ctx.buf.data_ptr()), ctx.nbytes, L.ptr(grad), L.stream_ptr()))
        ctx.buf = None
        if direct is not None:
            return (None, None, None, None) + (None,) * len(fp.offsets)
        return (None, None, None, None, *fp.views(grad))


def mlp_v2_train(module, positions, directions, dino_features=None):
    """(P,3) positions, (P,3) directions [, (P,C) DINO features for the use_dino=True form] -> rgb (P,3), density (P,1),
    differentiable with respect to the parameters (not the inputs: a feature tensor that requires grad is refused)."""
    pos = L.dev_f32(L.refuse_grad(positions, "NeRFMLP.forward(positions)")).reshape(-1, 3)
    dirs = L.dev_f32(L.refuse_grad(directions, "NeRFMLP.forward(directions)"), pos.device).reshape(-1, 3)
    dino = None
    if module.net == L.NRF_NET_V3:
        if dino_features is None:
            raise ValueError("use_dino=True needs dino_features")
        if getattr(dino_features, "requires_grad", False):
            raise NotImplementedError("no gradient with respect to the DINO features is produced (the feature extractor, LoRA "
                                      "included, is outside the HIP path: SURVEY.md section 8 f4); detach them")
        dino = L.dev_f32(dino_features, pos.device).reshape(-1, module.dino_dim)
    return _MLPV2Fn.apply(module, pos, dirs, dino, *module.flat_params().params())


def mlp_v1_train(module, x_enc):
    """(P, 63) encoded points -> (P, 4) = [sigmoid rgb, raw sigma], differentiable with respect to the parameters."""
    x = L.dev_f32(L.refuse_grad(x_enc, "NeRFMLP.forward(x_encoded)"))
    pe = 3 * (2 * module.pos_freq + 1)
    flat_in = x.reshape(-1, pe)
    out = _MLPV1Fn.apply(module, flat_in, *module.flat_params().params())
    return out.reshape(*x.shape[:-1], 4)


# ---------------------------------------------------------------------------------------------
# compositing
# ---------------------------------------------------------------------------------------------
class _CompositeFn(torch.autograd.Function):
    """rgb (R,S,Cs>=3 strided), sigma (R,S strided) -> rgb_map, depth, weights; gradients for rgb and sigma only
    (the reference never differentiates the sample depths or ray directions)."""

    @staticmethod
    def forward(ctx, packed, z, d, white_bkgd):
        R, S = z.shape
        dev = z.device
        out_rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
        out_depth = torch.empty((R,), dtype=torch.float32, device=dev)
        out_w = torch.empty((R, S), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.check(L.lib().nrf_composite(L.ptr(packed), 4, C.c_void_p(packed.data_ptr() + 12), 4, L.ptr(z), L.ptr(d), R, S,
                                          int(bool(white_bkgd)), L.ptr(out_rgb), L.ptr(out_depth), L.ptr(out_w), L.stream_ptr()))
        ctx.save_for_backward(packed, z, d)
        ctx.white = int(bool(white_bkgd))
        ctx.set_materialize_grads(False)      # unused outputs arrive as None, not as zero tensors
        return out_rgb, out_depth, out_w

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_w):
        packed, z, d = ctx.saved_tensors
        R, S = z.shape
        dev = z.device

        def prep(g):
            return None if g is None else g.to(torch.float32).contiguous()
        g_rgb, g_depth, g_w = prep(g_rgb), prep(g_depth), prep(g_w)
        d_packed = torch.empty_like(packed)
        with torch.cuda.device(dev):
            L.check(L.lib().nrf_composite_backward(L.ptr(packed), 4, C.c_void_p(packed.data_ptr() + 12), 4, L.ptr(z), L.ptr(d), R, S, ctx.white,
                                                   L.ptr(g_rgb), L.ptr(g_depth), L.ptr(g_w), L.ptr(d_packed), 4,
                                                   C.c_void_p(d_packed.data_ptr() + 12), 4, L.stream_ptr()))
        return d_packed, None, None, None


def composite(rgb_sigma, z, d, white_bkgd=False):
    """Differentiable alpha compositing of (R,S,4) [r,g,b,sigma] rows."""
    return _CompositeFn.apply(rgb_sigma.contiguous(), z, d, white_bkgd)


# ---------------------------------------------------------------------------------------------
# render_rays with grad enabled: the trainer-shaped entry point (train.py:188-242 inside train_step, :280-287)
# ---------------------------------------------------------------------------------------------
class _RenderFn(torch.autograd.Function):
    """Per-sample network inputs + depths + rays -> (rgb (R,3), depth (R,), weights (R,S)), differentiable with respect to the
    parameters: ONE autograd node over the kernels FusedStep runs (saving forward -> composite | composite backward -> dZ
    chain + weight gradients), so `loss.backward()` of the reference's unmodified train_step body lands in the parameters'
    .grad (views of the module's flat gradient vector)."""

    @staticmethod
    def forward(ctx, module, x, dirs, dino, z, d, white, mma_mode, *params):
        dev = z.device
        R, S = z.shape
        n = R * S
        h, mode = _train_handle(module, dev, mma_mode)
        v1 = module.net == L.NRF_NET_V1
        lib = L.lib()
        with torch.cuda.device(dev):
            nbytes = lib.nrf_train_context_bytes(h, mode, n)
            if nbytes < 0:
                raise L.NrfError(-2, lib.nrf_last_error().decode("utf-8", "replace"))
            buf = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=dev)
            o4 = torch.empty((n, 4), dtype=torch.float32, device=dev)          # V1: [r,g,b,sigma] rows; V2/V3: rgb (n,3) | density (n,1)
            out_rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
            out_depth = torch.empty((R,), dtype=torch.float32, device=dev)
            out_w = torch.empty((R, S), dtype=torch.float32, device=dev)
            st, cb = L.stream_ptr(), C.c_void_p(buf.data_ptr())
            if v1:
                L.check(lib.nrf_mlp_forward_train_v1(h, mode, L.ptr(x), n, L.ptr(o4), cb, nbytes, st))
                L.check(lib.nrf_composite(L.ptr(o4), 4, C.c_void_p(o4.data_ptr() + 12), 4, L.ptr(z), L.ptr(d), R, S, white,
                                          L.ptr(out_rgb), L.ptr(out_depth), L.ptr(out_w), st))
            else:
                rgb, den = o4.view(-1)[:3 * n].view(n, 3), o4.view(-1)[3 * n:].view(n, 1)
                L.check(lib.nrf_mlp_forward_train(h, mode, L.ptr(x), L.ptr(dirs), L.ptr(dino), n, L.ptr(rgb), L.ptr(den), cb, nbytes, st))
                L.check(lib.nrf_composite(L.ptr(rgb), 3, L.ptr(den), 1, L.ptr(z), L.ptr(d), R, S, white, L.ptr(out_rgb), L.ptr(out_depth), L.ptr(out_w), st))
        ctx.module, ctx.buf, ctx.nbytes, ctx.mode, ctx.white, ctx.v1 = module, buf, nbytes, mode, white, v1
        ctx.versions = module._packed                    # the versions handle() just packed (== module._versions(), not recomputed)
        ctx.save_for_backward(o4, z, d)
        ctx.set_materialize_grads(False)
        return out_rgb, out_depth, out_w

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_w):
        module = ctx.module
        if module._versions() != ctx.versions:
            raise RuntimeError("NeRFMLP parameters were modified between render_rays and backward: the saved activations "
                               "no longer match the packed weights")
        o4, z, d = ctx.saved_tensors
        R, S = z.shape
        n = R * S
        dev = z.device
        lib = L.lib()

        def prep(g):
            return None if g is None else g.to(torch.float32).contiguous()
        g_rgb, g_depth, g_w = prep(g_rgb), prep(g_depth), prep(g_w)
        n_in = 8
        fp = module.flat_params()
        if g_rgb is None and g_depth is None and g_w is None:
            return (None,) * (n_in + len(fp.offsets))
        direct = _grad_target(module)
        grad = direct if direct is not None else torch.zeros_like(fp.flat)
        with torch.cuda.device(dev):
            d4 = torch.empty_like(o4)
            st, cb = L.stream_ptr(), C.c_void_p(ctx.buf.data_ptr())
            if ctx.v1:
                L.check(lib.nrf_composite_backward(L.ptr(o4), 4, C.c_void_p(o4.data_ptr() + 12), 4, L.ptr(z), L.ptr(d), R, S, ctx.white,
                                                   L.ptr(g_rgb), L.ptr(g_depth), L.ptr(g_w), L.ptr(d4), 4, C.c_void_p(d4.data_ptr() + 12), 4, st))
                L.check(lib.nrf_mlp_backward_v1(module._handle, ctx.mode, L.ptr(o4), L.ptr(d4), n, cb, ctx.nbytes, L.ptr(grad), st))
            else:
                rgb, den = o4.view(-1)[:3 * n].view(n, 3), o4.view(-1)[3 * n:].view(n, 1)
                d_rgb, d_den = d4.view(-1)[:3 * n].view(n, 3), d4.view(-1)[3 * n:].view(n, 1)
                L.check(lib.nrf_composite_backward(L.ptr(rgb), 3, L.ptr(den), 1, L.ptr(z), L.ptr(d), R, S, ctx.white,
                                                   L.ptr(g_rgb), L.ptr(g_depth), L.ptr(g_w), L.ptr(d_rgb), 3, L.ptr(d_den), 1, st))
                L.check(lib.nrf_mlp_backward(module._handle, ctx.mode, L.ptr(rgb), L.ptr(den), L.ptr(d_rgb), L.ptr(d_den), n, cb, ctx.nbytes,
                                             L.ptr(grad), st))
        ctx.buf = None
        if direct is not None:
            return (None,) * (n_in + len(fp.offsets))
        return (None,) * n_in + tuple(fp.views(grad))


_encoders = {}


def render_rays_train(module, rays_o, rays_d, near, far, n_samples, perturb=True, t_rand=None, seed=None, lindisp=False,
                      white_bkgd=False, dino=None, z_in=None, mma_mode=None):
    """renderer.render_rays when grad is enabled: the reference's own sequence (train.py:188-242) -- stratified samples,
    [project + fetch DINO features,] NeRFMLP, VolumeRenderer -- returning {'rgb','depth','weights','z_vals'} that carry a grad_fn.
    Gradients reach the parameters only (rays, depths and features are data: a tensor that requires grad is refused).
    The arithmetic mode is `mma_mode` (default: the module's own) mapped to a training mode (_lib.TRAIN_MODE: the split mode
    trains in exact fp32); early ray termination does not apply."""
    from .ray_sampler import sample_points_along_rays
    o = L.dev_f32(L.refuse_grad(rays_o, "render_rays(rays_o)")).reshape(-1, 3)
    d = L.dev_f32(L.refuse_grad(rays_d, "render_rays(rays_d)"), o.device).reshape(-1, 3)
    R, S = o.shape[0], int(n_samples)
    if z_in is not None:
        z = L.dev_f32(z_in, o.device).reshape(R, S)
        pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
    else:
        pts, z = sample_points_along_rays(o, d, near, far, S, perturb=perturb, lindisp=lindisp, t_rand=t_rand, seed=seed)
    pts = pts.reshape(-1, 3)
    dirs = feats = None
    if module.net == L.NRF_NET_V1:
        from .positional_encoding import PositionalEncoding
        enc = _encoders.get(module.pos_freq)
        if enc is None:
            enc = _encoders[module.pos_freq] = PositionalEncoding(module.pos_freq)
        x = enc(pts)                                                     # train_minimal.py:101
    else:
        x = pts
        dirs = d[:, None, :].expand(R, S, 3).reshape(-1, 3).contiguous()          # train.py:225: raw ray directions per sample
        if module.net == L.NRF_NET_V3:
            if dino is None:
                raise ValueError("a use_dino model needs dino=dict(features=, pose=, focal=, H=, W=)")
            if getattr(dino.get("features"), "requires_grad", False):
                raise NotImplementedError("no gradient with respect to the DINO feature map is produced; detach it")
            from .renderer import make_dino
            dn, keep = make_dino(**dino)
            feats = torch.empty((R * S, module.dino_dim), dtype=torch.float32, device=o.device)
            with torch.cuda.device(o.device):
                L.check(L.lib().nrf_project_fetch(C.byref(dn), L.ptr(pts), R * S, L.ptr(feats), None, L.stream_ptr()))   # train.py:203-217
            del keep
    rgb, depth, w = _RenderFn.apply(module, x, dirs, feats, z, d, int(bool(white_bkgd)), mma_mode, *module.flat_params().params())
    return {"rgb": rgb, "depth": depth, "weights": w, "z_vals": z}


# ---------------------------------------------------------------------------------------------
# optimizer
# ---------------------------------------------------------------------------------------------
class Adam:
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) for a NeRFMLP whose parameters live in a FlatParams
    vector: one kernel per step.  Same update rule and defaults as the reference's optimizer (train.py:113-118)."""

    def __init__(self, module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.module = module
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.step_count = 0
        self.exp_avg = None
        self.exp_avg_sq = None
        self.grad = None

    def _buffers(self):
        fp = self.module.flat_params()
        flat = fp.ensure()
        if self.exp_avg is None or self.exp_avg.shape != flat.shape or self.exp_avg.device != flat.device:
            self.exp_avg = torch.zeros_like(flat)
            self.exp_avg_sq = torch.zeros_like(flat)
        return fp, flat

    @staticmethod
    def _flat_grad_of(ps, fp, flat):
        """The one vector all .grad tensors are views of, if they are (the backward of training.py hands out views of its
        flat gradient in exactly this layout and autograd keeps them when nothing else accumulated)."""
        g0 = ps[0].grad
        base = getattr(g0, "_base", None) if g0 is not None else None
        if base is not None and base.dim() != 1:
            base = None
        if base is None or base.shape != flat.shape or base.dtype != torch.float32 or base.device != flat.device or not base.is_contiguous():
            return None
        b = base.data_ptr()
        for p, off in zip(ps, fp.offsets):
            if p.grad is None or p.grad.data_ptr() != b + 4 * off:
                return None
        return base

    def zero_grad(self, set_to_none=True):
        for p in self.module.flat_params().params():
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def step(self):
        fp, flat = self._buffers()
        ps = fp.params()
        g = self._flat_grad_of(ps, fp, flat)
        if g is None:
            # gather the gradients autograd left on the parameters into the flat layout
            g = self.grad if self.grad is not None and self.grad.shape == flat.shape and self.grad.device == flat.device else torch.empty_like(flat)
            self.grad = g
            with torch.no_grad():
                for p, v in zip(ps, fp.views(g)):
                    if p.grad is None:
                        v.zero_()
                    elif p.grad.data_ptr() != v.data_ptr():
                        v.copy_(p.grad)
        self.step_count += 1
        with torch.no_grad(), torch.cuda.device(flat.device):
            L.check(L.lib().nrf_adam_step(L.ptr(flat), L.ptr(g), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq), flat.numel(), self.lr,
                                          self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count, L.stream_ptr()))
        self.module._gen += 1            # the packed streams are now older than the parameters


# ---------------------------------------------------------------------------------------------
# data-parallel training: one collective per step
# ---------------------------------------------------------------------------------------------
def _all_reduce_mean(t, group, average=True):
    """In-place all-reduce of a device vector.  RCCL (backend "nccl") reduces device memory directly; under gloo (CPU
    rehearsals, one-GPU test boxes) the vector takes the host round trip gloo needs."""
    import torch.distributed as dist
    if dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    if average:
        t.div_(dist.get_world_size(group))
    return t


def all_reduce_gradients(model, group=None, average=True):
    """Sum (or average) the gradients of all ranks before optimizer.step().  The gradients of a NeRFMLP live in ONE flat
    vector (see _grad_target), so data-parallel training costs a single all-reduce of nrf_param_count floats
    (1.9 MB for the 8x256 network: one RCCL ring step over xGMI) instead of one per parameter.  Rays shard over the ranks
    (every rank renders its own rays of the batch: no exchange inside the path); call between backward() and step()."""
    fp = model.flat_params()
    fg = getattr(model, "_flat_grad", None)
    ps = fp.params()
    if fg is None or any(p.grad is None for p in ps) or any(p.grad.data_ptr() != fg.data_ptr() + 4 * off for p, off in zip(ps, fp.offsets)):
        raise RuntimeError("all_reduce_gradients: the parameters' .grad are not the flat gradient vector of this module "
                           "(call it right after loss.backward())")
    return _all_reduce_mean(fg, group, average)


# ---------------------------------------------------------------------------------------------
# one optimisation step without autograd in between
# ---------------------------------------------------------------------------------------------
class FusedStep:
    """The body of the reference's inner loop -- render a batch of rays, `rgb_weight * mse(pred['rgb'], target)`
    (train.py:36-44,280-288; train_minimal.py:102-123), backward, Adam -- as a fixed sequence of libnerfhip calls on
    preallocated buffers: saving forward -> [composite, mse and its gradient, composite backward: one launch] -> dZ chain +
    weight gradients -> Adam -> (next step) device re-pack.  Same kernels and the same numbers as the autograd route; what it saves is the
    autograd graph, the per-parameter gradient tensors and the Python between the launches, which at the reference's batch
    sizes (1-2 k rays x 32-64 samples) cost more than the kernels.

        step = FusedStep(model, lr=5e-4, weight_decay=1e-6)
        loss = step(points, z_vals, rays_d, target)               # V1: points = encoded (R*S, 63)
        loss = step(points, z_vals, rays_d, target, dirs=dirs)    # V2: points (R*S, 3), dirs (R*S, 3)
        loss = step(points, z_vals, rays_d, target, dirs=dirs, dino=feats)    # V3: + per-sample DINO features (R*S, C)
    """

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, rgb_weight=1.0, white_bkgd=False,
                 process_group=None, data_parallel=False):
        """data_parallel=True (inside an initialised torch.distributed job): every rank passes ITS shard of the ray batch
        (equal sizes), the flat gradient vector is averaged over the ranks with one all-reduce before Adam, and the
        returned loss is this rank's."""
        if model.net not in (L.NRF_NET_V1, L.NRF_NET_V2, L.NRF_NET_V3):
            raise NotImplementedError("FusedStep: unknown network family")
        self.model = model
        self.data_parallel = bool(data_parallel)
        self.group = process_group
        self.opt = Adam(model, lr, betas, eps, weight_decay)
        self.rgb_weight = float(rgb_weight)
        self.white = int(bool(white_bkgd))
        self._key = None

    def _buffers(self, n, R, S, dev, h, mode):
        key = (n, R, S, str(dev), mode)
        if self._key != key:
            nbytes = L.lib().nrf_train_context_bytes(h, mode, n)
            if nbytes < 0:
                raise L.NrfError(-2, L.lib().nrf_last_error().decode("utf-8", "replace"))
            f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
            self.ctx = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=dev)
            self.nbytes = nbytes
            self.out4, self.d_out4 = f(n, 4), f(n, 4)                 # V1: [rgb, sigma] rows; V2: rgb | density packed in the same rows
            self.pred = f(R, 3)
            self.ray_loss = f(R)
            self.grad = torch.zeros(self.model.flat_params().flat.numel(), dtype=torch.float32, device=dev)
            self._key = key

    @torch.no_grad()
    def __call__(self, points, z_vals, rays_d, target, dirs=None, dino=None):
        m = self.model
        pts = L.dev_f32(points)
        dev = pts.device
        z = L.dev_f32(z_vals, dev)
        R, S = z.shape
        d = L.dev_f32(rays_d, dev).reshape(R, 3)
        tgt = L.dev_f32(target, dev).reshape(R, 3)
        v2 = m.net != L.NRF_NET_V1                                   # trainer forms: positions + directions (+ DINO features)
        pts = pts.reshape(R * S, 3 if v2 else 3 * (2 * m.pos_freq + 1))
        n = R * S
        lib = L.lib()
        h, mode = _train_handle(m, dev)
        with torch.cuda.device(dev):
            self._buffers(n, R, S, dev, h, mode)
            st = L.stream_ptr()
            ctx = C.c_void_p(self.ctx.data_ptr())
            o4, d4 = self.out4, self.d_out4
            if v2:
                # rgb -> columns 0..2 of the first 3n/4 rows' worth of storage is not strided: keep two plain tensors as views
                rgb, den = o4.view(-1)[:3 * n].view(n, 3), o4.view(-1)[3 * n:].view(n, 1)
                g_rgb, g_den = d4.view(-1)[:3 * n].view(n, 3), d4.view(-1)[3 * n:].view(n, 1)
                dirs_d = L.dev_f32(dirs, dev).reshape(n, 3)
                dino_d = L.dev_f32(dino, dev).reshape(n, m.dino_dim) if m.net == L.NRF_NET_V3 else None
                L.check(lib.nrf_mlp_forward_train(h, mode, L.ptr(pts), L.ptr(dirs_d), L.ptr(dino_d), n, L.ptr(rgb), L.ptr(den), ctx, self.nbytes, st))
                heads = (L.ptr(rgb), 3, L.ptr(den), 1)
                d_heads = (L.ptr(g_rgb), 3, L.ptr(g_den), 1)
            else:
                L.check(lib.nrf_mlp_forward_train_v1(h, mode, L.ptr(pts), n, L.ptr(o4), ctx, self.nbytes, st))
                heads = (L.ptr(o4), 4, C.c_void_p(o4.data_ptr() + 12), 4)
                d_heads = (L.ptr(d4), 4, C.c_void_p(d4.data_ptr() + 12), 4)
            # compositor -> d loss / d pred = 2 w (pred - target) / (3 R) -> compositor backward, and the flat gradient vector cleared:
            # one launch (a ray's loss gradient needs only its own prediction); the rays' squared errors stay in ray_loss
            loss = torch.empty((), dtype=torch.float32, device=dev)
            L.check(lib.nrf_composite_mse_backward(*heads, L.ptr(z), L.ptr(d), R, S, self.white, L.ptr(tgt), self.rgb_weight, L.ptr(self.pred),
                                                   *d_heads, L.ptr(self.ray_loss), L.ptr(self.grad), self.grad.numel(), st))
            if v2:
                L.check(lib.nrf_mlp_backward(h, mode, L.ptr(rgb), L.ptr(den), L.ptr(g_rgb), L.ptr(g_den), n, ctx, self.nbytes, L.ptr(self.grad), st))
            else:
                L.check(lib.nrf_mlp_backward_v1(h, mode, L.ptr(o4), L.ptr(d4), n, ctx, self.nbytes, L.ptr(self.grad), st))
            if self.data_parallel:
                _all_reduce_mean(self.grad, self.group)
            opt = self.opt
            fp, flat = opt._buffers()
            opt.step_count += 1
            # Adam, and as a side job of its launch the loss value: rgb_weight * sum(ray_loss) / (3 R) in a fixed order
            L.check(lib.nrf_adam_step_loss(L.ptr(flat), L.ptr(self.grad), L.ptr(opt.exp_avg), L.ptr(opt.exp_avg_sq), flat.numel(), opt.lr,
                                           opt.betas[0], opt.betas[1], opt.eps, opt.weight_decay, opt.step_count, L.ptr(self.ray_loss), R,
                                           self.rgb_weight, L.ptr(loss), st))
        m._gen += 1
        return loss
