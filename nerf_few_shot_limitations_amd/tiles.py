"""Multi-GPU frames: pixel-tile sharding + one collective at frame end.

Rays are independent, so there is no exchange while rendering.  The image's H*W ray ids are cut
into tiles of `tile_rays` consecutive ids dealt round-robin (tile t -> rank t mod world: with early
ray termination the cost of a tile is scene dependent, small interleaved tiles balance it); every
rank renders its tiles -- of a whole batch of views -- with ONE nrf_render_cameras_tiles launch into a
(views, tiles_per_rank*tile_rays, 4) buffer [r,g,b,depth] and the buffers are exchanged with ONE all_gather (backend "nccl" = RCCL over
xGMI on the GPU box; "gloo" in the CPU tests).  The partition / reassembly arithmetic lives in plain
functions so that it is testable without a GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L


def tile_plan(n_rays: int, world: int, tile_rays: int):
    """(tiles_total, tiles_per_rank): every rank gets the same count (the tail is padding)."""
    tiles_total = (n_rays + tile_rays - 1) // tile_rays
    per_rank = (tiles_total + world - 1) // world
    return tiles_total, per_rank


def local_ray_ids(rank: int, world: int, n_rays: int, tile_rays: int) -> torch.Tensor:
    """Global ray id of every row of rank's local buffer (padding rows clamp to the last ray) --
    the integer contract nrf_render_camera_tiles implements."""
    _, per_rank = tile_plan(n_rays, world, tile_rays)
    k = torch.arange(per_rank, dtype=torch.int64)[:, None]
    j = torch.arange(tile_rays, dtype=torch.int64)[None, :]
    ids = (rank + k * world) * tile_rays + j
    return ids.clamp_(max=n_rays - 1).reshape(-1)


def reassemble(gathered: torch.Tensor, n_rays: int, world: int, tile_rays: int) -> torch.Tensor:
    """gathered (world, per_rank*tile_rays, C) -> frame (n_rays, C) in ray-id order."""
    _, per_rank = tile_plan(n_rays, world, tile_rays)
    c = gathered.shape[-1]
    g = gathered.reshape(world, per_rank, tile_rays, c).permute(1, 0, 2, 3)        # tile t = k*world + rank
    return g.reshape(per_rank * world * tile_rays, c)[:n_rays]


def gather_frame(local: torch.Tensor, n_rays: int, tile_rays: int, group=None) -> torch.Tensor:
    """All ranks contribute their (per_rank*tile_rays, C) buffer; every rank gets the (n_rays, C) frame."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return reassemble(local[None], n_rays, 1, tile_rays)
    out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out.view(-1), local.contiguous().view(-1), group=group)
    return reassemble(out, n_rays, world, tile_rays)


def _poses12(c2w) -> "C.Array":
    """(V,4,4)/(V,3,4)/(4,4)/(3,4) poses -> V*12 floats (first three rows of each)."""
    m = torch.as_tensor(c2w).detach().to("cpu", torch.float32)
    if m.dim() == 2:
        m = m[None]
    if m.shape[1:] not in ((4, 4), (3, 4)):
        raise ValueError(f"poses must be (V,4,4) or (V,3,4), got {tuple(m.shape)}")
    flat = m[:, :3, :4].reshape(-1).tolist()
    return (C.c_float * len(flat))(*flat), m.shape[0]


class TileJob:
    """Everything one rank needs to render its tiles of a batch of views, prepared once:
    `launch()` enqueues exactly the render kernel(s), which write rgb+depth rows straight into the gather buffer `buf`."""

    def __init__(self, model, H, W, focal, c2w, near, far, N_samples, rank, world, tile_rays, perturb=False, seed=0, lindisp=False,
                 ert_eps=0.0, white_bkgd=False, mma_mode: Optional[str] = None, dino=None, device=None):
        from .renderer import _opts, make_dino
        L.require_gpu()
        self.H, self.W, self.focal = int(H), int(W), float(focal)
        self.n_rays = self.H * self.W
        self.rank, self.world, self.tile_rays = int(rank), int(world), int(tile_rays)
        self.tiles_total, self.per_rank = tile_plan(self.n_rays, self.world, self.tile_rays)
        if device is None:
            p = next(model.parameters())
            device = p.device if p.is_cuda else torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self._dn = self._keep = None
        if model.net == L.NRF_NET_V3:
            self._dn, self._keep = make_dino(**dino)
        self.mma_mode = mma_mode or model.mma_mode
        self.opts = _opts(near, far, N_samples, perturb, None, seed, lindisp, ert_eps, white_bkgd, self.mma_mode,
                          self._dn, self.device)
        self.model = model
        self.poses, self.V = _poses12(c2w)
        self.n_real = len(range(self.rank, self.tiles_total, self.world))   # tiles that start inside the image; the rest is padding
        self.opts.out_rgbd = 1                                               # the kernel writes [r,g,b,depth] rows: no packing pass
        with torch.cuda.device(self.device):
            self.buf = torch.zeros((self.V, self.per_rank * self.tile_rays, 4), dtype=torch.float32, device=self.device)

    @property
    def views_per_launch(self):
        """8 views per launch when the tiles deal evenly over the ranks (every rank's buffer is all real tiles), else one."""
        return 8 if self.n_real == self.per_rank else 1

    @property
    def launches_per_step(self):
        return 0 if self.n_real == 0 else -(-self.V // self.views_per_launch)

    @property
    def rays_per_step(self):
        """Rays this rank marches per `launch()`: all views x its real tiles."""
        return self.V * self.n_real * self.tile_rays

    @property
    def rays_per_launch(self):
        return min(self.views_per_launch, self.V) * self.n_real * self.tile_rays

    def launch(self):
        """Enqueue the render kernel(s): they write this rank's tiles straight into the gather buffer `self.buf`
        (V, per_rank*tile_rays, 4).  One launch per 8 views when the tiles deal evenly (every rank's buffer is all real tiles);
        otherwise one launch per view, each into the real-tile prefix of that view's rows (the padding rows stay zero)."""
        h = self.model.handle(self.device, self.mma_mode)
        n = self.n_real * self.tile_rays
        if n == 0:
            return
        even = self.n_real == self.per_rank
        step = self.views_per_launch
        seed0 = int(self.opts.rng_seed)
        try:
            with torch.cuda.device(self.device):
                for v0 in range(0, self.V, step):
                    nv = min(step, self.V - v0)
                    sub = (C.c_float * (12 * nv)).from_buffer(self.poses, 4 * 12 * v0)
                    out = self.buf[v0:v0 + nv] if even else self.buf[v0, :n]
                    # the kernel keys a view's jitter by seed + (camera index inside the launch) * 0x51ED27: offset the seed by the
                    # launch's first view so that the pattern is a function of the GLOBAL view index, however the views are batched
                    self.opts.rng_seed = (seed0 + v0 * 0x51ED27) & 0xFFFFFFFFFFFFFFFF
                    L.check(L.lib().nrf_render_cameras_tiles(h, self.H, self.W, self.focal, C.cast(sub, C.c_void_p), nv, self.tile_rays,
                                                             self.rank, self.world, self.n_real, C.byref(self.opts),
                                                             L.ptr(out), None, None, None, L.stream_ptr()))
        finally:
            self.opts.rng_seed = seed0

    def pack(self):
        """The gather buffer (kept for callers of the round-1 interface: the kernel has already written it)."""
        return self.buf


class OverlappedGather:
    """The exchange of a render loop, double-buffered: the all_gather of step i runs (on the backend's own stream / thread) while
    step i+1 renders into the OTHER buffer.  `bufs` are the per-slot local buffers the renders write (one: serial exchange, two:
    overlapped); `step(render)` waits for the gather that last read the slot's buffer, calls `render(slot)` -- which must write
    `bufs[slot]` -- and enqueues the asynchronous all_gather of that buffer.  A buffer is therefore never rewritten before the
    collective that reads it has completed, and `gathered(slot)` is valid once `wait(slot)` / `drain()` returned.  Every rank ends
    every step holding every rank's rows: (world,) + buf.shape.  Used by bench.py over RCCL; tests/test_tiles_gloo.py drives the
    same object over gloo on CPU tensors with two ranks and checks every step's rows."""

    def __init__(self, bufs, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.collective = dist.is_initialized()            # a one-rank world still goes through the backend (tests/rccl_worker.py: RCCL)
        self.world = dist.get_world_size(group) if self.collective else 1
        self.bufs = list(bufs)
        self.out = [torch.empty((self.world,) + tuple(b.shape), dtype=b.dtype, device=b.device) for b in self.bufs]
        self.pending = [None] * len(self.bufs)
        self.i = 0

    def wait(self, slot):
        if self.pending[slot] is not None:
            self.pending[slot].wait()
            self.pending[slot] = None

    def step(self, render):
        slot = self.i % len(self.bufs)
        self.wait(slot)                                    # the previous gather out of this slot's buffer
        render(slot)
        if self.collective:
            self.pending[slot] = self.dist.all_gather_into_tensor(self.out[slot].view(-1), self.bufs[slot].view(-1), group=self.group, async_op=True)
        else:
            self.out[slot][0].copy_(self.bufs[slot])
        self.i += 1
        return slot

    def drain(self):
        for slot in range(len(self.bufs)):
            self.wait(slot)

    def gathered(self, slot):
        return self.out[slot]


def render_tiles(model, H, W, focal, c2w, near, far, N_samples, rank, world, tile_rays, **kw):
    """This rank's tiles of a batch of V views (c2w (V,4,4), or one (4,4) pose) -> (V, per_rank*tile_rays, 4) = [r,g,b,depth].
    Up to 8 views go into one kernel launch."""
    job = TileJob(model, H, W, focal, c2w, near, far, N_samples, rank, world, tile_rays, **kw)
    job.launch()
    return job.pack()


def gather_frames(local: torch.Tensor, n_rays: int, tile_rays: int, group=None, world=None) -> torch.Tensor:
    """local (V, per_rank*tile_rays, C) on every rank -> (V, n_rays, C) on every rank with ONE all_gather.
    world=1: the caller rendered every tile itself (no collective, whatever process group happens to be initialised)."""
    import torch.distributed as dist
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    V = local.shape[0]
    if world == 1:
        g = local[None]
    else:
        g = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(g.view(-1), local.contiguous().view(-1), group=group)
    return torch.stack([reassemble(g[:, v], n_rays, world, tile_rays) for v in range(V)])


def render_frame_sharded(model, H, W, focal, c2w, near, far, N_samples=64, tile_rows=16, group=None, **kw):
    """Full frame(s) on every rank: (rgb (H,W,3), depth (H,W)) -- with a leading view axis if c2w is a batch.
    One launch (per 8 views) + one all_gather per rank."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    tile_rays = int(tile_rows) * int(W)
    local = render_tiles(model, H, W, focal, c2w, near, far, N_samples, rank, world, tile_rays, **kw)
    frames = gather_frames(local, int(H) * int(W), tile_rays, group)
    rgb, depth = frames[..., :3].reshape(-1, H, W, 3), frames[..., 3].reshape(-1, H, W)
    if torch.as_tensor(c2w).dim() == 2:
        return rgb[0], depth[0]
    return rgb, depth
