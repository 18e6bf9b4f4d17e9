"""N>1 path on CPU: the pixel-tile partition and the all_gather reassembly (gloo, world_size 2 and 3),
with ray ids standing in for rendered pixels -- the integer contract of SURVEY.md section 8e."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerf_few_shot_limitations_amd import tiles


def test_tile_plan_covers_every_ray_exactly_once():
    for n_rays, world, tile_rays in [(100, 1, 16), (640000, 8, 12800), (1000, 3, 64), (37 * 53, 4, 53 * 5), (7, 8, 3)]:
        seen = torch.zeros(n_rays, dtype=torch.int64)
        total, per_rank = tiles.tile_plan(n_rays, world, tile_rays)
        assert per_rank * world >= total
        for r in range(world):
            ids = tiles.local_ray_ids(r, world, n_rays, tile_rays)
            assert ids.shape[0] == per_rank * tile_rays
            raw = (r + torch.arange(per_rank)[:, None] * world) * tile_rays + torch.arange(tile_rays)[None, :]
            real = raw.reshape(-1) < n_rays
            seen.index_add_(0, ids[real], torch.ones(int(real.sum()), dtype=torch.int64))
        assert torch.all(seen == 1)
        # reassemble() inverts the partition
        g = torch.stack([tiles.local_ray_ids(r, world, n_rays, tile_rays) for r in range(world)]).unsqueeze(-1).float()
        frame = tiles.reassemble(g, n_rays, world, tile_rays)
        assert torch.equal(frame[:, 0].long(), torch.arange(n_rays))


def _worker(rank, world, port, n_rays, tile_rays, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ids = tiles.local_ray_ids(rank, world, n_rays, tile_rays)
        local = torch.stack([ids.float(), ids.float() * 2, ids.float() + 0.5, -ids.float()], -1)     # "rendered" [r,g,b,depth]
        frame = tiles.gather_frame(local, n_rays, tile_rays)
        want = torch.arange(n_rays).float()
        ok = (frame.shape == (n_rays, 4) and torch.equal(frame[:, 0], want) and torch.equal(frame[:, 1], want * 2)
              and torch.equal(frame[:, 3], -want))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n_rays,tile_rays", [(2, 40 * 30, 30 * 4), (3, 1000, 64)])
def test_gather_frame_gloo(world, n_rays, tile_rays):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rays, tile_rays, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(world))
    assert res == {r: True for r in range(world)}


# ---- data-parallel training: the gradient exchange (training.py), rehearsed on CPU tensors -------------------------
def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerf_few_shot_limitations_amd.training import _all_reduce_mean
        g = torch.arange(1000, dtype=torch.float32) * (rank + 1)          # this rank's flat gradient vector
        _all_reduce_mean(g, None)
        want = torch.arange(1000, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        s = torch.arange(8, dtype=torch.float32) + rank
        _all_reduce_mean(s, None, average=False)
        q.put((rank, bool(torch.allclose(g, want)) and bool(torch.equal(s, torch.arange(8, dtype=torch.float32) * world + sum(range(world))))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_flat_gradient_all_reduce_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, True) for r in range(world)]


def _overlap_worker(rank, world, port, n_bufs, steps, q):
    """bench.py's render / exchange loop on CPU tensors with REAL asynchronous collectives: every step's render writes
    (rank, step, element index) into its slot's buffer; after the loop (and, for the slot about to be reused, during it) every rank
    must hold exactly that pattern from every rank -- a buffer rewritten before its gather finished, or a gather read before it
    landed, shows up as a wrong step number."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 4096
        bufs = [torch.zeros(3, n, 4) for _ in range(n_bufs)]
        ex = tiles.OverlappedGather(bufs)
        ok = True
        last_step_of_slot = {}

        def check(slot, step):
            g = ex.gathered(slot)
            good = g.shape == (world, 3, n, 4)
            for r in range(world):
                good = good and bool((g[r, :, :, 0] == r).all()) and bool((g[r, :, :, 1] == step).all())
                good = good and torch.equal(g[r, 0, :, 2], torch.arange(n, dtype=torch.float32))
            return good

        for step in range(steps):
            def render(slot, step=step):
                if slot in last_step_of_slot:                  # the gather that read this buffer has been waited for: its result is final
                    nonlocal ok
                    ok = ok and check(slot, last_step_of_slot[slot])
                b = bufs[slot]
                b[..., 0] = rank
                b[..., 1] = step
                b[..., 2] = torch.arange(n, dtype=torch.float32)
            slot = ex.step(render)
            last_step_of_slot[slot] = step
        ex.drain()
        for slot, step in last_step_of_slot.items():
            ok = ok and check(slot, step)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_bufs", [(2, 2), (2, 1), (3, 2)])
def test_overlapped_gather_protocol_gloo(world, n_bufs):
    """tiles.OverlappedGather (the double-buffered exchange of bench.py's N > 1 loop) with async all_gathers over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, n_bufs, 7, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(world)]
