"""Worker of tests/test_gpu_parity.py::test_rccl_backend_single_rank: the bench's gather sequence on the REAL collective backend
("nccl" = RCCL) with a world of one rank -- two ranks cannot share the test box's one GPU under RCCL ("Duplicate GPU detected"),
but a one-rank world still goes through RCCL's communicator setup, the asynchronous all_gather_into_tensor on RCCL's stream
(tiles.OverlappedGather: the object bench.py's N > 1 loop uses), work.wait() and the barrier / all_reduce calls bench.py makes.  Prints 'rccl ok' when every step's frame equals render_camera."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import nerf_few_shot_limitations_amd as N                                     # noqa: E402
from nerf_few_shot_limitations_amd import tiles                               # noqa: E402
from oracle import nerf_oracle as O                                           # noqa: E402  (input generators only)


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", device_id=dev)
    world, rank = dist.get_world_size(), dist.get_rank()
    assert world == 1
    H, W, S = 48, 40, 16
    c2w = torch.from_numpy(O.LEGO_LIKE_C2W.copy())
    m = N.NeRFMLP(pos_dim=63, hidden_dim=256, n_layers=8, mma_mode="bf16")
    m.load_state_dict(O.make_weights("v1", 0, "solid"))
    m = m.cuda().eval()
    ref_rgb, ref_depth = N.render_camera(m, H, W, O.focal_for(W), c2w, 2.0, 6.0, S)
    tile_rays = 4 * W
    jobs = [tiles.TileJob(m, H, W, O.focal_for(W), c2w, 2.0, 6.0, S, rank, world, tile_rays, device=dev) for _ in range(2)]
    ex = tiles.OverlappedGather([j.buf for j in jobs])                        # bench.py's overlapped exchange, the very object

    def render(slot):
        jobs[slot].buf.zero_()
        jobs[slot].launch()
    for i in range(6):
        ex.step(render)
    ex.drain()
    frames = [ex.gathered(k) for k in range(2)]
    torch.cuda.synchronize()
    dist.barrier()
    t = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 1.5
    for k in range(2):
        f = tiles.reassemble(frames[k][:, 0], H * W, world, tile_rays)
        assert torch.equal(f[:, :3], ref_rgb) and torch.equal(f[:, 3], ref_depth)
    g = tiles.gather_frames(jobs[0].buf, H * W, tile_rays)                    # the serial form (dist initialised, world 1)
    assert torch.equal(g[0, :, :3], ref_rgb)
    dist.destroy_process_group()
    print("rccl ok")


if __name__ == "__main__":
    main()
