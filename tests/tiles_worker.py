"""Worker of tests/test_gpu_parity.py::test_two_process_tile_render_and_gather: one rank of a 2-rank pixel-tile render.
Launched with torch.distributed.run; every rank runs TileJob.launch (the kernel writes its tiles straight into the gather
buffer) -> gather_frames (ONE all_gather) and saves the frames it ends up holding.  gloo moves the bytes (the test box has
one GPU, which both ranks share; on a multi-GPU node NERF_TEST_BACKEND=nccl runs the same code over RCCL)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import nerf_few_shot_limitations_amd as N                                     # noqa: E402
from nerf_few_shot_limitations_amd import tiles                               # noqa: E402
from oracle import nerf_oracle as O                                           # noqa: E402  (input generators only)


def scene():
    c2w = torch.from_numpy(O.LEGO_LIKE_C2W.copy())
    c2w_b = c2w.clone()
    c2w_b[0, 3] += 0.3
    return torch.stack([c2w, c2w_b])


def main():
    out_dir, H, W, S, tile_rows, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    backend = os.environ.get("NERF_TEST_BACKEND", "gloo")
    dist.init_process_group(backend=backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0 if backend == "gloo" else int(os.environ.get("LOCAL_RANK", "0")))
    model = N.NeRFMLP(pos_dim=63, hidden_dim=256, n_layers=8, mma_mode=mode)
    model.load_state_dict(O.make_weights("v1", 0, "solid"))
    model = model.cuda().eval()
    poses = scene()
    tile_rays = tile_rows * W
    job = tiles.TileJob(model, H, W, O.focal_for(W), poses, 2.0, 6.0, S, rank, world, tile_rays)
    job.launch()
    local = job.pack()
    frames = tiles.gather_frames(local if backend != "gloo" else local.cpu(), H * W, tile_rays)      # (V, H*W, 4) on every rank
    np.save(os.path.join(out_dir, f"frames_rank{rank}.npy"), frames.cpu().numpy())
    # the one-call surface: render_frame_sharded picks rank / world from the process group
    if backend != "gloo":
        rgb, depth = tiles.render_frame_sharded(model, H, W, O.focal_for(W), poses[0], 2.0, 6.0, S, tile_rows=tile_rows)
        np.save(os.path.join(out_dir, f"sharded_rank{rank}.npy"), torch.cat([rgb.reshape(-1, 3), depth.reshape(-1, 1)], -1).cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
