"""Worker of tests/test_gpu_training.py::test_data_parallel_*: one rank of a 2-rank data-parallel training run.
Launched with torch.distributed.run; every rank trains on its half of a fixed ray batch and rank 0 writes the
parameters after the last step."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import nerf_few_shot_limitations_amd as N                                     # noqa: E402
from nerf_few_shot_limitations_amd.training import FusedStep, all_reduce_gradients   # noqa: E402
from oracle import nerf_oracle as O                                           # noqa: E402  (input generators only)


def batch(R, S):
    z = torch.sort(torch.from_numpy(O.uniform01(201, R * S).reshape(R, S) * 4 + 2).float(), dim=-1).values
    rd = torch.from_numpy(O.uniform01(202, R * 3).reshape(R, 3) - 0.5).float()
    tgt = torch.from_numpy(O.uniform01(203, R * 3).reshape(R, 3)).float()
    x = O.positional_encoding(torch.from_numpy(O.uniform01(204, R * S * 3).reshape(R * S, 3) * 4 - 2).float(), 10).reshape(R, S, 63)
    return x, z, rd, tgt


def main():
    route, out_path, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    dist.init_process_group(backend=os.environ.get("NERF_TEST_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)                       # the test box has one GPU: both ranks share it (gloo moves the bytes)
    R, S = 128, 16
    x, z, rd, tgt = batch(R, S)
    lo, hi = rank * R // world, (rank + 1) * R // world
    xs, zs, rds, tgts = x[lo:hi].reshape(-1, 63).cuda(), z[lo:hi].contiguous().cuda(), rd[lo:hi].contiguous().cuda(), tgt[lo:hi].contiguous().cuda()
    model = N.NeRFMLP(pos_dim=63, hidden_dim=256, n_layers=8, mma_mode="f32")
    model.load_state_dict(O.make_weights("v1", 0, "solid"))
    model = model.cuda().train()
    n = hi - lo
    if route == "fused":
        step = FusedStep(model, lr=1e-2, data_parallel=True)
        for _ in range(steps):
            step(xs, zs, rds, tgts)
    else:
        opt = torch.optim.SGD(model.parameters(), lr=1e-2)
        for _ in range(steps):
            opt.zero_grad()
            pred = N.volume_render_radiance(model(xs).view(n, 1, S, 4), zs.view(n, 1, S), rds.view(n, 1, 3)).view(n, 3)
            torch.nn.functional.mse_loss(pred, tgts).backward()
            all_reduce_gradients(model)
            opt.step()
    flat = model.flat_params().flat.detach().cpu().numpy()
    gathered = [None] * world
    dist.all_gather_object(gathered, float(np.abs(flat).sum()))
    if rank == 0:
        np.save(out_path, flat)
        assert all(abs(g - gathered[0]) < 1e-3 * abs(gathered[0]) for g in gathered), gathered
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
