#!/usr/bin/env python3
"""Times the staged (one-leaf-each) kernels at the headline shape and reports achieved GB/s against the HBM roofline
(8 TB/s spec, ~6.3 TB/s achievable: MI355X_MICROARCH.md).  They are HBM-bound elementwise / per-ray kernels; bytes are
the algorithmic sizes of the reference's tensors."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf_few_shot_limitations_amd as N
from oracle import nerf_oracle as O


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


H = W = 800
S = 64
R = H * W
c2w = torch.from_numpy(O.LEGO_LIKE_C2W.copy())
ro, rd = N.get_rays(H, W, O.focal_for(W), c2w)
ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
rows = []
ms = timeit(lambda: N.get_rays(H, W, O.focal_for(W), c2w)); rows.append(("get_rays", ms, R * 24))
ms = timeit(lambda: N.sample_points_along_rays(ro, rd, 2.0, 6.0, S, perturb=False)); rows.append(("sample (pts+z)", ms, R * 24 + R * S * 16))
pts, z = N.sample_points_along_rays(ro, rd, 2.0, 6.0, S, perturb=False)
pe = N.PositionalEncoding(10)
sub = pts.reshape(-1, 3)[: R * 8]
ms = timeit(lambda: pe(sub), 5); rows.append(("encode L=10 (1/8 frame)", ms, sub.shape[0] * (12 + 63 * 4)))
rgb = torch.rand(R, S, 3, device="cuda"); sig = torch.rand(R, S, 1, device="cuda") * 3
vr = N.VolumeRenderer().eval()
ms = timeit(lambda: vr(rgb, sig, z, rd)); rows.append(("composite (+weights)", ms, R * S * 24 + R * 28))
print(f"{'kernel':28s} {'ms':>8s} {'GB/s':>9s} {'% of 6.3 TB/s':>14s}")
for name, ms, nbytes in rows:
    gbs = nbytes / (ms * 1e-3) / 1e9
    print(f"{name:28s} {ms:8.3f} {gbs:9.1f} {100 * gbs / 6300:13.1f}%")
