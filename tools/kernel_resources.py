"""Registers / spills / scratch of every kernel of the last build (hipcc's kernel-resource-usage remarks, kept per object by
build.py) as a table:   python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_few_shot_limitations_amd import build as B  # noqa: E402

res = B.kernel_resources()
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'spillV':>6} {'scratch':>7} {'occ':>3}  kernel")
for name in sorted(res):
    r = res[name]
    short = name.replace("nrf::", "").replace("(anonymous namespace)::", "")
    print(f"{r.get('vgprs', -1):5d} {r.get('agprs', -1):5d} {r.get('sgprs', -1):5d} {r.get('vgpr_spill', -1):6d} {r.get('scratch', -1):7d} {r.get('occupancy', -1):3d}  {short[:200]}")
