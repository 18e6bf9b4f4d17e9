"""Time the staged V1 MLP forward (nrf_mlp_forward_v1) on P already-encoded samples: a same-box probe for MLP-walk variants
(NRF_LIB=<variant .so>).  Prints ms per launch and the credited PFLOP/s."""
import sys
import torch
from nerf_few_shot_limitations_amd import NeRFMLP

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 800 * 800 * 16
torch.manual_seed(0)
m = NeRFMLP(mma_mode=mode).to("cuda").eval()
for p in m.parameters():
    p.requires_grad_(False)
x = torch.rand(P, 63, device="cuda") * 2 - 1
ref = None
with torch.no_grad():
    for _ in range(2):
        out = m(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        out = m(x)
    b.record()
    torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
flops = 2 * (63 * 256 + 3 * 256 * 256 + 319 * 256 + 3 * 256 * 256 + 256 * 4) * P
print(mode, P, f"{ms:.3f} ms", f"{flops / ms / 1e12:.3f} PF", "checksum", float(out.double().sum()))
