"""Render a seeded frame in one mode and print a hash of the (rgb, depth) bits: same-box check that a kernel variant
(NRF_LIB=<variant .so>) leaves every bit of the image alone.  usage: image_hash.py MODE [H W S] [ert_eps]"""
import hashlib
import sys
import torch
import nerf_few_shot_limitations_amd as N

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
H, W, S = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (200, 200, 64)
eps = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
torch.manual_seed(3)
m = N.NeRFMLP(pos_dim=63, hidden_dim=256, n_layers=8, mma_mode=mode).to("cuda").eval()
with torch.no_grad():
    for p in m.parameters():
        p.mul_(1.6)
pose = torch.eye(4)
pose[2, 3] = 4.0
with torch.no_grad():
    out = N.render_camera(m, H, W, 0.9 * W, pose.to("cuda"), near=2.0, far=6.0, N_samples=S, perturb=False, ert_eps=eps)
rgb, depth = out[0], out[1]
hsh = hashlib.sha256(rgb.contiguous().cpu().numpy().tobytes() + depth.contiguous().cpu().numpy().tobytes()).hexdigest()[:16]
print(mode, H, W, S, eps, hsh, float(rgb.double().mean()), float(depth.double().mean()))
