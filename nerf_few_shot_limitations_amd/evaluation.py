"""Evaluation glue right after the hot path: render test views with the fused renderer,
PSNR / SSIM, PNG dumps -- what NeRFDINOTrainer.evaluate does around render_rays (src/training/train.py:294-342) with
torchmetrics / imageio, neither of which is installed here.

PSNR = -10 log10(mse), data range 1 (train_multiscale.py:294-295).  SSIM follows the algorithm of the metric the reference
instantiates, `torchmetrics.StructuralSimilarityIndexMeasure()` with its defaults (train.py:100,328): single scale, 11x11
Gaussian window of sigma 1.5, K1=.01, K2=.03, reflect-padded inputs, the padded border cropped from the index map before the
mean, and -- because the reference passes no data_range -- the range taken from the data, max(pred range, target range).
`ssim(..., data_range=1.0, crop_border=False)` gives the textbook form.  torchmetrics is not importable here; tests/test_host_glue.py
checks this code against an independent scipy restatement of the same definition and against the one known answer available
offline, the library's own docstring example (uniform noise vs 0.75 x itself, data_range 1: 0.9219).  LPIPS
needs VGG weights that cannot be fetched offline and is not provided.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

from . import tiles
from .config import render_settings


def psnr(pred: torch.Tensor, target: torch.Tensor) -> float:
    mse = torch.mean((pred.double() - target.double()) ** 2).item()
    return float("inf") if mse == 0 else -10.0 * math.log10(mse)


def _gauss_window(size=11, sigma=1.5, device=None):
    x = torch.arange(size, dtype=torch.float32, device=device) - (size - 1) / 2
    g = torch.exp(-(x ** 2) / (2 * sigma ** 2))
    g = g / g.sum()
    return (g[:, None] * g[None, :])[None, None]


def ssim(pred: torch.Tensor, target: torch.Tensor, size=11, sigma=1.5, data_range=None, crop_border=True) -> float:
    """pred/target (H,W,3) or (3,H,W).  data_range=None: from the data, as the reference's default-constructed metric does."""
    def chw(t):
        t = t.float()
        return (t.permute(2, 0, 1) if t.shape[-1] == 3 and t.shape[0] != 3 else t)[None]
    a, b = chw(pred), chw(target).to(pred.device)
    c = a.shape[1]
    w = _gauss_window(size, sigma, a.device).expand(c, 1, size, size)
    pad = size // 2
    a_p, b_p = F.pad(a, (pad,) * 4, mode="reflect"), F.pad(b, (pad,) * 4, mode="reflect")
    mu_a, mu_b = F.conv2d(a_p, w, groups=c), F.conv2d(b_p, w, groups=c)
    s_aa = F.conv2d(a_p * a_p, w, groups=c) - mu_a ** 2
    s_bb = F.conv2d(b_p * b_p, w, groups=c) - mu_b ** 2
    s_ab = F.conv2d(a_p * b_p, w, groups=c) - mu_a * mu_b
    if data_range is None:
        data_range = float(torch.maximum(a.max() - a.min(), b.max() - b.min()).item())
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    m = ((2 * mu_a * mu_b + c1) * (2 * s_ab + c2)) / ((mu_a ** 2 + mu_b ** 2 + c1) * (s_aa + s_bb + c2))
    if crop_border and m.shape[-1] > 2 * pad and m.shape[-2] > 2 * pad:
        m = m[..., pad:-pad, pad:-pad]
    return float(m.mean().item())


def save_png(path: str, img: torch.Tensor) -> None:
    """(H,W,3) in [0,1] -> 8-bit PNG, as train.py:331-336 does with imageio."""
    arr = (img.detach().clamp(0, 1).cpu().numpy() * 255).astype(np.uint8)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(arr).save(path)


@torch.no_grad()
def evaluate_views(model, poses, H, W, focal, near, far, N_samples=64, targets=None, out_dir=None, white_bkgd=False,
                   mma_mode=None, ert_eps=0.0, dino=None, max_png=5):
    """Render every test pose (8 views per kernel launch) and, when `targets` (V,H,W,3|4) are given, score them.
    Returns {'images' (V,H,W,3), 'depth' (V,H,W), 'psnr', 'ssim' (means), 'per_view': [...]} (train.py:294-342)."""
    model.eval()
    poses = torch.as_tensor(poses)
    V = poses.shape[0]
    tile_rays = 16 * int(W)
    local = tiles.render_tiles(model, H, W, focal, poses, near, far, N_samples, 0, 1, tile_rays, white_bkgd=white_bkgd,
                               mma_mode=mma_mode, ert_eps=ert_eps, dino=dino)
    frames = tiles.gather_frames(local, int(H) * int(W), tile_rays, world=1)      # rendered with rank 0 of 1: no collective
    images = frames[..., :3].reshape(V, H, W, 3)
    depth = frames[..., 3].reshape(V, H, W)
    out = {"images": images, "depth": depth, "per_view": []}
    if targets is not None:
        for v in range(V):
            t = torch.as_tensor(targets[v]).to(images.device).float()
            if t.shape[-1] == 4:                                   # train.py:181-184: composite RGBA over white
                t = t[..., :3] * t[..., 3:4] + (1.0 - t[..., 3:4])
            out["per_view"].append({"psnr": psnr(images[v], t), "ssim": ssim(images[v], t)})
        out["psnr"] = float(np.mean([p["psnr"] for p in out["per_view"]]))
        out["ssim"] = float(np.mean([p["ssim"] for p in out["per_view"]]))
    if out_dir is not None:
        for v in range(min(V, max_png)):
            save_png(os.path.join(out_dir, f"render_{v}.png"), images[v])
    return out


def evaluate_config(model, cfg, poses, H, W, focal, **kw):
    """`evaluate` driven by an experiments/*.yaml dict: near/far, white_bkgd and the eval sample count come from it."""
    rs = render_settings(cfg)
    return evaluate_views(model, poses, H, W, focal, rs["near"], rs["far"], rs["n_samples"], white_bkgd=rs["white_bkgd"], **kw)
