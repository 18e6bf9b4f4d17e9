"""The DINO side channel as the reference's trainer calls it (src/training/train.py:203-214): project the sampled points
into the source view, then fetch that view's feature map at the projections.  Drop-ins for
`utils.ray_utils.project_points_to_image` (ray_utils.py:176-210) and
`SpatialDINOFeatures.sample_features_at_points` (dino_feature_model.py:114-148 == lora_dino.py:110-144 ==
multi_scale_dino.py:156-183) on libnerfhip's staged kernel; the fused renderer does both inside the kernel, the training
path (where the features are an input of NeRFMLP.forward) needs them as tensors.  The feature map itself comes from the
DINOv2 extractor, which is outside this package."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .renderer import make_dino


def project_points_to_image(points_3d, pose, focal, H, W):
    """(N,3) world points -> (points_2d (N,2) in [-1,1], depths (N,), valid_mask (N,)) for the (4,4) camera-to-world `pose`."""
    L.require_gpu()
    pts = L.dev_f32(points_3d).reshape(-1, 3)
    n = pts.shape[0]
    dev = pts.device
    dummy = torch.zeros((1, 1, 1, 1), dtype=torch.float32, device=dev)               # the kernel fetches as it projects: a 1x1x1 map
    d, keep = make_dino(dummy, pose, focal, H, W)
    with torch.cuda.device(dev):
        xy = torch.empty((n, 2), dtype=torch.float32, device=dev)
        feats = torch.empty((n, 1), dtype=torch.float32, device=dev)
        L.check(L.lib().nrf_project_fetch(C.byref(d), L.ptr(pts), n, L.ptr(feats), L.ptr(xy), L.stream_ptr()))
    # depth and the in-front mask are by-products the trainer discards (train.py:212 keeps the first output only)
    inv = torch.inverse(torch.as_tensor(pose).detach().to("cpu", torch.float32)).to(dev)
    depths = pts @ inv[2, :3] + inv[2, 3]
    return xy, depths, depths > 0


def sample_features_at_points(features, points_2d):
    """features (B,Hp,Wp,C) channel-last, points_2d (N,2) in [-1,1] -> (N,C) (B == 1) or (B,N,C): bilinear, zeros padding,
    align_corners=False."""
    L.require_gpu()
    fm = L.dev_f32(features)
    if fm.dim() != 4:
        raise ValueError("features must be (B,Hp,Wp,C)")
    xy = L.dev_f32(points_2d, fm.device).reshape(-1, 2)
    n = xy.shape[0]
    B, Hp, Wp, Cc = (int(v) for v in fm.shape)
    out = torch.empty((B, n, Cc), dtype=torch.float32, device=fm.device)
    with torch.cuda.device(fm.device):
        for b in range(B):
            L.check(L.lib().nrf_sample_features(L.ptr(fm[b]), Hp, Wp, Cc, L.ptr(xy), n, L.ptr(out[b]), L.stream_ptr()))
    return out[0] if B == 1 else out
