"""Blender-format dataset loader: the step right before the hot path.

Interface of the reference's `load_blender_data` (src/models/data_loader.py:8-64) -- same arguments, same return values

    images (N,3,H,W) float32 in [0,1], poses (N,4,4) float32, (H, W, focal)

-- on a different pipeline: the split's geometry (target size, focal scale) is resolved once from the first frame, the PNGs are
decoded and LANCZOS-resized by a thread pool (PIL releases the GIL while decoding) straight into ONE preallocated uint8 block,
and the block is converted to float once (on `device`, if one is given: a single host-to-device copy of the uint8 data, a quarter
of the float bytes).  No torchvision.

What must agree with the reference, and does by construction: alpha is dropped, not composited (`.convert("RGB")`, :33);
`T.Resize((h, w), LANCZOS)` is `PIL.Image.resize((w, h), LANCZOS)`; `T.ToTensor()` is uint8 / 255;
focal = 0.5 * W / tan(0.5 * camera_angle_x) * focal_scale with focal_scale = img_size / W_orig, or 0.5 for half_res (:37-42,62).
The reference module needs torchvision and cannot be imported offline: outputs are unpinned by it; the behaviour above is
tested on a generated scene (tests/test_host_glue.py).
"""
from __future__ import annotations

import json
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
from PIL import Image


def _split_meta(basedir, split):
    with open(os.path.join(basedir, f"transforms_{split}.json"), "r") as f:
        meta = json.load(f)
    paths = [os.path.join(basedir, fr["file_path"] + ".png") for fr in meta["frames"]]
    for p in paths:
        if not os.path.exists(p):
            raise FileNotFoundError(f"Image not found: {p}")
    poses = np.asarray([fr["transform_matrix"] for fr in meta["frames"]], dtype=np.float32).reshape(-1, 4, 4)
    return paths, poses, float(meta["camera_angle_x"])


def _geometry(first_png, img_size, half_res):
    """(H, W, focal_scale) of the loaded images from the size of the stored ones (data_loader.py:35-44)."""
    with Image.open(first_png) as im:
        w0, h0 = im.size
    if img_size:
        return int(img_size), int(img_size), img_size / w0
    if half_res:
        return h0 // 2, w0 // 2, 0.5
    return h0, w0, 1.0


def _decode_into(block, i, path, H, W):
    with Image.open(path) as im:
        im = im.convert("RGB")
        if im.size != (W, H):
            im = im.resize((W, H), Image.LANCZOS)
        block[i] = np.asarray(im, dtype=np.uint8)


def load_blender_data(basedir, split="train", img_size=None, half_res=False, device=None, workers=8):
    paths, poses, angle = _split_meta(basedir, split)
    H, W, focal_scale = _geometry(paths[0], img_size, half_res)
    block = np.empty((len(paths), H, W, 3), dtype=np.uint8)
    with ThreadPoolExecutor(max_workers=max(1, min(workers, len(paths)))) as pool:
        list(pool.map(lambda a: _decode_into(block, a[0], a[1], H, W), enumerate(paths)))
    raw = torch.from_numpy(block)
    if device is not None:
        raw = raw.to(device)
    images = raw.permute(0, 3, 1, 2).float().div_(255.0).contiguous()
    focal = 0.5 * W / np.tan(0.5 * angle) * focal_scale
    return images, torch.from_numpy(poses), (H, W, float(focal))
