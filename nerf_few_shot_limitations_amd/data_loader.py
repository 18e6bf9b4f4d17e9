"""Blender-format dataset loader without torchvision (replaces
src/models/data_loader.py:8-64).  Same signature, same return values:

    images (N,3,H,W) float32 in [0,1], poses (N,4,4) float32, (H, W, focal)

`T.Resize(dims, interpolation=LANCZOS)` + `T.ToTensor()` of the reference are `PIL.Image.resize` + a uint8/255
conversion; focal = 0.5*W/tan(0.5*camera_angle_x)*focal_scale (data_loader.py:62).  The reference module cannot be
imported offline (it needs torchvision), so this restatement is unpinned by reference outputs; it is covered by a
procedurally generated fixture in tests/test_host_glue.py.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch
from PIL import Image


def load_blender_data(basedir, split="train", img_size=None, half_res=False):
    with open(os.path.join(basedir, f"transforms_{split}.json"), "r") as f:
        meta = json.load(f)
    images, poses = [], []
    focal_scale = 1.0
    for frame in meta["frames"]:
        img_path = os.path.join(basedir, frame["file_path"] + ".png")
        if not os.path.exists(img_path):
            raise FileNotFoundError(f"Image not found: {img_path}")
        img = Image.open(img_path).convert("RGB")                      # data_loader.py:33 (alpha dropped, not composited)
        W_orig, H_orig = img.size
        if img_size:
            dims, focal_scale = (img_size, img_size), img_size / W_orig   # :37-39
        elif half_res:
            dims, focal_scale = (H_orig // 2, W_orig // 2), 0.5            # :40-42
        else:
            dims, focal_scale = (H_orig, W_orig), 1.0
        if (dims[0], dims[1]) != (H_orig, W_orig):
            img = img.resize((dims[1], dims[0]), Image.LANCZOS)            # T.Resize takes (h, w), PIL takes (w, h)
        arr = np.asarray(img, dtype=np.uint8)
        images.append(torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div(255.0))   # T.ToTensor()
        poses.append(torch.from_numpy(np.array(frame["transform_matrix"], dtype=np.float32)))
    images = torch.stack(images)
    poses = torch.stack(poses)
    _, _, H, W = images.shape
    focal = 0.5 * W / np.tan(0.5 * meta["camera_angle_x"]) * focal_scale
    return images, poses, (H, W, float(focal))
