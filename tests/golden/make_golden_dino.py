#!/usr/bin/env python3
"""Golden vectors for SURVEY.md section 8 row f4 (DINO feature-map production), captured in the build container from

  * `transformers.Dinov2Model` (the architecture the reference fetches with from_pretrained) on a TINY random-init
    configuration -- the published checkpoints are not available offline, and a full-size random model would be a 340 MB
    fixture; the architecture code does not depend on the sizes;
  * the REFERENCE's own `SpatialDINOFeatures` / `MultiScaleDINOFeatures` forward code (src/models/dino_feature_model.py,
    multi_scale_dino.py) wrapped around that tiny backbone.  Their constructors call from_pretrained (a network fetch), so the
    objects are assembled field by field exactly as the constructors would (same sub-modules, their own `_inject_lora`),
    and `forward` is the reference's unmodified method.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_dino.py
Writes tests/golden/dino_extractors.npz: every parameter (by state_dict name), the inputs and the reference outputs.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("NERF_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, os.path.join(REF, "src", "models"))

from transformers import Dinov2Config, Dinov2Model            # noqa: E402
import models.dino_feature_model as ref_single               # noqa: E402
import models.multi_scale_dino as ref_multi                  # noqa: E402
from oracle import nerf_oracle as O                          # noqa: E402  (input generator only)

torch.set_grad_enabled(False)
TINY = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=2, mlp_ratio=4, image_size=56, patch_size=14)


def tiny_backbone(seed):
    torch.manual_seed(seed)
    m = Dinov2Model(Dinov2Config(**TINY)).eval()
    # the default init leaves position embeddings / cls token / layer scales nearly trivial: randomise so that an indexing or
    # ordering error cannot hide
    for name, p in m.named_parameters():
        if "lambda1" in name:
            p.copy_(0.5 + torch.rand_like(p))
        elif p.dim() <= 3 and "projection" not in name and ("bias" in name or "embeddings" in name or "token" in name):
            p.copy_(0.2 * torch.randn_like(p))
    return m


def randomise_lora(module, seed):
    g = torch.Generator().manual_seed(seed)
    for name, p in module.named_parameters():
        if "lora_B" in name:                                  # zero at init: the wrapper would be invisible
            p.copy_(0.05 * torch.randn(p.shape, generator=g))


def assemble_single(seed, image_size, pos_embed_dim, rank, alpha):
    cls = ref_single.SpatialDINOFeatures
    obj = cls.__new__(cls)
    nn.Module.__init__(obj)
    obj.processor = None
    obj.backbone = tiny_backbone(seed)                        # dino_feature_model.py:39 with the fetch replaced
    for p in obj.backbone.parameters():
        p.requires_grad = False
    obj.patch_size = obj.backbone.config.patch_size
    obj.embed_dim = obj.backbone.config.hidden_size
    obj._inject_lora(rank, alpha)                             # the reference's own method (:68-76)
    side = image_size // obj.patch_size
    torch.manual_seed(seed + 1)
    obj.spatial_pos_embed = nn.Parameter(torch.randn(1, side * side, pos_embed_dim))                 # :56
    obj.feature_proj = nn.Sequential(nn.Linear(obj.embed_dim + pos_embed_dim, 256), nn.ReLU(inplace=True),   # :59-65
                                     nn.Linear(256, 128), nn.ReLU(inplace=True), nn.Linear(128, 64))
    obj.output_dim = 64
    randomise_lora(obj, seed + 2)
    return obj.eval()


def assemble_multi(seed, rank, alpha):
    cls = ref_multi.MultiScaleDINOFeatures
    obj = cls.__new__(cls)
    nn.Module.__init__(obj)
    obj.processor = None
    obj.backbone = tiny_backbone(seed)
    for p in obj.backbone.parameters():
        p.requires_grad = False
    obj.patch_size = obj.backbone.config.patch_size
    obj.embed_dim = obj.backbone.config.hidden_size
    obj._inject_lora(rank, alpha)                             # multi_scale_dino.py:52-60
    torch.manual_seed(seed + 1)
    obj.scales = [1, 2, 4]                                    # :28-49
    obj.feature_fusion = nn.ModuleDict({f"scale_{s}": nn.Sequential(nn.Linear(obj.embed_dim, 256), nn.ReLU(inplace=True), nn.Linear(256, 128))
                                        for s in obj.scales})
    obj.cross_scale_attention = nn.MultiheadAttention(embed_dim=128, num_heads=8, batch_first=True)
    obj.final_proj = nn.Sequential(nn.Linear(128 * 3, 256), nn.ReLU(inplace=True), nn.Linear(256, 128))
    obj.output_dim = 128
    randomise_lora(obj, seed + 2)
    return obj.eval()


def npf(t):
    return np.ascontiguousarray(t.detach().cpu().numpy())


def main():
    out = {}
    # 1. the backbone alone: native size (stored position embeddings) and a non-square larger input (bicubic resampling)
    bb = tiny_backbone(100)
    x56 = torch.from_numpy((O.uniform01(61, 2 * 3 * 56 * 56).reshape(2, 3, 56, 56) * 4 - 2).astype(np.float32))
    x_rect = torch.from_numpy((O.uniform01(62, 1 * 3 * 70 * 98).reshape(1, 3, 70, 98) * 4 - 2).astype(np.float32))
    out["bb_x56"], out["bb_y56"] = npf(x56), npf(bb(pixel_values=x56).last_hidden_state)
    out["bb_xrect"], out["bb_yrect"] = npf(x_rect), npf(bb(pixel_values=x_rect).last_hidden_state)
    # a size that is no multiple of the patch (the reference's 128 x 128 views are not): 60 x 58 -> 4 x 4 patches
    x_odd = torch.from_numpy((O.uniform01(64, 1 * 3 * 60 * 58).reshape(1, 3, 60, 58) * 4 - 2).astype(np.float32))
    out["bb_xodd"], out["bb_yodd"] = npf(x_odd), npf(bb(pixel_values=x_odd).last_hidden_state)
    for k, v in bb.state_dict().items():
        out["bb/" + k] = npf(v)
    # 2. SpatialDINOFeatures (reference forward), LoRA rank 4 / alpha 8, 56 x 56 images -> (B,4,4,64)
    single = assemble_single(200, image_size=56, pos_embed_dim=8, rank=4, alpha=8)
    out["single_x"], out["single_y"] = npf(x56), npf(single(x56))
    for k, v in single.state_dict().items():
        out["single/" + k] = npf(v)
    # 3. MultiScaleDINOFeatures (reference forward), 112 x 112 -> 8x8 / 4x4 / 2x2 patch grids -> (B,8,8,128)
    multi = assemble_multi(300, rank=4, alpha=8)
    x112 = torch.from_numpy((O.uniform01(63, 1 * 3 * 112 * 112).reshape(1, 3, 112, 112) * 4 - 2).astype(np.float32))
    out["multi_x"], out["multi_y"] = npf(x112), npf(multi(x112))
    for k, v in multi.state_dict().items():
        out["multi/" + k] = npf(v)
    path = os.path.join(HERE, "dino_extractors.npz")
    np.savez_compressed(path, **out)
    print(f"dino_extractors: {os.path.getsize(path) / 1024:.1f} KiB; shapes", out["bb_y56"].shape, out["bb_yrect"].shape, out["single_y"].shape, out["multi_y"].shape)


if __name__ == "__main__":
    main()
