"""The DINO feature extractors of the reference on PyTorch-ROCm:

  LoRALinear              src/models/dino_feature_model.py:7-32 (== multi_scale_dino.py:185-209, lora_dino.py)
  SpatialDINOFeatures     src/models/dino_feature_model.py:34-148   one (B,Hp,Wp,64) map per image  (dino_nerf.yaml, lora.yaml)
  MultiScaleDINOFeatures  src/models/multi_scale_dino.py:7-183      three scales fused to (B,Hp,Wp,128) (multiscale.yaml)

Same constructor arguments, attribute names and state_dict keys as the reference classes (a state_dict of theirs loads by
name), the same forward arithmetic; the backbone is dino_backbone.Dinov2Backbone instead of a `from_pretrained` fetch
(`weights=` names a local checkpoint; none -> random init, said so by a warning).  `sample_features_at_points` is the staged
HIP kernel (dino_features.py) instead of F.grid_sample.

How the trainer uses them (and what that means for gradients): `NeRFDINOTrainer.precompute_dino_features` (train.py:158-169)
and `MultiScaleNeRFDINOTrainer` (train_multiscale.py:114-120) run the extractor ONCE per training view under
`torch.no_grad()` and keep the maps; the LoRA matrices are handed to the optimizer (train.py:105-110) but no gradient ever
reaches them -- and with lora_B initialised to zero the wrappers add exactly 0.  The hot path therefore needs the maps as
constants, which is what the fused renderer and NeRFMLP.forward take (SURVEY.md defect ledger: this is D13).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .dino_backbone import build_backbone, preprocess_pil


class LoRALinear(nn.Module):
    """y = W x + (alpha / rank) * B(dropout(A x)); W frozen, B = 0 at init (dino_feature_model.py:7-32)."""

    def __init__(self, original_layer, rank=16, alpha=16, dropout=0.1):
        super().__init__()
        self.original = original_layer
        self.rank, self.alpha = rank, alpha
        self.scaling = alpha / rank
        for p in self.original.parameters():
            p.requires_grad = False
        self.lora_A = nn.Linear(original_layer.in_features, rank, bias=False)
        self.lora_B = nn.Linear(rank, original_layer.out_features, bias=False)
        self.dropout = nn.Dropout(dropout)
        nn.init.kaiming_uniform_(self.lora_A.weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_B.weight)

    @property
    def in_features(self):
        return self.original.in_features

    @property
    def out_features(self):
        return self.original.out_features

    def forward(self, x):
        return self.original(x) + self.scaling * self.lora_B(self.dropout(self.lora_A(x)))


def _inject_lora(backbone, rank, alpha):
    """Q, K, V projections of every block (dino_feature_model.py:68-76 == multi_scale_dino.py:52-60)."""
    for layer in backbone.encoder.layer:
        att = layer.attention.attention
        att.query = LoRALinear(att.query, rank=rank, alpha=alpha)
        att.key = LoRALinear(att.key, rank=rank, alpha=alpha)
        att.value = LoRALinear(att.value, rank=rank, alpha=alpha)


def _pixel_values(module, images):
    if isinstance(images, (list, tuple)):                                    # PIL images: the checkpoint's image processor
        return preprocess_pil(images).to(next(module.parameters()).device)
    return images                                                           # an already normalised (B,3,H,W) tensor (train.py:164-166)


def _sample(features, points_2d):
    from .dino_features import sample_features_at_points
    return sample_features_at_points(features, points_2d)


class SpatialDINOFeatures(nn.Module):
    """Patch features of one DINOv2 pass + a learned (Hp*Wp, 64) position table -> 3-layer projection to 64 channels."""

    def __init__(self, model_name="facebook/dinov2-base", use_lora=True, lora_rank=16, lora_alpha=16, image_size=128, pos_embed_dim=64,
                 weights=None, config=None):
        super().__init__()
        self.processor = None                                                # dino_backbone.preprocess_pil
        self.backbone = build_backbone(model_name, weights, config)
        for p in self.backbone.parameters():
            p.requires_grad = False
        self.patch_size = self.backbone.config.patch_size
        self.embed_dim = self.backbone.config.hidden_size
        if use_lora:
            _inject_lora(self.backbone, lora_rank, lora_alpha)
        side = image_size // self.patch_size
        self.spatial_pos_embed = nn.Parameter(torch.randn(1, side * side, pos_embed_dim))
        self.feature_proj = nn.Sequential(nn.Linear(self.embed_dim + pos_embed_dim, 256), nn.ReLU(inplace=True),
                                          nn.Linear(256, 128), nn.ReLU(inplace=True), nn.Linear(128, 64))
        self.output_dim = 64

    def forward(self, images):
        """(B,3,H,W) normalised tensor or list of PIL images -> (B, Hp, Wp, 64) (dino_feature_model.py:77-112)."""
        x = self.backbone(pixel_values=_pixel_values(self, images)).last_hidden_state[:, 1:, :]      # drop [CLS]
        b, n, d = x.shape
        side = int(math.sqrt(n))
        spatial = x.view(b, side, side, d)
        pos = self.spatial_pos_embed.view(1, side, side, -1).expand(b, -1, -1, -1)
        return self.feature_proj(torch.cat([spatial, pos], -1))

    def sample_features_at_points(self, features, points_2d):
        """dino_feature_model.py:114-148 on the staged HIP kernel."""
        return _sample(features, points_2d)


class MultiScaleDINOFeatures(nn.Module):
    """DINOv2 at input scales 1, 1/2, 1/4 -> per-scale 128-d projection -> self-attention inside each scale -> bilinear
    upsampling to the finest grid -> concatenation -> 128-d projection (multi_scale_dino.py:62-154)."""

    def __init__(self, model_name="facebook/dinov2-base", use_lora=True, lora_rank=16, lora_alpha=16, weights=None, config=None):
        super().__init__()
        self.processor = None
        self.backbone = build_backbone(model_name, weights, config)
        for p in self.backbone.parameters():
            p.requires_grad = False
        self.patch_size = self.backbone.config.patch_size
        self.embed_dim = self.backbone.config.hidden_size
        if use_lora:
            _inject_lora(self.backbone, lora_rank, lora_alpha)
        self.scales = [1, 2, 4]
        self.feature_fusion = nn.ModuleDict({f"scale_{s}": nn.Sequential(nn.Linear(self.embed_dim, 256), nn.ReLU(inplace=True), nn.Linear(256, 128))
                                             for s in self.scales})
        self.cross_scale_attention = nn.MultiheadAttention(embed_dim=128, num_heads=8, batch_first=True)
        self.final_proj = nn.Sequential(nn.Linear(128 * len(self.scales), 256), nn.ReLU(inplace=True), nn.Linear(256, 128))
        self.output_dim = 128

    def extract_multi_scale_features(self, images):
        pixel_values = _pixel_values(self, images)
        b = pixel_values.shape[0]
        out = {}
        for s in self.scales:
            x = pixel_values
            if s != 1:
                h, w = pixel_values.shape[2] // s, pixel_values.shape[3] // s
                x = F.interpolate(pixel_values, size=(h, w), mode="bilinear", align_corners=False)
            with torch.no_grad():                                            # multi_scale_dino.py:87: the backbone pass carries no gradient
                tok = self.backbone(pixel_values=x).last_hidden_state[:, 1:, :]
            side = int(math.sqrt(tok.shape[1]))
            out[s] = self.feature_fusion[f"scale_{s}"](tok.view(b, side, side, self.embed_dim))
        return out

    def fuse_multi_scale_features(self, multi_scale_features):
        b = next(iter(multi_scale_features.values())).shape[0]
        fused = []
        for s in self.scales:
            f = multi_scale_features[s]
            h, w = f.shape[1], f.shape[2]
            flat = f.reshape(b, h * w, 128)
            att, _ = self.cross_scale_attention(flat, flat, flat)            # attention among the tokens of ONE scale
            fused.append(att.view(b, h, w, 128))
        th, tw = fused[0].shape[1], fused[0].shape[2]
        aligned = [fused[0]] + [F.interpolate(f.permute(0, 3, 1, 2), size=(th, tw), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
                                for f in fused[1:]]
        return self.final_proj(torch.cat(aligned, -1))

    def forward(self, images):
        return self.fuse_multi_scale_features(self.extract_multi_scale_features(images))

    def sample_features_at_points(self, features, points_2d):
        """multi_scale_dino.py:156-183 on the staged HIP kernel."""
        return _sample(features, points_2d)
