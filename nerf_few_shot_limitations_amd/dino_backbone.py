"""DINOv2 vision transformer backbone (feature-map production).

The reference builds its feature extractors on `transformers.AutoModel.from_pretrained("facebook/dinov2-base")`
(src/models/dino_feature_model.py:38-39, multi_scale_dino.py:12-13) -- a network fetch.  This module is the backbone itself:
the published DINOv2 ViT (patch embedding, [CLS] token, bicubically interpolated position embeddings, pre-norm blocks with
LayerScale, final LayerNorm), written for PyTorch-ROCm with the parameter names of `transformers.Dinov2Model`, so that a
checkpoint of that model loads by name (`load_backbone_weights`) and the reference's LoRA injection
(`layer.attention.attention.{query,key,value}`, dino_feature_model.py:68-76) finds the same attribute path.

It runs ONCE per view, off the per-sample hot path (train.py:158-169 precomputes the maps under torch.no_grad): plain library
GEMMs (rocBLAS / hipBLASLt through torch) and torch's fused attention are the right tools here; the hand-written HIP kernels
of this package start where the map is sampled (csrc/nets.hpp:dino_taps).

Weights: the DINOv2 checkpoints are not available offline.  Without `weights=` the backbone is randomly
initialised and says so: everything downstream (shapes, LoRA wrappers, projection heads, the sampling kernels) is exercised,
image quality is not -- lego PSNR parity of the DINO variants stays unpinned.
"""
from __future__ import annotations

import os
import warnings
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

# hidden size, layers, heads of the published checkpoints (patch 14, image 518, mlp ratio 4)
KNOWN = {
    "facebook/dinov2-small": dict(hidden_size=384, num_hidden_layers=12, num_attention_heads=6),
    "facebook/dinov2-base": dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12),
    "facebook/dinov2-large": dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16),
}


def dinov2_config(model_name="facebook/dinov2-base", **overrides):
    cfg = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, mlp_ratio=4, patch_size=14, image_size=518,
               num_channels=3, layer_norm_eps=1e-6, layerscale_value=1.0, qkv_bias=True)
    if model_name in KNOWN:
        cfg.update(KNOWN[model_name])
    elif model_name is not None and not overrides:
        raise ValueError(f"unknown DINOv2 variant {model_name!r}: pass config=dict(hidden_size=..., num_hidden_layers=..., num_attention_heads=...)")
    cfg.update(overrides)
    return SimpleNamespace(**cfg)


class _PatchEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.projection = nn.Conv2d(cfg.num_channels, cfg.hidden_size, kernel_size=cfg.patch_size, stride=cfg.patch_size)

    def forward(self, x):
        return self.projection(x).flatten(2).transpose(1, 2)                  # (B, Hp*Wp, D), row-major patches


class _Embeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        n = (cfg.image_size // cfg.patch_size) ** 2
        self.cls_token = nn.Parameter(torch.randn(1, 1, cfg.hidden_size))
        self.mask_token = nn.Parameter(torch.zeros(1, cfg.hidden_size))       # pre-training only; kept so that checkpoints load strictly
        self.position_embeddings = nn.Parameter(torch.randn(1, n + 1, cfg.hidden_size))
        self.patch_embeddings = _PatchEmbeddings(cfg)
        self.patch_size = cfg.patch_size

    def position_encoding(self, n_patches, height, width):
        n_pos = self.position_embeddings.shape[1] - 1
        if n_patches == n_pos and height == width:
            return self.position_embeddings
        # other input sizes: the patch grid of position embeddings is resampled bicubically (DINOv2's own recipe)
        cls_pos, patch_pos = self.position_embeddings[:, :1], self.position_embeddings[:, 1:]
        dim, side = patch_pos.shape[-1], int(round(n_pos ** 0.5))
        grid = patch_pos.reshape(1, side, side, dim).permute(0, 3, 1, 2).to(torch.float32)
        grid = F.interpolate(grid, size=(height // self.patch_size, width // self.patch_size), mode="bicubic", align_corners=False)
        return torch.cat([cls_pos, grid.to(patch_pos.dtype).permute(0, 2, 3, 1).reshape(1, -1, dim)], 1)

    def forward(self, pixel_values):
        b, _, height, width = pixel_values.shape
        x = self.patch_embeddings(pixel_values.to(self.patch_embeddings.projection.weight.dtype))
        x = torch.cat([self.cls_token.expand(b, -1, -1), x], 1)
        return x + self.position_encoding(x.shape[1] - 1, height, width)


class _SelfAttention(nn.Module):
    """query / key / value projections by name: the reference replaces them with LoRALinear wrappers in place."""

    def __init__(self, cfg):
        super().__init__()
        d = cfg.hidden_size
        self.num_heads = cfg.num_attention_heads
        self.query = nn.Linear(d, d, bias=cfg.qkv_bias)
        self.key = nn.Linear(d, d, bias=cfg.qkv_bias)
        self.value = nn.Linear(d, d, bias=cfg.qkv_bias)

    def forward(self, x):
        b, n, d = x.shape
        hd = d // self.num_heads

        def heads(t):
            return t.view(b, n, self.num_heads, hd).transpose(1, 2)
        o = F.scaled_dot_product_attention(heads(self.query(x)), heads(self.key(x)), heads(self.value(x)))    # softmax(q k^T / sqrt(hd)) v
        return o.transpose(1, 2).reshape(b, n, d)


class _AttentionOutput(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.hidden_size)

    def forward(self, x):
        return self.dense(x)


class _Attention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.attention = _SelfAttention(cfg)
        self.output = _AttentionOutput(cfg)

    def forward(self, x):
        return self.output(self.attention(x))


class _LayerScale(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.lambda1 = nn.Parameter(cfg.layerscale_value * torch.ones(cfg.hidden_size))

    def forward(self, x):
        return x * self.lambda1


class _MLP(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.fc1 = nn.Linear(cfg.hidden_size, cfg.hidden_size * cfg.mlp_ratio)
        self.fc2 = nn.Linear(cfg.hidden_size * cfg.mlp_ratio, cfg.hidden_size)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class _Layer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.norm1 = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.attention = _Attention(cfg)
        self.layer_scale1 = _LayerScale(cfg)
        self.norm2 = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.mlp = _MLP(cfg)
        self.layer_scale2 = _LayerScale(cfg)

    def forward(self, x):
        x = x + self.layer_scale1(self.attention(self.norm1(x)))
        return x + self.layer_scale2(self.mlp(self.norm2(x)))


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(cfg) for _ in range(cfg.num_hidden_layers)])

    def forward(self, x):
        for blk in self.layer:
            x = blk(x)
        return x


class Dinov2Backbone(nn.Module):
    """`backbone(pixel_values=x).last_hidden_state` -> (B, 1 + Hp*Wp, D): [CLS] first, patches row-major (what
    dino_feature_model.py:93-94 and multi_scale_dino.py:88-89 consume)."""

    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.embeddings = _Embeddings(cfg)
        self.encoder = _Encoder(cfg)
        self.layernorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.apply(self._init)
        nn.init.trunc_normal_(self.embeddings.position_embeddings, std=0.02)
        nn.init.trunc_normal_(self.embeddings.cls_token, std=0.02)

    @staticmethod
    def _init(m):
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)

    def forward(self, pixel_values):
        # sizes that are not a multiple of the patch size lose their right / bottom remainder to the strided patch convolution,
        # as in the checkpointed model: the reference feeds 128 x 128 views -> 9 x 9 patches (dino_feature_model.py:54-56)
        if min(pixel_values.shape[-2:]) < self.config.patch_size:
            raise ValueError(f"image {tuple(pixel_values.shape[-2:])} is smaller than one {self.config.patch_size}-pixel patch")
        x = self.layernorm(self.encoder(self.embeddings(pixel_values)))
        return SimpleNamespace(last_hidden_state=x, pooler_output=x[:, 0])


def load_backbone_weights(backbone: Dinov2Backbone, path: str):
    """Load a `transformers.Dinov2Model` checkpoint by name: a directory holding model.safetensors / pytorch_model.bin, or
    such a file.  Only loaders that execute nothing from the file are used (safetensors, torch.load(weights_only=True))."""
    if os.path.isdir(path):
        for name in ("model.safetensors", "pytorch_model.bin"):
            if os.path.exists(os.path.join(path, name)):
                path = os.path.join(path, name)
                break
        else:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
    sd = {k[len("dinov2."):] if k.startswith("dinov2.") else k: v for k, v in sd.items()}
    missing, unexpected = backbone.load_state_dict(sd, strict=False)
    bad = [k for k in missing if not k.endswith("mask_token")]
    if bad or unexpected:
        raise RuntimeError(f"checkpoint does not match the backbone: missing {bad[:5]}, unexpected {list(unexpected)[:5]}")
    return backbone


def build_backbone(model_name="facebook/dinov2-base", weights=None, config=None):
    cfg = dinov2_config(model_name, **(config or {}))
    bb = Dinov2Backbone(cfg)
    if weights is not None:
        load_backbone_weights(bb, weights)
    else:
        warnings.warn(f"DINOv2 backbone {model_name!r}: no weights given (the published checkpoints are not available offline) -- "
                      "randomly initialised; pass weights=<dir or file of a transformers Dinov2Model checkpoint>", stacklevel=3)
    return bb


# ---- AutoImageProcessor of the DINOv2 checkpoints, for callers that hand over PIL images (multi_scale_dino.py:62-64,
#      train_multiscale.py:120): shortest edge -> 256 (bicubic), centre crop 224, 1/255, ImageNet mean / std
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def preprocess_pil(images, shortest_edge=256, crop=224):
    from PIL import Image
    import numpy as np
    out = []
    for im in images:
        im = im.convert("RGB")
        w, h = im.size
        s = shortest_edge / min(w, h)
        nw, nh = (shortest_edge, int(h * s)) if w <= h else (int(w * s), shortest_edge)
        im = im.resize((nw, nh), Image.BICUBIC)
        left, top = (nw - crop) // 2, (nh - crop) // 2
        im = im.crop((left, top, left + crop, top + crop))
        a = torch.from_numpy(np.asarray(im, dtype=np.float32) / 255.0).permute(2, 0, 1)
        out.append((a - torch.tensor(IMAGENET_MEAN).view(3, 1, 1)) / torch.tensor(IMAGENET_STD).view(3, 1, 1))
    return torch.stack(out)
