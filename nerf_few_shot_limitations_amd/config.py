"""experiments/*.yaml -> renderer/model settings (the keys src/training/train.py reads;
SURVEY.md section 5 'config / flags')."""
from __future__ import annotations

import yaml

from .nerf_model import NeRFMLP


def load_config(path):
    with open(path, "r") as f:
        return yaml.safe_load(f)                       # train.py:397-398


def resolve_near_far(cfg):
    """train.py:192-193 reads top-level near/far, which baseline.yaml / dino_nerf.yaml only define under
    `rendering:` / `data:`: resolve top-level -> rendering -> data."""
    for scope in (cfg, cfg.get("rendering", {}) or {}, cfg.get("data", {}) or {}):
        if "near" in scope and "far" in scope:
            return float(scope["near"]), float(scope["far"])
    raise KeyError("near/far not found in config (top level, rendering:, data:)")


def eval_samples(cfg):
    """N_samples of the eval loop: training.progressive_schedule.epochs_100_plus[2] (train.py:316)."""
    return int(cfg["training"]["progressive_schedule"]["epochs_100_plus"][2])


def model_from_config(cfg, dino_dim=64, mma_mode="f32"):
    """NeRFMLP as train.py:81-89 builds it."""
    nc = cfg["nerf_model"]
    use_dino = bool(cfg.get("model", {}).get("use_dino", True))
    return NeRFMLP(pos_freq=nc["pos_freq"], dir_freq=nc["dir_freq"], hidden_dim=nc["hidden_dim"],
                   num_density_layers=nc["num_layers"], use_dino=use_dino, dino_dim=dino_dim if use_dino else 0, mma_mode=mma_mode)


def render_settings(cfg):
    r = cfg.get("rendering", {}) or {}
    near, far = resolve_near_far(cfg)
    return dict(near=near, far=far, chunk_size=int(r.get("chunk_size", 1024)), white_bkgd=bool(r.get("white_bkgd", False)),
                n_samples=eval_samples(cfg))


def dino_model_from_config(cfg, weights=None, backbone_config=None):
    """The feature extractor as train.py:57-75 builds it: MultiScaleDINOFeatures for `model.dino_model_type: multi_scale`, else
    SpatialDINOFeatures(image_size=data.resolution); `dino_model: {name, lora_rank, lora_alpha, use_lora}`.  `weights`: a local
    transformers Dinov2Model checkpoint (directory or file); None -> random init with a warning (no network here)."""
    from .dino_feature_model import MultiScaleDINOFeatures, SpatialDINOFeatures
    dc = cfg.get("dino_model", {}) or {}
    kw = dict(model_name=dc.get("name", "facebook/dinov2-base"), lora_rank=int(dc.get("lora_rank", 16)), lora_alpha=int(dc.get("lora_alpha", 16)),
              use_lora=bool(dc.get("use_lora", True)), weights=weights, config=backbone_config)
    if cfg.get("model", {}).get("dino_model_type", "single_scale") == "multi_scale":
        return MultiScaleDINOFeatures(**kw)
    return SpatialDINOFeatures(image_size=int(cfg["data"]["resolution"]), **kw)


def precompute_dino_features(dino_model, images):
    """train.py:158-169: one map per training view, computed once under no_grad in eval mode.  `images`: (V,3,H,W) or (V,H,W,3) in
    [0,1]; normalised with the ImageNet statistics of train.py:128-131 -> (V,Hp,Wp,C) on the extractor's device."""
    import torch
    from .dino_backbone import IMAGENET_MEAN, IMAGENET_STD
    x = torch.as_tensor(images).float()
    if x.shape[-1] == 3 and x.shape[1] != 3:
        x = x.permute(0, 3, 1, 2)
    dev = next(dino_model.parameters()).device
    mean = torch.tensor(IMAGENET_MEAN, device=dev).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=dev).view(1, 3, 1, 1)
    was_training = dino_model.training
    dino_model.eval()
    maps = []
    with torch.no_grad():
        for v in range(x.shape[0]):
            maps.append(dino_model((x[v:v + 1].to(dev) - mean) / std))
    dino_model.train(was_training)
    return torch.cat(maps, 0)
