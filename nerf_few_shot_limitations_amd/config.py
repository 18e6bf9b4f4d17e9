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
