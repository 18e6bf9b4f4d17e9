"""nerf_few_shot_limitations_amd -- MI355X (gfx950) renderer for the ray-marching hot
path of ANKITSANJYAL/nerf-few-shot-limitations, behind the reference's own call
surface.  Everything numeric runs in libnerfhip.so
(hand-written HIP); this package is the thin host side.

(The directory is spelled with underscores because a Python package name cannot
contain '-'.)
"""
from .ray_sampler import get_rays, sample_points_along_rays, hierarchical_sampling, sample_pdf, get_ray_batch
from .positional_encoding import PositionalEncoding
from .nerf_model import NeRFMLP, load_checkpoint_into
from .volume_renderer import VolumeRenderer, volume_render_radiance
from .renderer import render_rays, render_camera, render_hierarchical, NeRFRenderer, make_dino
from .config import load_config, resolve_near_far, model_from_config, render_settings, dino_model_from_config, precompute_dino_features
from .data_loader import load_blender_data
from .evaluation import psnr, ssim, save_png, evaluate_views, evaluate_config
from .training import Adam, FusedStep, all_reduce_gradients
from .dino_features import project_points_to_image, sample_features_at_points
from .dino_feature_model import LoRALinear, SpatialDINOFeatures, MultiScaleDINOFeatures
from .dino_backbone import Dinov2Backbone, build_backbone, load_backbone_weights

__all__ = ["get_rays", "sample_points_along_rays", "hierarchical_sampling", "sample_pdf", "get_ray_batch",
           "PositionalEncoding", "NeRFMLP", "load_checkpoint_into", "VolumeRenderer", "volume_render_radiance",
           "render_rays", "render_camera", "render_hierarchical", "NeRFRenderer", "make_dino",
           "load_config", "resolve_near_far", "model_from_config", "render_settings",
           "load_blender_data", "psnr", "ssim", "save_png", "evaluate_views", "evaluate_config",
           "Adam", "FusedStep", "all_reduce_gradients", "project_points_to_image", "sample_features_at_points",
           "dino_model_from_config", "precompute_dino_features", "LoRALinear", "SpatialDINOFeatures", "MultiScaleDINOFeatures",
           "Dinov2Backbone", "build_backbone", "load_backbone_weights"]
