"""python -m nerf_few_shot_limitations_amd.evaluate_cli --config experiments/baseline.yaml --data data/nerf_synthetic/lego \\
        [--checkpoint results/.../best.pth] [--split test] [--out results/eval] [--mode f16x3|f32|f16|bf16] [--max-views N]

What `NeRFDINOTrainer.evaluate` does (src/training/train.py:294-342) for a use_dino=False config, on the fused renderer:
load the YAML unchanged, the Blender split, the checkpoint (either key set), render every view (8 per launch), score
PSNR/SSIM, dump PNGs and a metrics.json.  Without --checkpoint the weights are the module's random init (smoke use).
use_dino configs condition every test view on the feature map and pose of TRAINING view 0 (train.py:203-208: `feat_idx = 0` outside
training).  The map comes from the config's extractor (config.dino_model_from_config) run once on that view -- --dino-weights
names a local transformers Dinov2Model checkpoint, --dino-random-init builds it with random weights (the published weights are
not available offline) -- or from --dino-map, a saved (1,Hp,Wp,C) tensor; without any of them the CLI refuses.
"""
from __future__ import annotations

import argparse
import json
import os

import torch

from . import (dino_model_from_config, evaluate_views, load_blender_data, load_checkpoint_into, load_config, model_from_config,
               precompute_dino_features, render_settings)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", required=True)
    ap.add_argument("--data", required=True, help="dataset directory holding transforms_<split>.json")
    ap.add_argument("--checkpoint")
    ap.add_argument("--split", default="test")
    ap.add_argument("--out", default=None)
    ap.add_argument("--mode", default="f16x3", choices=["f16x3", "f32", "f16", "bf16"],
                    help="f16x3 / f32: parity-grade (1e-4 of the reference); f16 / bf16: full MFMA rate")
    ap.add_argument("--max-views", type=int, default=None)
    ap.add_argument("--ert", type=float, default=0.0)
    ap.add_argument("--dino-map", default=None)
    ap.add_argument("--dino-weights", default=None, help="local transformers Dinov2Model checkpoint (dir or file) for the extractor of the config")
    ap.add_argument("--dino-random-init", action="store_true", help="build the extractor with random weights (pipeline runs, features meaningless)")
    args = ap.parse_args(argv)

    cfg = load_config(args.config)
    rs = render_settings(cfg)
    images, poses, (H, W, focal) = load_blender_data(args.data, args.split, img_size=cfg["data"].get("resolution"))
    if args.max_views:
        images, poses = images[: args.max_views], poses[: args.max_views]
    use_dino = bool(cfg.get("model", {}).get("use_dino", True))
    dino = None
    dino_dim = 128 if cfg.get("model", {}).get("dino_model_type") == "multi_scale" else 64
    if use_dino:
        # the source view of every evaluation render is TRAINING view 0: its map and its pose (train.py:203-208)
        tr_images, tr_poses, _ = load_blender_data(args.data, "train", img_size=cfg["data"].get("resolution"))
        if args.dino_map:
            fm = torch.load(args.dino_map, map_location="cpu", weights_only=True)
        elif args.dino_weights or args.dino_random_init:
            extractor = dino_model_from_config(cfg, weights=args.dino_weights).cuda()
            fm = precompute_dino_features(extractor, tr_images[:1]).float()
            del extractor
        else:
            raise SystemExit("this config conditions on DINO features: pass --dino-weights <local Dinov2Model checkpoint> (or --dino-random-init), "
                             "or --dino-map <tensor (1,Hp,Wp,C) saved with torch.save>")
        dino = dict(features=fm[:1], pose=tr_poses[0], focal=focal, H=H, W=W)
        dino_dim = int(fm.shape[-1])
    model = model_from_config(cfg, dino_dim=dino_dim, mma_mode=args.mode)
    if args.checkpoint:
        load_checkpoint_into(model, torch.load(args.checkpoint, map_location="cpu", weights_only=True))
    model = model.cuda().eval()
    targets = images.permute(0, 2, 3, 1).contiguous()
    res = evaluate_views(model, poses, H, W, focal, rs["near"], rs["far"], rs["n_samples"], targets=targets, out_dir=args.out,
                         white_bkgd=rs["white_bkgd"], mma_mode=args.mode, ert_eps=args.ert, dino=dino)
    metrics = {"psnr": res["psnr"], "ssim": res["ssim"], "views": len(res["per_view"]), "per_view": res["per_view"],
               "H": H, "W": W, "n_samples": rs["n_samples"], "mode": args.mode}
    if args.out:
        os.makedirs(args.out, exist_ok=True)
        with open(os.path.join(args.out, "metrics.json"), "w") as f:
            json.dump(metrics, f, indent=1)
    print(json.dumps({k: metrics[k] for k in ("psnr", "ssim", "views")}))
    return metrics


if __name__ == "__main__":
    main()
