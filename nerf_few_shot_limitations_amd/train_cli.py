"""python -m nerf_few_shot_limitations_amd.train_cli --config experiments/baseline.yaml --data data/nerf_synthetic/lego \\
        [--epochs N] [--mode bf16|f16|f32] [--eval-mode f16] [--out DIR] [--checkpoint CKPT] [--dino-weights DIR | --dino-maps maps.pt] [--seed 0]

The training run of `NeRFDINOTrainer` (src/training/train.py:244-292 `train_step`, :344-372 `train`) as a command on the
HIP path: the YAML loads unchanged; per epoch and training view the rays are cast at the progressive schedule's
resolution (focal scaled, target resized bilinearly: train.py:262-266), shuffled (:268) and consumed in ray batches
(:270-288), each of them one `training.FusedStep` -- stratified samples -> NeRFMLP -> VolumeRenderer ->
rgb_weight * mse -> backward -> Adam(lr, weight_decay) -- with MultiStepLR between epochs (:119-123), validation every
`output.val_freq` epochs on the fused renderer (`evaluate_views`) and checkpoints under the reference's key names
(:374-389, readable by evaluate.py:22-33 and by evaluate_cli).

Objective.  The step minimises `loss.rgb_weight * mse(rgb, target)` -- exactly what the reference's training loss returns:
train.py:27-44 (the `NeRFLoss` class train.py defines and uses) computes only that term; `loss.depth_weight` /
`loss.reg_weight` of the YAMLs are read into the constructor and never used there.  (nerf_mlp.NeRFLoss, a different class the
trainer does not import, adds mean(weights^2); tests/test_gpu_training.py covers it through the autograd route.)

use_dino configs condition on the DINOv2 feature map of every training view.  As in train.py:158-169 the maps are computed ONCE,
under no_grad, by the extractor the config names (config.dino_model_from_config: SpatialDINOFeatures or
MultiScaleDINOFeatures with LoRA wrappers) -- from --dino-weights (a local transformers Dinov2Model checkpoint; the weights are
not available offline) or, to exercise the pipeline without them, --dino-random-init; --dino-maps takes precomputed maps
(V,Hp,Wp,C) instead.  The features of a sample are fetched by projection into the view being trained on (:203-214).  Because the
maps are constants, no gradient reaches the extractor -- in the reference too: its LoRA matrices sit in the optimizer
(train.py:105-110) but never receive one.

--checkpoint resumes a run: weights, Adam moments and step count, epoch counter and best PSNR (the reference's train.py saves
these keys, :374-389, but has no resume path); the LR schedule is a function of the epoch.  wandb and LPIPS are not part of
this command.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import time

import torch
import torch.nn.functional as F

from . import _lib as L
from . import (dino_model_from_config, evaluate_views, get_rays, load_blender_data, load_checkpoint_into, load_config, model_from_config,
               precompute_dino_features, render_settings, sample_points_along_rays)
from .renderer import make_dino
from .training import FusedStep


def schedule_for(cfg, epoch):
    """(H_train, W_train, N_samples, batch_size) of an epoch: train.py:249-259."""
    t = cfg["training"]
    s = t["progressive_schedule"]
    if epoch < 50:
        return (*s["epochs_0_50"], t["batch_size"] * 2)
    if epoch < 100:
        return (*s["epochs_50_100"], t["batch_size"])
    return (*s["epochs_100_plus"], t["batch_size"] // 2)


def lr_at(cfg, epoch):
    """MultiStepLR(milestones, gamma) evaluated for the epoch about to run (train.py:119-123, scheduler.step() after each epoch)."""
    o = cfg["optimizer"]
    return float(o["lr"]) * float(o["lr_gamma"]) ** sum(1 for m in o["lr_milestones"] if epoch >= m)


def view_rays(image, pose, H, W, focal, Ht, Wt):
    """Rays and target of one training view at the schedule's resolution (train.py:174-186,262-266)."""
    tgt = image
    if tgt.shape[-1] == 4:
        tgt = tgt[..., :3] * tgt[..., 3:4] + (1.0 - tgt[..., 3:4])
    if (Ht, Wt) == (H, W):
        ro, rd = get_rays(H, W, focal, pose)
    else:
        ro, rd = get_rays(Ht, Wt, focal * (Ht / H), pose)
        tgt = F.interpolate(tgt.permute(2, 0, 1).unsqueeze(0), size=(Ht, Wt), mode="bilinear", align_corners=False).squeeze(0).permute(1, 2, 0)
    return ro.reshape(-1, 3), rd.reshape(-1, 3), tgt.reshape(-1, 3).contiguous()


def fetch_features(dino_struct, pts):
    """(n,3) points -> (n,C) features of the source view (project_points_to_image + sample_features_at_points)."""
    d, fm = dino_struct
    n = pts.shape[0]
    feats = torch.empty((n, int(fm.shape[-1])), dtype=torch.float32, device=pts.device)
    L.check(L.lib().nrf_project_fetch(C.byref(d), L.ptr(pts), n, L.ptr(feats), None, L.stream_ptr()))
    return feats


def train_epoch(step, cfg, epoch, images, poses, H, W, focal, near, far, gen, dino_maps=None, max_batches=None, rank=0, world=1):
    """One pass of train.py:261-290 over the training views; returns (mean loss, ray-samples processed).
    world > 1 (data parallel, `FusedStep(data_parallel=True)`): every rank draws the SAME shuffle (same generator seed) and
    takes rays rank, rank+world, ... of each batch -- the batches are those of one process (the stratified jitter of a ray is
    keyed by its position inside the call, so the sample depths differ from a single-process run's)."""
    Ht, Wt, S, batch = schedule_for(cfg, epoch)
    model = step.model
    use_dino = model.net == L.NRF_NET_V3
    total, n_batches, samples = None, 0, 0
    for v in range(len(images)):
        ro, rd, tgt = view_rays(images[v], poses[v], H, W, focal, Ht, Wt)
        dino = make_dino(dino_maps[v:v + 1], poses[v], focal, H, W) if use_dino else None       # train.py:204-206: full-resolution intrinsics
        order = torch.randperm(ro.shape[0], device=ro.device, generator=gen)
        for i in range(0, order.shape[0], batch):
            idx = order[i:i + batch]
            if world > 1:
                idx = idx[: idx.shape[0] // world * world][rank::world]        # equal shards: the all-reduce averages per-rank means
                if idx.shape[0] == 0:
                    continue
            o, d, t = ro[idx], rd[idx], tgt[idx]
            pts, z = sample_points_along_rays(o, d, near, far, S, perturb=True, seed=epoch * 1_000_003 + v * 10_007 + i)
            n = idx.shape[0]
            dirs = d[:, None, :].expand(n, S, 3).reshape(-1, 3)                         # train.py:225: raw ray directions per sample
            feats = fetch_features(dino, pts.reshape(-1, 3)) if use_dino else None
            loss = step(pts.reshape(-1, 3), z, d, t, dirs=dirs, dino=feats)
            total = loss if total is None else total + loss
            n_batches += 1
            samples += n * S
            if max_batches is not None and n_batches >= max_batches:
                return float(total) / n_batches, samples
    return (float(total) / max(n_batches, 1)) if total is not None else 0.0, samples


def save_checkpoint(path, model, step, epoch, best_psnr, cfg):
    """train.py:374-389's dictionary: `nerf_model_state_dict` is what evaluate.py:27 / load_checkpoint_into read."""
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    opt = step.opt
    o = cfg["optimizer"]
    torch.save({"epoch": epoch, "best_psnr": best_psnr,
                "nerf_model_state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                # flat-vector Adam state (training.Adam; layout = include/nerfhip.h's flat parameter order), restored by --checkpoint
                "optimizer_state_dict": {"step": opt.step_count, "exp_avg": None if opt.exp_avg is None else opt.exp_avg.cpu(),
                                         "exp_avg_sq": None if opt.exp_avg_sq is None else opt.exp_avg_sq.cpu(), "lr": opt.lr},
                # MultiStepLR is a pure function of the epoch (lr_at): its state is the epoch counter
                "scheduler_state_dict": {"last_epoch": epoch + 1, "milestones": list(o["lr_milestones"]), "gamma": float(o["lr_gamma"])},
                "config": cfg}, path)


def resume_from(ckpt, model, step):
    """Restore what save_checkpoint wrote beyond the weights: Adam moments + step count; returns (first epoch to run, best PSNR).
    A checkpoint without them (e.g. one written by the reference) resumes the weights only, from epoch 0."""
    os_ = ckpt.get("optimizer_state_dict") if isinstance(ckpt, dict) else None
    start, best = 0, 0.0
    if isinstance(os_, dict) and os_.get("exp_avg") is not None and "step" in os_:
        fp, flat = step.opt._buffers()
        if os_["exp_avg"].numel() == flat.numel():
            step.opt.exp_avg.copy_(os_["exp_avg"].to(flat.device))
            step.opt.exp_avg_sq.copy_(os_["exp_avg_sq"].to(flat.device))
            step.opt.step_count = int(os_["step"])
            start, best = int(ckpt.get("epoch", -1)) + 1, float(ckpt.get("best_psnr", 0.0))
    return start, best


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", required=True)
    ap.add_argument("--data", required=True, help="dataset directory holding transforms_train.json / transforms_test.json")
    ap.add_argument("--epochs", type=int, default=None, help="default: training.epochs of the config")
    ap.add_argument("--mode", default="bf16", choices=["bf16", "f16", "f32"],
                    help="arithmetic of the TRAINING kernels (bf16: its exponent range suits the unscaled gradients)")
    ap.add_argument("--eval-mode", default="f16", choices=["bf16", "f16", "f16x3", "f32"],
                    help="arithmetic of the validation renders: f16 keeps the reported PSNR within 0.01 dB of an fp32 render (bf16 does not)")
    ap.add_argument("--out", default=None, help="default: output.save_dir of the config")
    ap.add_argument("--checkpoint", default=None, help="resume from this file: weights, Adam moments / step, epoch and best PSNR")
    ap.add_argument("--dino-maps", default=None, help="precomputed feature maps (V,Hp,Wp,C), torch.save'd, one per training view")
    ap.add_argument("--dino-weights", default=None, help="local transformers Dinov2Model checkpoint (dir or file) for the extractor of the config")
    ap.add_argument("--dino-random-init", action="store_true", help="build the extractor with random weights (pipeline runs, features meaningless)")
    ap.add_argument("--max-test-views", type=int, default=None)
    ap.add_argument("--max-batches", type=int, default=None, help="stop every epoch after this many ray batches (smoke runs)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--data-parallel", action="store_true",
                    help="launched with torch.distributed.run, one process per GPU: ray batches shard over the ranks, one all-reduce of the "
                         "flat gradient vector per step (RCCL); rank 0 validates and writes")
    ap.add_argument("--rehearse", action="store_true",
                    help="--data-parallel on a box with ONE GPU: every rank on cuda:0, the gradient all-reduce over gloo through host memory "
                         "(RCCL refuses two ranks on one card); exercises the sharding and the collective, not multi-GPU speed")
    args = ap.parse_args(argv)
    rank, world = 0, 1
    if args.data_parallel:
        import torch.distributed as dist
        local = 0 if args.rehearse else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        if not dist.is_initialized():
            if args.rehearse:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        rank, world = dist.get_rank(), dist.get_world_size()

    torch.manual_seed(args.seed)                          # parameter init (when no checkpoint is given) and the ray shuffles
    cfg = load_config(args.config)
    rs = render_settings(cfg)
    out_dir = args.out or cfg["output"]["save_dir"]
    epochs = args.epochs if args.epochs is not None else int(cfg["training"]["epochs"])
    name = cfg.get("experiment", {}).get("name", "run")
    res = cfg["data"].get("resolution")
    images, poses, (H, W, focal) = load_blender_data(args.data, "train", img_size=res)
    nv = cfg["data"].get("num_views")
    if nv:
        images, poses = images[:nv], poses[:nv]                                          # train.py:141-143
    dev = torch.device("cuda", torch.cuda.current_device())
    images = [im.permute(1, 2, 0).float().to(dev) for im in images]
    poses = [p.float() for p in poses]
    test_images, test_poses, _ = load_blender_data(args.data, "test", img_size=res)
    if args.max_test_views:
        test_images, test_poses = test_images[: args.max_test_views], test_poses[: args.max_test_views]

    use_dino = bool(cfg.get("model", {}).get("use_dino", True))
    dino_maps, dino_dim = None, 64
    if use_dino:
        if args.dino_maps:
            dino_maps = torch.load(args.dino_maps, map_location="cpu", weights_only=True).float().to(dev)
        elif args.dino_weights or args.dino_random_init:
            extractor = dino_model_from_config(cfg, weights=args.dino_weights).to(dev)           # train.py:57-75
            dino_maps = precompute_dino_features(extractor, torch.stack(images)[..., :3]).float()   # train.py:158-169: once, under no_grad
            del extractor
        else:
            raise SystemExit("this config conditions on DINO features: pass --dino-weights <local Dinov2Model checkpoint> (or --dino-random-init), "
                             "or --dino-maps <tensor (V,Hp,Wp,C) saved with torch.save>, one map per training view")
        if dino_maps.dim() != 4 or dino_maps.shape[0] < len(images):
            raise SystemExit("--dino-maps must hold one (Hp,Wp,C) map per training view")
        dino_dim = int(dino_maps.shape[-1])
    model = model_from_config(cfg, dino_dim=dino_dim, mma_mode=args.mode)
    ckpt = torch.load(args.checkpoint, map_location="cpu", weights_only=True) if args.checkpoint else None
    if ckpt is not None:
        load_checkpoint_into(model, ckpt)
    model = model.to(dev).train()
    o, lw = cfg["optimizer"], cfg.get("loss", {})
    step = FusedStep(model, lr=float(o["lr"]), weight_decay=float(o["weight_decay"]), rgb_weight=float(lw.get("rgb_weight", 1.0)),
                     white_bkgd=rs["white_bkgd"], data_parallel=world > 1)
    gen = torch.Generator(device=dev)
    gen.manual_seed(args.seed)
    best, log, first_epoch = 0.0, [], 0
    if ckpt is not None:
        first_epoch, best = resume_from(ckpt, model, step)
    targets = test_images.permute(0, 2, 3, 1).contiguous()
    eval_dino = dict(features=dino_maps[0:1], pose=poses[0], focal=focal, H=H, W=W) if use_dino else None      # train.py:203-208
    for epoch in range(first_epoch, epochs):
        step.opt.lr = lr_at(cfg, epoch)
        t0 = time.perf_counter()
        loss, samples = train_epoch(step, cfg, epoch, images, poses, H, W, focal, rs["near"], rs["far"], gen, dino_maps, args.max_batches, rank, world)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rec = {"epoch": epoch + 1, "loss": loss, "lr": step.opt.lr, "seconds": round(dt, 3), "Msamples_per_s": round(world * samples / dt / 1e6, 2)}
        if rank != 0:                                                         # every rank holds the same parameters: rank 0 validates and writes
            log.append(rec)
            continue
        if (epoch + 1) % int(cfg["output"]["val_freq"]) == 0 or epoch + 1 == epochs:
            m = evaluate_views(model, test_poses, H, W, focal, rs["near"], rs["far"], rs["n_samples"], targets=targets, white_bkgd=rs["white_bkgd"],
                               mma_mode=args.eval_mode, dino=eval_dino, out_dir=os.path.join(out_dir, f"val_{epoch + 1}"))
            model.train()
            rec.update(psnr=m["psnr"], ssim=m["ssim"])
            if m["psnr"] > best:
                best = m["psnr"]
                save_checkpoint(os.path.join(out_dir, f"best_{name}.pth"), model, step, epoch, best, cfg)
        if (epoch + 1) % int(cfg["output"]["save_freq"]) == 0:
            save_checkpoint(os.path.join(out_dir, f"epoch_{epoch + 1}.pth"), model, step, epoch, best, cfg)
        log.append(rec)
        print(json.dumps(rec), flush=True)
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "train_log.json"), "w") as f:
            json.dump(log, f, indent=1)
    if args.data_parallel:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return log


if __name__ == "__main__":
    main()
