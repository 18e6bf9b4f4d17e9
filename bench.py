#!/usr/bin/env python3
"""bench.py -- M ray-samples/s of the fused sample+encode+MLP+composite renderer.

    python bench.py --gpus N --steps K --warmup W          (N>1: one rank per GPU over RCCL -- under torch.distributed.run, or
                                                            self-launched: with no launcher in the environment this command starts
                                                            its own N ranks as fresh child processes)

Workload (BASELINE.json metric): synthetic 800x800 camera frames, 64 samples per ray, the 8x256
NeRF MLP (nerf_model.NeRFMLP, 951 808 FLOP per ray-sample), deterministic random-init weights
("solid" scene; early ray termination OFF, so every one of the R*S samples is evaluated), f16 MFMA (HEADLINE_MODE: the
fastest mode whose PSNR stays within 0.01 dB of the fp32 render -- asserted by tests/test_gpu_trained_scene.py; --mode bf16 | f16x3 | f32).

--scaling weak (default): one step renders n_gpus views of the sensor: every view's rays are cut into 16-row pixel tiles
dealt round-robin over the ranks, so per-GPU work is one frame's worth of rays whatever N is; each rank renders its
tiles of ALL views with ONE kernel launch straight into the gather buffer and the tiles are exchanged with ONE RCCL
all_gather per step so that every rank holds all frames; the all_gather of step i runs on RCCL's stream while step i+1 renders
(two gather buffers; --no-overlap serialises them).  --scaling strong: ONE view per step cut over the ranks
(BASELINE.json config 5's shape with --samples 128).  Whichever is the headline, the other one is measured too (a few
steps after the timed region) and reported as `strong_scaling` / `weak_scaling`; the all_gather is timed separately
(`gather_ms`).  Inputs are generated in-kernel (camera mode): nothing is read from the host in the timed region.

The JSON line also carries
  roofline     -- algorithmic FLOPs of the render kernel / its mean launch duration (HIP events on the launch stream)
                  against the dense bf16 MFMA peak (2.5 PFLOP/s);
  cpu_baseline -- the CPU oracle (oracle/nerf_oracle.py, a port of the reference's PyTorch CPU path) timed on a band of
                  rows of the same frame on this host's cores;
  parity       -- max abs error / PSNR of every arithmetic mode vs that oracle band, and the PSNR delta against a common
                  ground truth (the band marched with 2x the samples); parity.trained_scene: a field trained in this run on a
                  generated Blender-format scene (tools/trained_scene.py), every mode's PSNR against the ground-truth images;
                  parity.headline_mode: whether the mode of `value` meets the 0.01 dB bar on both;
  parity_mode  -- the parity-grade fast mode (f16x3: split-f16, 3 MFMAs per product): ms per frame, TFLOP/s credited 1x,
                  fraction of the 2.5 PFLOP/s peak -- the mode that meets the 1e-4 / 0.01 dB bars -- beside f16 and f32;
  ert          -- early ray termination on the "smooth" scene: ms, speed-up, max |d rgb| vs the full march (must be <= eps).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The arithmetic mode of the headline number: the fastest mode whose PSNR stays within 0.01 dB of the fp32 render on a TRAINED
# field and on the solid-scene band (BASELINE.json's PSNR bar; tools/trained_scene.py, tests/test_gpu_trained_scene.py assert it).
# bf16 (same MFMA rate, 8 mantissa bits) misses that bar by 10-30x and stays available as --mode bf16.
HEADLINE_MODE = "f16"

PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3, "f16x3": 2500.0}      # /opt/skills/guides/MI355X_MICROARCH.md, dense


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes (torch.distributed.run, one
    per GPU) before this process has touched a GPU or imported torch, pass rank 0's JSON line through (the children inherit
    stdout) and exit with their status.  Nothing is ever exec'd from a process that initialised HIP."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    print(f"[bench] launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def training_line(N, args, dev):
    """SURVEY.md section 8 row f1, reported next to the render metric (not part of `value`): one optimisation step of the
    reference's loop (train.py:280-288 / train_minimal.py:102-123: render a ray batch, mse on rgb, backward, Adam) at the
    reference's own batch (baseline.yaml:32-34: 2048 rays x 32 samples), random-init weights, through training.FusedStep."""
    import torch
    from nerf_few_shot_limitations_amd.training import FusedStep
    R, S = 2048, 32
    mode = args.mode if args.mode != "f16x3" else "f32"
    torch.manual_seed(0)
    if args.net == "v3":
        m = N.NeRFMLP(pos_freq=12, dir_freq=4, hidden_dim=256, num_density_layers=8, use_dino=True, dino_dim=64, mma_mode=mode).to(dev).train()
        pts = torch.rand(R * S, 3, device=dev) * 4 - 2
    elif args.net == "v2":
        m = N.NeRFMLP(pos_freq=10, dir_freq=4, hidden_dim=256, num_density_layers=8, use_dino=False, mma_mode=mode).to(dev).train()
        pts = torch.rand(R * S, 3, device=dev) * 4 - 2
    else:
        m = N.NeRFMLP(pos_dim=63, hidden_dim=256, n_layers=8, mma_mode=mode).to(dev).train()
        pts = torch.rand(R * S, 63, device=dev) * 2 - 1
    dirs = torch.rand(R * S, 3, device=dev) * 2 - 1
    z = torch.sort(torch.rand(R, S, device=dev) * 4 + 2, dim=-1).values.contiguous()
    d = torch.rand(R, 3, device=dev) - 0.5
    tgt = torch.rand(R, 3, device=dev)
    step = FusedStep(m, lr=5e-4, weight_decay=1e-6)
    kw = dict(dirs=dirs) if args.net != "v1" else {}
    if args.net == "v3":
        kw["dino"] = torch.rand(R * S, 64, device=dev) * 2 - 1
    first = None
    for _ in range(3):
        loss = step(pts, z, d, tgt, **kw)
        first = loss.item() if first is None else first
    torch.cuda.synchronize()
    k = 200        # long enough to leave the first dozens of steps after a pause behind (they run 10-15 % faster: the chip's power state)
    t0 = time.perf_counter()
    for _ in range(k):
        loss = step(pts, z, d, tgt, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    out = {"metric": "M ray-samples/s per optimisation step (forward + backward + Adam)", "value": round(R * S / dt / 1e6, 2),
           "ms_per_step": round(dt * 1e3, 4), "steps_timed": k, "rays": R, "samples_per_ray": S, "net": args.net, "dtype": mode,
           "loss_first": round(first, 6), "loss_last": round(loss.item(), 6)}
    # the reference's UNMODIFIED train_step body (train.py:280-287) on the drop-in surface: render_rays with a grad_fn, torch's own Adam
    rend = N.NeRFRenderer(m, 2.0, 6.0, dino_features=[torch.rand(1, 9, 9, 64, device=dev) * 2 - 1] if args.net == "v3" else None,
                          poses=[torch.eye(4)], focal=100.0, H=128, W=128)
    ro = torch.rand(R, 3, device=dev) - 0.5 + torch.tensor([0.0, 0.0, 4.0], device=dev)
    opt = torch.optim.Adam(m.parameters(), lr=5e-4, weight_decay=1e-6)

    def body():
        pred = rend.render_rays(ro, d, 0, S)
        ls = torch.nn.functional.mse_loss(pred["rgb"], tgt)
        opt.zero_grad()
        ls.backward()
        opt.step()
    for _ in range(3):
        body()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        body()
    torch.cuda.synchronize()
    out["autograd_render_rays_ms_per_step"] = round((time.perf_counter() - t0) / k * 1e3, 4)
    # the same body with torch's fused Adam (one keyword): the default foreach step() is 0.27 ms of host time for 24 parameter tensors
    opt = torch.optim.Adam(m.parameters(), lr=5e-4, weight_decay=1e-6, fused=True)
    for _ in range(3):
        body()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        body()
    torch.cuda.synchronize()
    out["autograd_render_rays_fused_adam_ms_per_step"] = round((time.perf_counter() - t0) / k * 1e3, 4)
    out["autograd_note"] = "train.py:280-287 verbatim in shape: predictions = render_rays(...), mse, zero_grad, backward, torch.optim.Adam.step"
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--mode", default=HEADLINE_MODE, choices=["bf16", "f16", "f32", "f16x3"])
    ap.add_argument("--net", default="v1", choices=["v1", "v2", "v3"])
    ap.add_argument("--scene", default="solid", choices=["fog", "solid", "smooth"])
    ap.add_argument("--ert", type=float, default=0.0)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: n_gpus views per step (one frame's worth of rays per GPU); strong: ONE view per step cut over the ranks")
    ap.add_argument("--other-steps", type=int, default=10, help="steps of the other scaling mode measured after the timed region (0 = skip)")
    ap.add_argument("--tile-rows", type=int, default=0, help="rows per pixel tile; 0 = largest <= 16 that deals the tiles evenly")
    ap.add_argument("--no-train", action="store_true", help="skip the optimisation-step timing appended as 'training' (N=1 only)")
    ap.add_argument("--no-extras", action="store_true", help="skip the parity_mode / ert legs (N=1 only)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: do not overlap the all_gather of a step with the next step's render")
    ap.add_argument("--cpu-rows", type=int, default=16, help="rows of the frame the CPU baseline renders (0 = skip)")
    ap.add_argument("--no-trained-scene", action="store_true", help="skip parity.trained_scene (train a field on a generated scene, PSNR delta of every mode; N=1 only)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on a box with ONE GPU: every rank on cuda:0, tiles exchanged over gloo through host memory (RCCL refuses two ranks on "
                         "one card).  Exercises the launcher, the tile dealing and the gather; its numbers are not multi-GPU numbers")
    args = ap.parse_args(argv)

    # one rank per GPU, decided from the launcher's environment BEFORE anything touches a device (torch is not even imported yet)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or with no launcher at all")
    import torch
    dev_index = 0 if args.rehearse else local_rank
    backend = "gloo" if args.rehearse else "nccl"
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            # rehearsal only: gloo's transport prints its connection chatter on stdout (C level), where exactly ONE JSON line is
            # expected -- park fd 1 on stderr while the group comes up
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group(backend=backend)
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)

    import nerf_few_shot_limitations_amd as N
    from nerf_few_shot_limitations_amd import tiles
    from oracle import nerf_oracle as O               # cpu_baseline / parity legs only

    H, W, S = args.height, args.width, args.samples
    c2w = torch.from_numpy(O.LEGO_LIKE_C2W.copy())
    focal = O.focal_for(W)
    seed = {"v1": 0, "v2": 1, "v3": 2}[args.net]

    def make_model(scene, mode):
        p = O.make_weights(args.net, seed, scene)
        if args.net == "v1":
            m = N.NeRFMLP(pos_dim=63, hidden_dim=256, n_layers=8, mma_mode=mode)
            m.load_state_dict(p)
        elif args.net == "v2":
            m = N.NeRFMLP(pos_freq=10, dir_freq=4, hidden_dim=256, num_density_layers=8, use_dino=False, mma_mode=mode)
            m.load_state_dict(p, strict=False)
        else:
            m = N.NeRFMLP(pos_freq=12, dir_freq=4, hidden_dim=256, num_density_layers=8, use_dino=True, dino_dim=64, mma_mode=mode)
            m.load_state_dict(p, strict=False)
        return m.to(dev).eval(), p

    model, p = make_model(args.scene, args.mode)
    dino = None
    if args.net == "v3":
        fm = torch.from_numpy(O.uniform01(7, 28 * 28 * 64).reshape(1, 28, 28, 64) * 2 - 1)
        dino = dict(features=fm, pose=c2w, focal=focal, H=H, W=W)
    flops_per_sample = model.flops_per_sample()

    def orbit(n):
        poses = []
        for v in range(n):
            th = 2 * math.pi * v / 8
            rz = torch.tensor([[math.cos(th), -math.sin(th), 0, 0], [math.sin(th), math.cos(th), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=torch.float32)
            poses.append(rz @ c2w)
        return torch.stack(poses)

    tile_rows = args.tile_rows
    if tile_rows <= 0:
        even = [r for r in range(16, 0, -1) if H % r == 0 and (H // r) % world == 0]
        tile_rows = even[0] if even else 16
    tile_rays = tile_rows * W

    def make_job(n_views):
        return tiles.TileJob(model, H, W, focal, orbit(n_views), 2.0, 6.0, S, rank, world, tile_rays, ert_eps=args.ert, device=dev, dino=dino)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(n_views, steps, warmup, overlap):
        """`warmup` untimed + `steps` timed steps: (wall seconds [max over ranks], mean kernel ms, mean gather ms or None, rays/launch).
        overlap (N > 1): two jobs with their own gather buffers take turns; the all_gather of step i is enqueued asynchronously (RCCL
        runs it on its own stream, after the render it depends on) while step i+1 renders -- every step still ends with every rank
        holding every frame, the exchange just no longer sits between two renders.  A buffer is rendered into again only after the
        gather that read it has completed (work.wait() orders the render stream behind it)."""
        jobs = [make_job(n_views) for _ in range(2 if (overlap and world > 1) else 1)]
        frames = [None] * len(jobs)
        async_gather = world > 1 and backend == "nccl" and len(jobs) == 2
        ex = tiles.OverlappedGather([job.buf for job in jobs]) if async_gather else None      # the double-buffered exchange (tiles.py)
        kev, gev = [], []

        def step(i, timed):
            k = i % len(jobs)
            job = jobs[k]
            if timed:
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            if async_gather:
                def render(slot):
                    if timed:
                        e0.record()
                    jobs[slot].launch()                    # ONE render kernel launch: this rank's tiles of all views, written
                    if timed:                              # straight into the gather buffer
                        e1.record()
                ex.step(render)                            # waits for the gather that last read this buffer, renders, enqueues the all_gather
            else:
                if timed:
                    e0.record()
                job.launch()
                if timed:
                    e1.record()
                if world > 1:
                    if backend != "nccl":
                        frames[k] = tiles.gather_frames(job.buf.cpu(), H * W, tile_rays)
                    else:
                        frames[k] = tiles.gather_frames(job.buf, H * W, tile_rays)          # ONE all_gather (RCCL) + the view into frame order
            if timed:
                e2.record()
                kev.append((e0, e1)); gev.append((e1, e2))

        def drain():
            if ex is not None:
                ex.drain()

        for i in range(warmup):
            step(i, False)
        drain()
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i, True)
        drain()
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        kms = sum(a.elapsed_time(b) for a, b in kev) / max(len(kev), 1)
        gms = sum(a.elapsed_time(b) for a, b in gev) / max(len(gev), 1) if (world > 1 and len(jobs) == 1) else None
        return dt, kms, gms, jobs[0].rays_per_step, jobs[0].launches_per_step

    views_main = world if args.scaling == "weak" else 1
    overlap = world > 1 and backend == "nccl" and not args.no_overlap
    # no automatic fall-back to the serial exchange: a rank that switched protocol on its own would leave the others inside
    # mismatched collectives (a hang instead of an error).  An RCCL failure surfaces; --no-overlap selects the serial form.
    dt, kernel_ms, gather_ms, rays_per_step_rank, launches_per_step = run(views_main, args.steps, args.warmup, overlap)
    print(f"[bench] rank {rank}: timed region done", file=sys.stderr, flush=True)
    if world > 1 and gather_ms is None and args.other_steps > 0:
        # the exchange on its own: a few steps with the all_gather NOT overlapped, timed with events around it
        _, _, gather_ms, _, _ = run(views_main, min(args.other_steps, 5), 1, False)

    samples_per_launch = rays_per_step_rank * S                   # what this rank's launch(es) of one step march: kernel_ms spans exactly them
    samples_per_step = views_main * H * W * S                     # all ranks together
    value = samples_per_step * args.steps / dt / 1e6
    achieved = samples_per_launch * flops_per_sample / (kernel_ms * 1e-3) / 1e12

    # HBM bytes per launch come from PMC counters, which bench.py cannot collect itself: a STATIC figure read from the
    # committed rocprofv3 summary of this very workload (profiles/), else null
    traffic, traffic_src = None, None
    if (H, W, S, world, args.scaling, args.scene, args.ert) == (800, 800, 64, 1, "weak", "solid", 0.0):
        name = f"r03_pmc_{args.net}_{args.mode}_summary.json"          # tools/pmc_target.sh of this very workload (one 800x800x64 launch)
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                traffic = json.load(f)["kernels"][0]["derived"]["hbm_bytes_per_launch"]
            traffic_src = f"static (profiled offline): profiles/{name} (FETCH_SIZE*2 + WRITE_SIZE, bytes per launch)"
        except (OSError, KeyError, ValueError, IndexError):
            pass

    out = {
        "metric": "M ray-samples/sec (sample+MLP+composite) at 800^2x64",
        "value": round(value, 2), "unit": "M ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": args.mode, "data": "synthetic",
        "config": {"workload": f"{H}x{W} camera frame x {S} samples/ray, NeRFMLP {args.net} 8x256, scene {args.scene}, "
                               f"{views_main} view(s)/step, {tile_rows}-row pixel tiles dealt round-robin over {world} GPU(s), one launch + one all_gather per step",
                   "rays_per_gpu_per_step": rays_per_step_rank, "samples_per_ray": S, "ert_eps": args.ert,
                   "flops_per_sample": flops_per_sample, "parallelism": f"pixel-tile x{world}",
                   "gather": ("all_gather of step i overlapped with the render of step i+1 (two gather buffers); no fallback fired (there is none: a failure raises)"
                              if overlap else (("one all_gather between renders" + (" over gloo through host memory (rehearsal)" if args.rehearse else ""))
                                               if world > 1 else "none (one GPU)"))},
        "gather_ms": None if gather_ms is None else round(gather_ms, 4),
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_TFLOPS[args.mode], "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_TFLOPS[args.mode], 4), "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "render_kernel", "kernel_ms": round(kernel_ms, 4), "launches_timed": args.steps * launches_per_step,
                     "launches_per_step": launches_per_step},
        "distributed": {"world_size": world if dist is None else dist.get_world_size(), "backend": "none (one process)" if dist is None else dist.get_backend(),
                        "rehearsal_on_one_gpu": bool(args.rehearse), "launcher": "torch.distributed.run (self-launched when no WORLD_SIZE is set)"},
    }

    # the other scaling mode, a few steps (every rank takes part)
    if args.other_steps > 0 and world > 1:
        other = "strong" if args.scaling == "weak" else "weak"
        views_o = 1 if other == "strong" else world
        dto, kmo, gmo, rays_o, _ = run(views_o, args.other_steps, 2, overlap)
        if gmo is None:
            _, _, gmo, _, _ = run(views_o, min(args.other_steps, 5), 1, False)
        out[f"{other}_scaling"] = {
            "workload": f"{views_o} view(s) of {H}x{W}x{S} per step cut over {world} GPU(s)", "steps": args.other_steps,
            "value": round(views_o * H * W * S * args.other_steps / dto / 1e6, 2), "unit": "M ray-samples/s",
            "ms_per_step": round(dto / args.other_steps * 1e3, 4), "kernel_ms": round(kmo, 4), "gather_ms": None if gmo is None else round(gmo, 4),
            "rays_per_gpu_per_step": rays_o}

    single = rank == 0 and world == 1
    if single and args.cpu_rows > 0:
        # CPU oracle on a band of rows of the same frame (port of the reference's torch-CPU path)
        # the box's CPU share, not the host's core count (a cgroup-limited box reports far more cores than it may use)
        threads = min(len(os.sched_getaffinity(0)), int(os.environ.get("NERF_BENCH_CPU_THREADS", "16")))
        print(f"[bench] GPU part done: {value:.1f} M samples/s; timing the CPU oracle on {threads} threads ...", file=sys.stderr, flush=True)
        torch.set_num_threads(threads)
        rows = min(args.cpu_rows, H)
        r0 = (H // 2) * W
        r1 = r0 + rows * W
        ro, rd = O.get_rays(H, W, focal, c2w)
        ro, rd = ro.reshape(-1, 3)[r0:r1].contiguous(), rd.reshape(-1, 3)[r0:r1].contiguous()
        best = None
        ref = None
        for _ in range(3):
            tc = time.perf_counter()
            ref = O.render_rays(p, args.net, ro, rd, 2.0, 6.0, S, chunk=2048, dino=dino)
            el = time.perf_counter() - tc
            best = el if best is None else min(best, el)
            print(f"[bench] cpu oracle pass: {el:.2f} s", file=sys.stderr, flush=True)
        out["cpu_baseline"] = {"value": round(rows * W * S / best / 1e6, 4), "unit": "M ray-samples/s", "cores": threads,
                               "kind": "port", "sample": f"{rows} rows ({rows * W} rays x {S} samples) of the same {H}x{W} frame, "
                                                          f"torch fp32 CPU, chunk 2048 rays, best of 3"}
        # PSNR delta against a common ground truth (the lego images are not available offline: SURVEY.md section 8d):
        # GT = the same rows marched with twice the samples by the CPU oracle; every mode at S samples vs that GT
        gt = O.render_rays(p, args.net, ro, rd, 2.0, 6.0, 2 * S, chunk=2048, dino=dino)["rgb"]
        ps_ref = O.psnr(ref["rgb"], gt)
        par = {"band_rows": rows, "psnr_oracle_vs_gt_db": round(ps_ref, 3)}
        for mode in ("bf16", "f16", "f16x3", "f32"):
            rgb_m, depth_m = N.render_camera(model, H, W, focal, c2w, 2.0, 6.0, S, ray_begin=r0, ray_end=r1, mma_mode=mode, device=dev, dino=dino)
            torch.cuda.synchronize()
            par[mode] = {"max_abs_rgb": float((rgb_m.cpu() - ref["rgb"]).abs().max()), "max_abs_depth": float((depth_m.cpu() - ref["depth"]).abs().max()),
                         "psnr_vs_oracle_db": round(O.psnr(rgb_m.cpu(), ref["rgb"]), 2),
                         "psnr_delta_db": round(abs(O.psnr(rgb_m.cpu(), gt) - ps_ref), 6),
                         "meets_1e-4": bool(float((rgb_m.cpu() - ref["rgb"]).abs().max()) <= 1e-4 and float((depth_m.cpu() - ref["depth"]).abs().max()) <= 1e-4),
                         "meets_0.01dB": bool(abs(O.psnr(rgb_m.cpu(), gt) - ps_ref) <= 0.01)}
        out["parity"] = par
        if not args.no_trained_scene:
            # PSNR delta of every mode on a TRAINED field (tools/trained_scene.py): a generated Blender-format scene, baseline.yaml's
            # schedule through the HIP training path, the same weights rendered in every mode against the ground-truth images
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import trained_scene
            print("[bench] parity.trained_scene: training a field on the generated scene ...", file=sys.stderr, flush=True)
            par["trained_scene"] = trained_scene.run(net=args.net, train_mode="bf16")     # trained in bf16 (its exponent range suits the unscaled gradients); every mode renders the SAME weights
            par["headline_mode"] = {"mode": args.mode,
                                    "meets_0.01dB_on_band": par[args.mode]["meets_0.01dB"],
                                    "meets_0.01dB_on_trained_scene": bool(par["trained_scene"]["train"][args.mode]["meets_0.01dB"]
                                                                          and par["trained_scene"]["test"][args.mode]["meets_0.01dB"])}

    if single and not args.no_extras:
        def time_frames(mdl, mode, reps, **kw):
            """mean HIP-event ms of `reps` whole-frame launches (one warm-up first)"""
            N.render_camera(mdl, H, W, focal, c2w, 2.0, 6.0, S, mma_mode=mode, device=dev, dino=dino, **kw)
            torch.cuda.synchronize()
            ev = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                res = N.render_camera(mdl, H, W, focal, c2w, 2.0, 6.0, S, mma_mode=mode, device=dev, dino=dino, **kw)
                e1.record()
                ev.append((e0, e1))
            torch.cuda.synchronize()
            return sum(a.elapsed_time(b) for a, b in ev) / reps, res

        # the parity-grade fast mode next to the other modes, same frame, credited 1x the algorithmic FLOPs
        pm = {}
        for mode, reps in (("f16x3", 5), ("f16", 10), ("bf16", 10), ("f32", 3)):
            ms, _ = time_frames(model, mode, reps)
            tf = H * W * S * flops_per_sample / (ms * 1e-3) / 1e12
            pm[mode] = {"ms_per_frame": round(ms, 3), "M_ray_samples_per_s": round(H * W * S / ms / 1e3, 1), "TFLOP_per_s_credited": round(tf, 1),
                        "frac_of_2.5PF": round(tf / 2500.0, 4)}
        pm["mode"] = "f16x3"
        pm["note"] = ("f16x3 = split f16 (hi+lo operands, 3 MFMAs per product at the 32x32x16 rate): meets the 1e-4 abs and 0.01 dB bars (see parity); "
                      "f16 (the headline) meets the 0.01 dB bar only; bf16 neither; f32 = exact fp32 MFMA at 1/16 rate")
        out["parity_mode"] = pm
        # early ray termination where it can act: the coherent opaque "smooth" scene
        eps = 1e-2
        ms_mod, _ = make_model("smooth", args.mode)
        ms_off, full = time_frames(ms_mod, args.mode, 5)
        ms_on, ert = time_frames(ms_mod, args.mode, 5, ert_eps=eps)
        ms_solid_off, _ = time_frames(model, args.mode, 5)
        ms_solid_on, _ = time_frames(model, args.mode, 5, ert_eps=1e-30)         # nothing terminates: the cost of the ERT machinery alone
        out["ert"] = {"scene": "smooth", "eps": eps, "ms_off": round(ms_off, 3), "ms_on": round(ms_on, 3), "speedup": round(ms_off / ms_on, 3),
                      "max_abs_rgb_vs_full_march": float((ert[0] - full[0]).abs().max()), "max_abs_depth_vs_full_march": float((ert[1] - full[1]).abs().max()),
                      "no_termination_overhead": {"scene": args.scene, "ms_plain": round(ms_solid_off, 3), "ms_ert_kernel": round(ms_solid_on, 3),
                                                  "ratio": round(ms_solid_on / ms_solid_off, 4)}}
    if single and not args.no_train:
        out["training"] = training_line(N, args, dev)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
