"""Build libnerfhip.so (gfx950) in-tree with hipcc.

    python -m nerf_few_shot_limitations_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build
container; the resulting .so (git-ignored) travels to the GPU box with the
repository snapshot.  There is exactly one target architecture and no fallback.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "build")
LIB = os.path.join(PKG, "libnerfhip.so")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")

SOURCES = ["fused_v1.hip", "fused_v2.hip", "fused_v3.hip", "fused_v3w.hip", "fused_kernels.hip", "train_v1.hip", "train_v2.hip", "train_v3.hip", "staged_kernels.hip", "api.cpp", "packing.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
         "-fno-gpu-rdc", "-ffp-contract=off", f"-I{INCLUDE}"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libnerfhip.so cannot be built (ROCm toolchain required)")
    return exe


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def _deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "nerfhip.h")]


def _compile(src: str, verbose: bool) -> str:
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    cmd = [hipcc(), *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return obj


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str = LIB, obj_dir: str = OBJ) -> str:
    """Compile (if stale) and return the path of libnerfhip.so.  `extra_flags`/`out`/`obj_dir` build a tuning variant
    beside the product (loaded with NRF_LIB=<path> for same-process-free A/B runs on one GPU box)."""
    deps = _deps()
    if not force and os.path.exists(out) and os.path.getmtime(out) >= _newest(deps):
        return out
    os.makedirs(obj_dir, exist_ok=True)

    def one(src):
        obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc(), *FLAGS, *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with cf.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(one, SOURCES))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out + ".tmp", *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    # python -m nerf_few_shot_limitations_amd.build [--force] [--variant NAME -DNRF_PREFETCH=6 ...]
    argv = sys.argv[1:]
    if "--variant" in argv:
        name = argv[argv.index("--variant") + 1]
        flags = [a for a in argv if a.startswith("-D")]
        path = build(force=True, verbose=False, extra_flags=flags, out=os.path.join(PKG, f"libnerfhip_{name}.so"),
                     obj_dir=os.path.join(PKG, "build", name))
    else:
        path = build(force="--force" in argv, verbose=True)
    print(path)
