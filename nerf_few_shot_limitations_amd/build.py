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

# The fused renderer / forward of every network family is compiled twice (fused_impl.hpp, NRF_TU_HALF): once for the 16-bit
# MFMA modes -- with VGPR-form MFMAs, so that the pinned walk can park finished operand images in the AGPR half of the register
# file (mlp_core.hpp NRF_PARK_ACT) -- and once for the fp32-class modes (split-f16, fp32 MFMA) with hipcc's own choice.
HALF16 = ["-DNRF_TU_HALF=16", "-DNRF_ACT_AGPR=1", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]
HALF32 = ["-DNRF_TU_HALF=32"]
FUSED = [(f"fused_{fam}.hip", f"fused_{fam}_{half}", flags) for fam in ("v1", "v2", "v3", "v3w") for half, flags in (("16", HALF16), ("32", HALF32))]
SOURCES = FUSED + [(f, os.path.splitext(f)[0], []) for f in
                   ("fused_kernels.hip", "train_v1.hip", "train_v2.hip", "train_v3.hip", "staged_kernels.hip", "api.cpp", "packing.cpp")]
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


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str = LIB, obj_dir: str = OBJ, fused_only: bool = False) -> str:
    """Compile (if stale) and return the path of libnerfhip.so.  `extra_flags`/`out`/`obj_dir` build a tuning variant
    beside the product (loaded with NRF_LIB=<path> for same-box A/B runs); `fused_only` recompiles only the fused
    renderer / forward objects with the extra flags and reuses the product build's other objects."""
    deps = _deps()
    if not force and os.path.exists(out) and os.path.getmtime(out) >= _newest(deps):
        return out
    os.makedirs(obj_dir, exist_ok=True)

    def one(item):
        src, name, flags = item
        obj = os.path.join(obj_dir, name + ".o")
        if fused_only and item not in FUSED:
            shutil.copyfile(os.path.join(OBJ, name + ".o"), obj)
            return obj
        cmd = [hipcc(), *FLAGS, *flags, *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with cf.ThreadPoolExecutor(max_workers=os.cpu_count() or 8) as ex:
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
    # python -m nerf_few_shot_limitations_amd.build [--force] [--variant NAME [--fused-only] -DNRF_PREFETCH=6 ...]
    argv = sys.argv[1:]
    if "--variant" in argv:
        name = argv[argv.index("--variant") + 1]
        flags = [a for a in argv if a.startswith("-D")]
        path = build(force=True, verbose=False, extra_flags=flags, out=os.path.join(PKG, f"libnerfhip_{name}.so"),
                     obj_dir=os.path.join(PKG, "build", name), fused_only="--fused-only" in argv)
    else:
        path = build(force="--force" in argv, verbose=True)
    print(path)
