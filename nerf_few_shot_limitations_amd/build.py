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
# -Rpass-analysis=kernel-resource-usage: the backend reports every kernel's registers / spills / scratch; kept beside the object
# (<name>.o.remarks, see kernel_resources()) so that a toolchain or flag change that breaks the AGPR parking or introduces
# spills in a headline kernel is caught by tests/test_kernel_resources.py
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
         "-fno-gpu-rdc", "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage", f"-I{INCLUDE}"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libnerfhip.so cannot be built (ROCm toolchain required)")
    return exe


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def _deps():
    # (build.py's flag lists -- VGPR-form MFMAs, AGPR parking, per-half defines -- decide the code too: every object records the
    # command that built it and is rebuilt when that changes, see _obj_fresh)
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if os.path.isfile(os.path.join(CSRC, f))] + [os.path.join(INCLUDE, "nerfhip.h"), os.path.abspath(__file__)]


def _obj_fresh(obj: str, cmd_key: str) -> bool:
    """An object is reused when the command that built it is unchanged (recorded beside it: flag lists are part of the code) and it
    is newer than every file hipcc read for it (its -MD dependency list: the source and the headers it includes)."""
    dep, key = obj + ".d", obj + ".cmd"
    if not (os.path.exists(obj) and os.path.exists(dep) and os.path.exists(key)):
        return False
    with open(key) as f:
        if f.read() != cmd_key:
            return False
    with open(dep) as f:
        words = f.read().replace("\\\n", " ").split()
    files = [w for w in words[1:] if not w.endswith(":")]
    t = os.path.getmtime(obj)
    return all(os.path.exists(p) and os.path.getmtime(p) <= t for p in files)


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str = LIB, obj_dir: str = OBJ, fused_only: bool = False, only=None) -> str:
    """Compile (what is stale) and return the path of libnerfhip.so.  `extra_flags`/`out`/`obj_dir` build a tuning variant
    beside the product (loaded with NRF_LIB=<path> for same-box A/B runs); `fused_only` recompiles only the fused
    renderer / forward objects with the extra flags and reuses the product build's other objects; `only` = the object names
    (e.g. {"train_v1"}) to recompile instead."""
    deps = _deps()
    # fast path (and the only one on the GPU box, where the object directory does not travel): the library is newer than every
    # source, header and this file
    if not force and not extra_flags and os.path.exists(out) and os.path.getmtime(out) >= _newest(deps):
        return out
    os.makedirs(obj_dir, exist_ok=True)

    def one(item):
        src, name, flags = item
        obj = os.path.join(obj_dir, name + ".o")
        if (fused_only and item not in FUSED) or (only is not None and name not in only):
            # a variant that recompiles only the fused kernels links the PRODUCT build's other objects: they must exist and be
            # newer than every source (else the A/B would silently compare against stale code)
            prod = os.path.join(OBJ, name + ".o")
            prod_cmd = " ".join([hipcc(), *FLAGS, *flags, "-MD", "-MF", prod + ".d", "-c", os.path.join(CSRC, src), "-o", prod])
            if not _obj_fresh(prod, prod_cmd):
                raise RuntimeError(f"variant build: product object {prod} is missing or older than a file it was built from; run the plain build first")
            shutil.copyfile(prod, obj)
            return obj, False
        cmd = [hipcc(), *FLAGS, *flags, *extra_flags, "-MD", "-MF", obj + ".d", "-c", os.path.join(CSRC, src), "-o", obj]
        cmd_key = " ".join(cmd)
        if not force and _obj_fresh(obj, cmd_key):
            return obj, False
        if verbose:
            print(cmd_key, flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        with open(obj + ".remarks", "w") as f:
            f.write(r.stderr)
        with open(obj + ".cmd", "w") as f:
            f.write(cmd_key)
        return obj, True

    with cf.ThreadPoolExecutor(max_workers=os.cpu_count() or 8) as ex:
        res = list(ex.map(one, SOURCES))
    objs = [o for o, _ in res]
    if not any(built for _, built in res) and os.path.exists(out) and os.path.getmtime(out) >= _newest(objs):
        os.utime(out)          # nothing to do (e.g. only a comment of this file changed): restore the fast path
        return out
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out + ".tmp", *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(out + ".tmp", out)
    return out


def kernel_resources(obj_dir: str = OBJ):
    """{demangled kernel name: {'vgprs', 'agprs', 'sgprs', 'scratch', 'vgpr_spill', 'sgpr_spill', 'occupancy', 'tu'}} of the last build,
    parsed from the backend's kernel-resource-usage remarks (hipcc's stderr, saved per object)."""
    import re
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch", "VGPRs Spill": "vgpr_spill",
            "SGPRs Spill": "sgpr_spill", "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds"}
    out, mangled = {}, []
    for fn in sorted(os.listdir(obj_dir)):
        if not fn.endswith(".o.remarks"):
            continue
        cur = None
        with open(os.path.join(obj_dir, fn)) as f:
            for line in f:
                m = re.search(r"remark:\s+Function Name: (\S+)", line)
                if m:
                    cur = {"tu": fn[: -len(".o.remarks")]}
                    out[(fn, m.group(1))] = cur
                    mangled.append((fn, m.group(1)))
                    continue
                m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]+\])?): (\d+)", line)
                if m and cur is not None and m.group(1).strip() in keys:
                    cur[keys[m.group(1).strip()]] = int(m.group(2))
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/usr/bin/c++filt"
    names = [n for _, n in mangled]
    if names and os.path.exists(filt):
        names = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return {f"{out[k]['tu']}: {n}": out[k] for k, n in zip(mangled, names)}


if __name__ == "__main__":
    # python -m nerf_few_shot_limitations_amd.build [--force] [--variant NAME [--fused-only | --only train_v1,...] -DNRF_PREFETCH=6 ...]
    argv = sys.argv[1:]
    if "--variant" in argv:
        name = argv[argv.index("--variant") + 1]
        flags = [a for a in argv if a.startswith("-D")]
        only = set(argv[argv.index("--only") + 1].split(",")) if "--only" in argv else None
        path = build(force=True, verbose=False, extra_flags=flags, out=os.path.join(PKG, f"libnerfhip_{name}.so"),
                     obj_dir=os.path.join(PKG, "build", name), fused_only="--fused-only" in argv, only=only)
    else:
        path = build(force="--force" in argv, verbose=True)
    print(path)
