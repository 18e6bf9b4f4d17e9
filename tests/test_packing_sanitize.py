"""The host-side weight packer under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on the
pool; the packer, the plan builders and the source tables are plain C++): tests/host/packing_sanitize.cpp walks every network
family (V1, V2, V3 at dino_dim 64 / 128), trunk depths 1 / 2 / 5 / 8 and every arithmetic mode -- forward and transposed
streams, device re-pack source tables, bias tables, training plans -- and two malformed architectures."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_packer_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "pack_san")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-D__host__=", "-D__device__=", f"-I{os.path.join(ROOT, 'include')}", "-o", exe,
           os.path.join(ROOT, "tests", "host", "packing_sanitize.cpp"), os.path.join(ROOT, "nerf_few_shot_limitations_amd", "csrc", "packing.cpp")]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "asan" in (b.stderr + b.stdout).lower() and "cannot find" in (b.stderr + b.stdout).lower():
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "sanitize ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    assert r.stdout.count(" ok:") == 16
