"""CPU tests of bench.py's host logic: `python bench.py --gpus N` with no launcher starts N fresh ranks through
torch.distributed.run BEFORE torch is imported or a GPU touched, forwards every flag, and relays the children's status."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_self_launch_builds_the_torchrun_command(monkeypatch):
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = list(cmd), dict(env)
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    argv = ["--gpus", "4", "--steps", "3", "--warmup", "1", "--rehearse", "--mode", "bf16"]
    with pytest.raises(SystemExit) as e:
        bench.main(argv)
    assert e.value.code == 7                                     # the children's status is the parent's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == argv                              # every flag reaches the ranks unchanged
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "torch" not in bench.__dict__                         # the parent never imported torch at module level


def test_launcher_environment_mismatch_is_refused(monkeypatch):
    import bench
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "4"])
    assert "WORLD_SIZE=2" in str(e.value.code)


def test_headline_mode_is_declared_once():
    import bench
    assert bench.HEADLINE_MODE in ("f16", "bf16", "f16x3", "f32")
    import __graft_entry__ as g
    assert g.HEADLINE_MODE == bench.HEADLINE_MODE
