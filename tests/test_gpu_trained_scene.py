"""The PSNR bar on a TRAINED field (BASELINE.json: "PSNR within 0.01 dB of reference on lego"; README.md:31 of the reference,
src/training/train.py:294-342).  lego is not available offline; tools/trained_scene.py generates a Blender-format scene with
ground-truth images, trains a field on it through the HIP training path and renders the same weights in every arithmetic mode.

Asserted: the mode bench.py runs by default (bench.HEADLINE_MODE) stays within 0.01 dB of the fp32 render on the train views
(tightest fit) AND on the held-out views; the parity-grade modes stay within 1e-4 abs of each other on rgb.  bf16's miss is
recorded, not asserted (it is the reason it is not the headline)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def result():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import bench
    import trained_scene
    return trained_scene.run(net="v2", train_mode="bf16"), bench.HEADLINE_MODE


def test_field_actually_trained(result):
    r, _ = result
    assert r["loss_last_epoch"] < 0.25 * r["loss_first_epoch"], r
    assert r["train"]["f32"]["psnr_db"] > 24.0, r["train"]["f32"]          # a fitted field, not the collapsed all-background one
    assert r["test"]["f32"]["psnr_db"] > 15.0, r["test"]["f32"]


def test_headline_mode_within_0p01_db_on_trained_field(result):
    r, mode = result
    for split in ("train", "test"):
        assert r[split][mode]["psnr_delta_vs_f32_db"] <= 0.01, (split, r[split][mode])


def test_parity_grade_mode_within_1e4_on_trained_field(result):
    r, _ = result
    for split in ("train", "test"):
        assert r[split]["f16x3"]["max_abs_rgb_vs_f32"] <= 1e-4, (split, r[split]["f16x3"])
        assert r[split]["f16x3"]["psnr_delta_vs_f32_db"] <= 1e-3
