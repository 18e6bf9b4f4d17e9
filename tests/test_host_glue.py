"""Host-side glue either side of the hot path: the Blender loader and the metrics.
The reference's own modules for these need torchvision / torchmetrics (absent): parity unpinned, behaviour tested."""
import json
import math
import os

import numpy as np
import pytest
import torch
from PIL import Image

import nerf_few_shot_limitations_amd as N


def _make_scene(root, n=3, size=12):
    os.makedirs(os.path.join(root, "test"), exist_ok=True)
    frames = []
    rng = np.random.RandomState(0)
    for i in range(n):
        arr = rng.randint(0, 256, (size, size, 4)).astype(np.uint8)
        Image.fromarray(arr, "RGBA").save(os.path.join(root, "test", f"r_{i}.png"))
        m = np.eye(4); m[0, 3] = i
        frames.append({"file_path": f"./test/r_{i}", "transform_matrix": m.tolist()})
    with open(os.path.join(root, "transforms_test.json"), "w") as f:
        json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, f)


def test_blender_loader(tmp_path):
    root = str(tmp_path)
    _make_scene(root, n=3, size=12)
    imgs, poses, (H, W, focal) = N.load_blender_data(root, "test")
    assert imgs.shape == (3, 3, 12, 12) and poses.shape == (3, 4, 4) and (H, W) == (12, 12)
    assert imgs.dtype == torch.float32 and 0 <= float(imgs.min()) and float(imgs.max()) <= 1
    assert abs(focal - 0.5 * 12 / math.tan(0.5 * 0.6911112070083618)) < 1e-9          # data_loader.py:62
    assert float(poses[2, 0, 3]) == 2.0
    ref = np.asarray(Image.open(os.path.join(root, "test", "r_1.png")).convert("RGB"), np.float32) / 255
    assert np.array_equal(imgs[1].permute(1, 2, 0).numpy(), ref)
    # img_size overrides half_res and rescales the focal by img_size / W_orig (data_loader.py:37-39)
    imgs6, _, (H6, W6, f6) = N.load_blender_data(root, "test", img_size=6, half_res=True)
    assert imgs6.shape == (3, 3, 6, 6) and abs(f6 - 0.5 * 6 / math.tan(0.5 * 0.6911112070083618) * 0.5) < 1e-9
    want = np.asarray(Image.open(os.path.join(root, "test", "r_0.png")).convert("RGB").resize((6, 6), Image.LANCZOS), np.float32) / 255
    assert np.array_equal(imgs6[0].permute(1, 2, 0).numpy(), want)
    _, _, (Hh, Wh, fh) = N.load_blender_data(root, "test", half_res=True)
    assert (Hh, Wh) == (6, 6) and abs(fh - 0.5 * 6 / math.tan(0.5 * 0.6911112070083618) * 0.5) < 1e-9
    with pytest.raises(FileNotFoundError):
        N.load_blender_data(root, "train")


def test_psnr_ssim_and_png(tmp_path):
    a = torch.rand(24, 20, 3)
    assert N.psnr(a, a) == float("inf")
    b = (a + 0.1).clamp(0, 1)
    mse = float(((a - b) ** 2).mean())
    assert abs(N.psnr(a, b) + 10 * math.log10(mse)) < 1e-5                     # train_multiscale.py:294-295
    assert abs(N.ssim(a, a) - 1.0) < 1e-6
    noisy1 = (a + 0.05 * torch.randn_like(a)).clamp(0, 1)
    noisy2 = (a + 0.30 * torch.randn_like(a)).clamp(0, 1)
    assert 1.0 > N.ssim(a, noisy1) > N.ssim(a, noisy2) > -1.0
    assert abs(N.ssim(a, noisy1) - N.ssim(noisy1, a)) < 1e-6
    p = os.path.join(str(tmp_path), "out", "render_0.png")
    N.save_png(p, a)
    back = np.asarray(Image.open(p), np.float32) / 255
    assert back.shape == (24, 20, 3) and np.abs(back - a.numpy()).max() <= 1 / 255 + 1e-6


def test_ssim_against_an_independent_restatement():
    """The SSIM of evaluation.py against a scipy restatement of the same published definition (Wang et al. 2004, as the metric
    the reference instantiates computes it: 11-tap Gaussian of sigma 1.5 applied separably with mirrored borders, K1 = .01,
    K2 = .03, data range from the data, border of 5 pixels cropped from the index map).  torchmetrics itself is not importable
    here: parity with it stays unpinned; this pins the arithmetic to the definition."""
    from scipy.ndimage import correlate1d
    rng = np.random.RandomState(3)
    a = rng.rand(40, 33, 3).astype(np.float32) * 0.8 + 0.1
    b = np.clip(a + 0.1 * rng.randn(40, 33, 3).astype(np.float32), 0, 1)
    x = np.arange(11) - 5.0
    g = np.exp(-(x ** 2) / (2 * 1.5 ** 2)); g /= g.sum()

    def blur(img):                                             # per channel, rows then columns, mirrored (no edge repeat) borders
        return correlate1d(correlate1d(img.astype(np.float64), g, axis=0, mode="mirror"), g, axis=1, mode="mirror")

    def ref_ssim(p, t, data_range, crop):
        c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
        mp, mt = blur(p), blur(t)
        spp, stt, spt = blur(p * p) - mp ** 2, blur(t * t) - mt ** 2, blur(p * t) - mp * mt
        m = ((2 * mp * mt + c1) * (2 * spt + c2)) / ((mp ** 2 + mt ** 2 + c1) * (spp + stt + c2))
        if crop:
            m = m[5:-5, 5:-5]
        return float(m.mean())

    dr = max(a.max() - a.min(), b.max() - b.min())
    assert abs(N.ssim(torch.from_numpy(a), torch.from_numpy(b)) - ref_ssim(a, b, dr, True)) < 2e-6
    assert abs(N.ssim(torch.from_numpy(a), torch.from_numpy(b), data_range=1.0, crop_border=False) - ref_ssim(a, b, 1.0, False)) < 2e-6
    assert abs(N.ssim(torch.from_numpy(a).permute(2, 0, 1), torch.from_numpy(b).permute(2, 0, 1)) - ref_ssim(a, b, dr, True)) < 2e-6


def test_ssim_reproduces_the_metric_librarys_documented_example():
    """A known answer from outside this repository: the docstring example of `torchmetrics.StructuralSimilarityIndexMeasure`
    (the metric the reference instantiates, train.py:100) -- `preds = torch.rand([3, 3, 256, 256]); target = preds * 0.75;
    StructuralSimilarityIndexMeasure(data_range=1.0)(preds, target)` prints `tensor(0.9219)`.  On uniform noise the value does not
    depend on the draw to the digits shown (0.92189 for every seed tried), so it pins the definition this module restates
    (11 x 11 Gaussian window of sigma 1.5, K1 = 0.01, K2 = 0.03, reflect padding, border crop) without the library being importable."""
    for seed in (42, 7):
        torch.manual_seed(seed)
        preds = torch.rand([3, 3, 256, 256])
        target = preds * 0.75
        v = float(np.mean([N.ssim(preds[i], target[i], data_range=1.0) for i in range(3)]))
        assert abs(v - 0.9219) < 5e-5, v


def test_entry_points_and_tools_compile():
    """The driver imports __graft_entry__ and runs bench.py as scripts: a syntax error there is invisible to every other test."""
    import glob
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in [os.path.join(root, "__graft_entry__.py"), os.path.join(root, "bench.py")] + sorted(glob.glob(os.path.join(root, "tools", "*.py"))):
        py_compile.compile(path, doraise=True)
    import importlib
    entry = importlib.import_module("__graft_entry__")
    assert callable(entry.build) and callable(entry.smoke)
