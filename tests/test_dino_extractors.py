"""DINO feature-map production (nerf_few_shot_limitations_amd/dino_backbone.py,
dino_feature_model.py) against golden vectors captured from `transformers.Dinov2Model` and from the REFERENCE's own
SpatialDINOFeatures / MultiScaleDINOFeatures forward code on a tiny random-init backbone
(tests/golden/make_golden_dino.py; the published DINOv2 weights are not available offline, so image-quality parity of the
DINO variants stays unpinned).  The modules are plain PyTorch (library GEMMs): they run on the CPU here and on the GPU under
-m gpu, where the map they produce feeds the HIP sampling kernel and the fused renderer."""
import warnings

import numpy as np
import pytest
import torch

TINY = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=2, mlp_ratio=4, image_size=56, patch_size=14)


def T(a):
    return torch.from_numpy(np.asarray(a))


def state(g, prefix):
    return {k[len(prefix):]: T(v) for k, v in g.items() if k.startswith(prefix)}


def build(kind, g, device):
    from nerf_few_shot_limitations_amd import dino_backbone as B, dino_feature_model as F
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                       # "randomly initialised": the fixture's weights are loaded right after
        if kind == "bb":
            m = B.build_backbone(None, config=TINY)
        elif kind == "single":
            m = F.SpatialDINOFeatures(None, use_lora=True, lora_rank=4, lora_alpha=8, image_size=56, pos_embed_dim=8, config=TINY)
        else:
            m = F.MultiScaleDINOFeatures(None, use_lora=True, lora_rank=4, lora_alpha=8, config=TINY)
    missing, unexpected = m.load_state_dict(state(g, kind + "/"), strict=True), None
    return m.to(device).eval()


def check_all(g, device, tol):
    with torch.no_grad():
        bb = build("bb", g, device)
        y = bb(pixel_values=T(g["bb_x56"]).to(device)).last_hidden_state
        assert y.shape == (2, 17, 64) and float((y.cpu() - T(g["bb_y56"])).abs().max()) <= tol
        y = bb(pixel_values=T(g["bb_xrect"]).to(device)).last_hidden_state          # 5 x 7 patches: resampled position embeddings
        assert y.shape == (1, 36, 64) and float((y.cpu() - T(g["bb_yrect"])).abs().max()) <= tol
        y = bb(pixel_values=T(g["bb_xodd"]).to(device)).last_hidden_state           # 60 x 58: the remainder of the patch grid is dropped
        assert y.shape == (1, 17, 64) and float((y.cpu() - T(g["bb_yodd"])).abs().max()) <= tol
        single = build("single", g, device)
        f = single(T(g["single_x"]).to(device))
        assert f.shape == (2, 4, 4, 64) and single.output_dim == 64
        assert float((f.cpu() - T(g["single_y"])).abs().max()) <= tol
        multi = build("multi", g, device)
        f = multi(T(g["multi_x"]).to(device))
        assert f.shape == (1, 8, 8, 128) and multi.output_dim == 128
        assert float((f.cpu() - T(g["multi_y"])).abs().max()) <= tol
    return single, multi


def test_extractors_match_reference_forward_cpu(golden):
    g = golden("dino_extractors")
    single, multi = check_all(g, "cpu", 2e-5)
    # LoRA wrappers sit where the reference injects them, frozen originals, trainable A / B (dino_feature_model.py:16-22, 68-76)
    att = single.backbone.encoder.layer[0].attention.attention
    from nerf_few_shot_limitations_amd.dino_feature_model import LoRALinear
    assert all(isinstance(getattr(att, n), LoRALinear) for n in ("query", "key", "value"))
    assert not att.query.original.weight.requires_grad and att.query.lora_A.weight.requires_grad and att.query.scaling == 2.0
    names = [n for n, p in single.named_parameters() if p.requires_grad]
    assert all(("lora" in n) or n.startswith(("spatial_pos_embed", "feature_proj")) for n in names)
    # a fresh wrapper adds exactly nothing (lora_B = 0), dropout only acts in training mode
    lin = torch.nn.Linear(8, 8)
    w = LoRALinear(lin, rank=2, alpha=4).eval()
    x = torch.randn(3, 8)
    assert torch.equal(w(x), lin(x))


def test_backbone_checkpoint_loader_and_config(tmp_path, golden):
    from nerf_few_shot_limitations_amd import dino_backbone as B
    g = golden("dino_extractors")
    sd = state(g, "bb/")
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in sd.items()}, str(tmp_path / "model.safetensors"))
    bb = B.build_backbone(None, weights=str(tmp_path), config=TINY).eval()         # a directory, as from_pretrained takes
    with torch.no_grad():
        y = bb(pixel_values=T(g["bb_x56"])).last_hidden_state
    assert float((y - T(g["bb_y56"])).abs().max()) <= 2e-5
    torch.save({"dinov2." + k: v for k, v in sd.items()}, str(tmp_path / "pytorch_model.bin"))   # prefixed keys of a task checkpoint
    bb2 = B.build_backbone(None, weights=str(tmp_path / "pytorch_model.bin"), config=TINY)
    assert torch.equal(bb2.layernorm.weight, bb.layernorm.weight)
    with pytest.raises(RuntimeError):
        B.load_backbone_weights(bb, _bad(tmp_path, sd))
    cfg = B.dinov2_config("facebook/dinov2-base")
    assert (cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.patch_size) == (768, 12, 12, 14)
    with pytest.raises(ValueError):
        B.dinov2_config("facebook/unknown")
    with pytest.warns(UserWarning, match="randomly initialised"):
        B.build_backbone(None, config=TINY)
    with pytest.raises(ValueError):
        bb(pixel_values=torch.zeros(1, 3, 10, 56))                                   # smaller than one patch


def _bad(tmp_path, sd):
    bad = dict(sd)
    bad.pop("layernorm.weight")
    p = str(tmp_path / "bad.bin")
    torch.save(bad, p)
    return p


def test_pil_preprocessing_shapes():
    from PIL import Image
    from nerf_few_shot_limitations_amd.dino_backbone import preprocess_pil, IMAGENET_MEAN, IMAGENET_STD
    im = Image.fromarray((np.random.RandomState(0).rand(300, 400, 3) * 255).astype(np.uint8))
    x = preprocess_pil([im, im.resize((128, 128))])
    assert x.shape == (2, 3, 224, 224)
    white = preprocess_pil([Image.new("RGB", (256, 256), (255, 255, 255))])
    for c in range(3):
        assert abs(float(white[0, c].mean()) - (1 - IMAGENET_MEAN[c]) / IMAGENET_STD[c]) < 1e-5


@pytest.mark.gpu
def test_extractors_on_gpu_feed_the_hip_path(golden):
    """The maps produced on the GPU equal the reference's, and flow into the HIP side: sample_features_at_points (staged kernel)
    against the oracle's grid_sample restatement, and a fused V3 render that takes the map as its side channel."""
    import nerf_few_shot_limitations_amd as N
    from oracle import nerf_oracle as O
    g = golden("dino_extractors")
    single, multi = check_all(g, "cuda", 1e-4)
    with torch.no_grad():
        fmap = single(T(g["single_x"]).cuda())[:1]                                   # (1,4,4,64)
        fmap_w = multi(T(g["multi_x"]).cuda())                                       # (1,8,8,128)
    xy = torch.from_numpy(O.uniform01(71, 500 * 2).reshape(500, 2) * 2.4 - 1.2).float()
    for fm, mod in ((fmap, single), (fmap_w, multi)):
        got = mod.sample_features_at_points(fm, xy.cuda())
        assert float((got.cpu() - O.sample_features_at_points(fm.cpu(), xy)).abs().max()) <= 1e-5
    H = W = 16
    c2w = T(O.LEGO_LIKE_C2W)
    p = O.make_weights("v3", 2, "fog")
    m = N.NeRFMLP(pos_freq=12, dir_freq=4, hidden_dim=256, num_density_layers=8, use_dino=True, dino_dim=64, mma_mode="f16x3")
    m.load_state_dict(p, strict=False)
    m = m.cuda().eval()
    dino = dict(features=fmap, pose=c2w, focal=O.focal_for(W), H=H, W=W)
    rgb, depth = N.render_camera(m, H, W, O.focal_for(W), c2w, 2.0, 6.0, 16, dino=dino)
    ro, rd = O.get_rays(H, W, O.focal_for(W), c2w)
    ref = O.render_rays(p, "v3", ro, rd, 2.0, 6.0, 16, dino=dict(dino, features=fmap.cpu()))
    assert float((rgb.cpu() - ref["rgb"]).abs().max()) <= 1e-4 and float((depth.cpu() - ref["depth"]).abs().max()) <= 1e-4
