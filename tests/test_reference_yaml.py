"""The reference's experiments/*.yaml must load unchanged (last row).  The files live only
in the build container (/root/reference is absent on the GPU box): the test skips where they are not present."""
import glob
import os

import pytest

REF = os.environ.get("NERF_REFERENCE", "/root/reference")
YAMLS = sorted(glob.glob(os.path.join(REF, "experiments", "*.yaml")))


@pytest.mark.skipif(not YAMLS, reason="reference configs not present on this machine")
@pytest.mark.parametrize("path", YAMLS, ids=[os.path.basename(p) for p in YAMLS])
def test_experiment_yaml_loads(path):
    import nerf_few_shot_limitations_amd as N
    cfg = N.load_config(path)
    name = os.path.basename(path)
    if name == "projection.yaml":
        # flat schema that no script of the reference reads (SURVEY.md section 5, config / flags)
        assert "nerf_model" not in cfg or "pos_freq" not in cfg.get("nerf_model", {})
        return
    near, far = N.resolve_near_far(cfg)
    assert (near, far) == (2.0, 6.0)
    use_dino = bool(cfg["model"].get("use_dino", True))
    dino_dim = 128 if cfg["model"].get("dino_model_type") == "multi_scale" else 64     # multi_scale_dino.py:50 / dino_feature_model.py:66
    m = N.model_from_config(cfg, dino_dim=dino_dim)
    rs = N.render_settings(cfg)
    assert rs["n_samples"] == 64 and rs["chunk_size"] in (1024, 2048)
    if not use_dino:
        assert m.net == 2 and m.pos_freq == 10 and m.flops_per_sample() == 1170560
    else:
        assert m.net == 3 and m.pos_freq == 12 and m.dino_dim == dino_dim
        if dino_dim == 64:
            assert m.flops_per_sample() == 1837952
    keys = set(m.state_dict())
    assert "density_mlp.density_layers.14.weight" in keys and "color_mlp.color_layers.4.bias" in keys
