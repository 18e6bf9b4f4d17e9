"""CPU oracle for the ray-marching hot path (TEST INFRASTRUCTURE -- NOT PRODUCT CODE).

This file is a from-scratch CPU restatement (torch fp32 on CPU + numpy) of the
algorithm of the reference's hot path
    src/models/ray_sampler.py -> positional_encoding.py -> nerf_model.py /
    nerf_mlp.py -> volume_renderer.py            (paths below are relative to
                                                  /root/reference)
It exists only so that `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` can check / time the HIP path against it.
Nothing under `nerf_few_shot_limitations_amd/` imports it; the product path
fails loudly when `libnerfhip.so` is missing instead of falling back to this.

Pinning: the reference ships no tests, fixtures or golden vectors of its own
(SURVEY.md section 4), so this oracle is pinned by
  * the closed-form known-answer values K1..K6 of SURVEY.md section 4, and
  * golden vectors captured by importing the reference's leaf modules in the
    build container (`tests/golden/make_golden.py` -> `tests/golden/*.npz`),
both checked in `tests/test_oracle_golden.py`.
`sample_pdf` (hierarchical sampling) is the one exception: the reference's
`hierarchical_sampling` raises on every input, so that function
is a restatement of its *intent* and its parity is UNPINNED.

All arithmetic is fp32, like the reference.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# a1  rays
# --------------------------------------------------------------------------

def get_rays(H: int, W: int, focal: float, c2w: torch.Tensor):
    """Pinhole rays, one per pixel, row-major ray id r = y*W + x.

    Follows src/models/ray_sampler.py:4-30 (== src/utils/ray_utils.py:4-37):
    x = column index, y = row index, dirs = [(x-W/2)/f, -(y-H/2)/f, -1],
    rays_d = R @ dirs (as a broadcast multiply + sum over the last axis),
    rays_o = translation column broadcast. No half-pixel offset, no
    normalisation.
    """
    c2w = torch.as_tensor(c2w, dtype=torch.float32)
    xs = torch.arange(W, dtype=torch.float32)
    ys = torch.arange(H, dtype=torch.float32)
    x = xs[None, :].expand(H, W)
    y = ys[:, None].expand(H, W)
    dirs = torch.stack([(x - W * 0.5) / focal, -(y - H * 0.5) / focal, -torch.ones_like(x)], -1)
    rot = c2w[:3, :3]
    rays_d = (dirs[..., None, :] * rot).sum(-1)
    rays_o = c2w[:3, 3].expand(rays_d.shape)
    return rays_o, rays_d


# --------------------------------------------------------------------------
# a2  stratified samples along rays
# --------------------------------------------------------------------------

def z_steps(near: float, far: float, n_samples: int, lindisp: bool = False) -> torch.Tensor:
    """The per-ray depth ladder before jitter.

    src/utils/ray_utils.py:58-66 (lindisp branch :59-62; depth branch :64-66),
    src/models/ray_sampler.py:49-50: t = linspace(0,1,S);
    z = near*(1-t) + far*t   or   1/(1/near*(1-t) + 1/far*t).
    """
    t = torch.linspace(0.0, 1.0, n_samples, dtype=torch.float32)
    if lindisp:
        return 1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t)
    return near * (1.0 - t) + far * t


def sample_points_along_rays(rays_o, rays_d, near, far, n_samples, t_rand=None, lindisp=False):
    """pts = o + d*z for S depths per ray; optional stratified jitter.

    src/utils/ray_utils.py:39-84 (flat layout) == src/models/ray_sampler.py:32-61
    (image layout); both layouts are handled by broadcasting on the leading
    axes. `t_rand` (same shape as z_vals, U[0,1)) replaces the reference's
    in-function `torch.rand` (ray_utils.py:78) so that the jitter is an input:
    mids = .5*(z[1:]+z[:-1]); upper=[mids,z_last]; lower=[z_0,mids];
    z = lower + (upper-lower)*t_rand.  t_rand=None <=> perturb=False.
    """
    rays_o = torch.as_tensor(rays_o, dtype=torch.float32)
    rays_d = torch.as_tensor(rays_d, dtype=torch.float32)
    lead = rays_o.shape[:-1]
    z = z_steps(near, far, n_samples, lindisp).expand(*lead, n_samples)
    if t_rand is not None:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * torch.as_tensor(t_rand, dtype=torch.float32)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
    return pts, z


# --------------------------------------------------------------------------
# a3  hierarchical (inverse-cdf) resampling -- PARITY UNPINNED
# --------------------------------------------------------------------------

def sample_pdf(z_vals, weights, n_importance, u=None):
    """Importance samples from the coarse weights; returns (new z, sorted union).

    Restates the INTENT of src/utils/ray_utils.py:86-143, which raises on every
    input (gathers S-entry `z_vals` with indices up to S).  Every
    line is kept (weights+1e-5 :104, pdf/cdf with leading 0 :107-109,
    u = linspace(0,1,Ni) when not perturbing :115-116, searchsorted right=True
    :120, below/above clamps :121-122, denom<1e-5 -> 1 :131, linear interp
    :132-133, sorted union :136) except that the S+1 cdf knots are paired with
    S+1 bin EDGES  [z_0, mids..., z_{S-1}]  -- the very intervals the reference
    assigns to each sample when stratifying (ray_utils.py:73-75) -- instead of
    the S raw z values.  No reference output exists for this function.
    """
    z_vals = torch.as_tensor(z_vals, dtype=torch.float32)
    weights = torch.as_tensor(weights, dtype=torch.float32)
    n_rays, n_samples = z_vals.shape
    w = weights + 1e-5
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)          # (R, S+1)
    if u is None:
        u = torch.linspace(0.0, 1.0, n_importance, dtype=torch.float32).expand(n_rays, n_importance)
    u = torch.as_tensor(u, dtype=torch.float32).contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(idx - 1, min=0)
    above = torch.clamp(idx, max=cdf.shape[-1] - 1)
    mids = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
    edges = torch.cat([z_vals[..., :1], mids, z_vals[..., -1:]], -1)   # (R, S+1)
    cdf_b = torch.gather(cdf, 1, below)
    cdf_a = torch.gather(cdf, 1, above)
    bin_b = torch.gather(edges, 1, below)
    bin_a = torch.gather(edges, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    samples = bin_b + t * (bin_a - bin_b)
    union, _ = torch.sort(torch.cat([z_vals, samples], -1), -1)
    return samples, union


# --------------------------------------------------------------------------
# a4  positional encoding
# --------------------------------------------------------------------------

def positional_encoding(x, num_freqs: int, include_input: bool = True, log_sampling: bool = True):
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)].

    src/models/positional_encoding.py:13-14,27-33 == src/models/nerf_mlp.py:13-33.
    Blocks of D values, sin before cos, frequency-major.  log_sampling=False: frequencies
    linspace(1, 2^(L-1), L) (positional_encoding.py:17-18).
    """
    x = torch.as_tensor(x, dtype=torch.float32)
    if log_sampling:
        freqs = 2.0 ** torch.linspace(0.0, num_freqs - 1, num_freqs)
    else:
        freqs = torch.linspace(2.0 ** 0.0, 2.0 ** (num_freqs - 1), num_freqs)
    out = [x] if include_input else []
    for f in freqs:
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


def encoded_dim(num_freqs: int, input_dim: int = 3, include_input: bool = True) -> int:
    """src/models/nerf_mlp.py:35-39."""
    return input_dim * 2 * num_freqs + (input_dim if include_input else 0)


# --------------------------------------------------------------------------
# a5..a7  the MLPs (weights are a flat dict name -> tensor, nn.Linear layout
#         weight (out,in), bias (out,))
# --------------------------------------------------------------------------

def _lin(p, name, x):
    return F.linear(x, p[name + ".weight"], p[name + ".bias"])


def mlp_v1(p: Dict[str, torch.Tensor], x_enc: torch.Tensor) -> torch.Tensor:
    """The 8x256 NeRF MLP: (P,63) -> (P,4) = [sigmoid rgb, raw sigma].

    src/models/nerf_model.py:16-24: n_layers x (Linear, ReLU) -- ReLU after
    every layer, no skip -- then sigma_out (raw) and sigmoid(rgb_out);
    returns cat([rgb, sigma]).
    """
    h = x_enc
    i = 0
    while f"layers.{i}.weight" in p:
        h = F.relu(_lin(p, f"layers.{i}", h))
        i += 1
    sigma = _lin(p, "sigma_out", h)
    rgb = torch.sigmoid(_lin(p, "rgb_out", h))
    return torch.cat([rgb, sigma], -1)


def density_mlp(p, prefix, x):
    """src/models/nerf_mlp.py:60-66: Linear+ReLU stack, relu(density_head), feature_head (no act)."""
    h = x
    i = 0
    while f"{prefix}density_layers.{i}.weight" in p:
        h = F.relu(_lin(p, f"{prefix}density_layers.{i}", h))
        i += 2                                   # Sequential indices 0,2,4,.. (ReLUs in between)
    density = F.relu(_lin(p, f"{prefix}density_head", h))
    feat = _lin(p, f"{prefix}feature_head", h)
    return density, feat


def color_mlp(p, prefix, feat, dir_enc):
    """src/models/nerf_mlp.py:72-84: cat[feat, dir_enc] -> L+ReLU -> L+ReLU -> L+Sigmoid."""
    h = torch.cat([feat, dir_enc], -1)
    h = F.relu(_lin(p, f"{prefix}color_layers.0", h))
    h = F.relu(_lin(p, f"{prefix}color_layers.2", h))
    return torch.sigmoid(_lin(p, f"{prefix}color_layers.4", h))


def dino_fusion(p, prefix, pos_enc, dino):
    """src/models/lora_dino.py:171-193 (== dino_feature_model.py:175-197).

    fused = fusion(cat[pe, dino]); w = softmax(attention(fused));
    out = output_proj(fusion(cat[pe*w0, dino*w1]))  -- the SAME `fusion`
    weights run twice.
    """
    def fusion(x):
        h = F.relu(_lin(p, f"{prefix}fusion.0", x))
        return F.relu(_lin(p, f"{prefix}fusion.2", h))

    fused = fusion(torch.cat([pos_enc, dino], -1))
    a = F.relu(_lin(p, f"{prefix}attention.0", fused))
    w = torch.softmax(_lin(p, f"{prefix}attention.2", a), -1)
    final = fusion(torch.cat([pos_enc * w[:, 0:1], dino * w[:, 1:2]], -1))
    return _lin(p, f"{prefix}output_proj", final)


def mlp_v2(p, positions, directions, pos_freq=10, dir_freq=4):
    """Baseline under the train.py surface (use_dino=False):
    PE(pos) -> DensityMLP -> ColorMLP(feature, PE(dir)); returns (rgb (P,3), density (P,1)).

    The only composition of existing reference classes that satisfies the call
    at src/training/train.py:82-89,229:
    src/models/nerf_mlp.py:41-84 with NeRFWithDINO's wiring (:144-158) minus
    the fusion block.
    """
    pe = positional_encoding(positions, pos_freq)
    de = positional_encoding(directions, dir_freq)
    density, feat = density_mlp(p, "density_mlp.", pe)
    rgb = color_mlp(p, "color_mlp.", feat, de)
    return rgb, density


def mlp_v3(p, positions, directions, dino, pos_freq=12, dir_freq=4):
    """NeRFWithDINO.forward, src/models/nerf_mlp.py:134-158."""
    pe = positional_encoding(positions, pos_freq)
    de = positional_encoding(directions, dir_freq)
    fused = dino_fusion(p, "dino_fusion.", pe, dino)
    density, feat = density_mlp(p, "density_mlp.", fused)
    rgb = color_mlp(p, "color_mlp.", feat, de)
    return rgb, density


# --------------------------------------------------------------------------
# a8  DINO side channel: projection + bilinear fetch
# --------------------------------------------------------------------------

def project_points_to_image(points, pose, focal, H, W):
    """src/utils/ray_utils.py:176-210: p_cam = [p,1] @ inv(pose)^T;
    x = X/(Z+1e-8)*f + W/2 (same for y with H); normalise to [-1,1];
    returns (xy_norm (N,2), depth (N,), mask Z>0)."""
    pose = torch.as_tensor(pose, dtype=torch.float32)
    inv = torch.inverse(pose)
    ph = torch.cat([points, torch.ones_like(points[..., :1])], -1)
    pc = torch.matmul(ph, inv.T)[..., :3]
    mask = pc[..., 2] > 0
    x = pc[..., 0] / (pc[..., 2] + 1e-8) * focal + W / 2
    y = pc[..., 1] / (pc[..., 2] + 1e-8) * focal + H / 2
    xy = torch.stack([(x / W) * 2 - 1, (y / H) * 2 - 1], -1)
    return xy, pc[..., 2], mask


def sample_features_at_points(features, points_2d):
    """src/models/dino_feature_model.py:114-148: bilinear grid_sample of a
    (1,Hp,Wp,C) map at (N,2) normalised points, zeros padding,
    align_corners=False -> (N,C).  Written out tap by tap (no F.grid_sample)."""
    fm = torch.as_tensor(features, dtype=torch.float32)[0]          # (Hp,Wp,C)
    Hp, Wp, C = fm.shape
    gx = ((points_2d[:, 0] + 1) * Wp - 1) / 2                       # align_corners=False unnormalise
    gy = ((points_2d[:, 1] + 1) * Hp - 1) / 2
    x0 = torch.floor(gx)
    y0 = torch.floor(gy)
    out = torch.zeros(points_2d.shape[0], C, dtype=torch.float32)
    for dy in (0, 1):
        for dx in (0, 1):
            xi = x0 + dx
            yi = y0 + dy
            wx = (gx - x0) if dx else (x0 + 1 - gx)
            wy = (gy - y0) if dy else (y0 + 1 - gy)
            ok = (xi >= 0) & (xi <= Wp - 1) & (yi >= 0) & (yi <= Hp - 1)
            xi_c = xi.clamp(0, Wp - 1).long()
            yi_c = yi.clamp(0, Hp - 1).long()
            tap = fm[yi_c, xi_c]                                    # (N,C)
            out = out + tap * (wx * wy * ok)[:, None]
    return out


# --------------------------------------------------------------------------
# a9 / a10  alpha compositing
# --------------------------------------------------------------------------

def volume_render(rgb, density, z_vals, rays_d, white_bkgd=False):
    """src/models/nerf_mlp.py:165-215 (eval path, noise off).

    dists = diff(z) ++ [1e10]; dists *= |d|; alpha = 1-exp(-relu(sigma)*dists);
    T = exclusive cumprod of (1-alpha+1e-10); w = alpha*T; rgb = sum w c;
    depth = sum w z; white_bkgd adds (1 - sum w).  Returns (rgb, depth, weights).
    """
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    alpha = 1.0 - torch.exp(-F.relu(density) * dists[..., None])
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1, :]), 1.0 - alpha + 1e-10], -2), -2)[..., :-1, :]
    w = alpha * trans
    rgb_out = (w * rgb).sum(-2)
    depth = (w[..., 0] * z_vals).sum(-1)
    if white_bkgd:
        rgb_out = rgb_out + (1.0 - w.sum(-2))
    return rgb_out, depth, w[..., 0]


def volume_render_radiance(rgb_sigma, z_vals, rays_d):
    """src/models/volume_renderer.py:4-43 (noise off): same maths on the
    (H,W,S,4)=[r,g,b,sigma] layout, rgb map only."""
    rgb, _, _ = volume_render(rgb_sigma[..., :3], rgb_sigma[..., 3:4], z_vals, rays_d)
    return rgb


# --------------------------------------------------------------------------
# a11  glue: render a batch of rays / a camera
# --------------------------------------------------------------------------

def render_rays(p, variant, rays_o, rays_d, near, far, n_samples, t_rand=None,
                white_bkgd=False, pos_freq=None, dir_freq=4, dino=None, chunk=2048, z_in=None):
    """src/training/train.py:188-242 with D1-D4 repaired.

    variant 'v1': PE(10) -> nerf_model.NeRFMLP -> compositor (train_minimal.py:97-102 wiring);
            'v2': train.py baseline (use_dino False); 'v3': NeRFWithDINO, with
    `dino` = dict(features (1,Hp,Wp,C), pose (4,4), focal, H, W) (train.py:203-214).
    View directions are the raw (un-normalised) rays_d expanded over samples (train.py:225).
    Chunked over rays like the reference's eval loop (train.py:309-317).
    """
    rays_o = torch.as_tensor(rays_o, dtype=torch.float32).reshape(-1, 3)
    rays_d = torch.as_tensor(rays_d, dtype=torch.float32).reshape(-1, 3)
    R = rays_o.shape[0]
    outs = {"rgb": [], "depth": [], "weights": [], "z_vals": []}
    if pos_freq is None:
        pos_freq = 12 if variant == "v3" else 10
    with torch.no_grad():
        for b in range(0, R, chunk):
            o, d = rays_o[b:b + chunk], rays_d[b:b + chunk]
            tr = None if t_rand is None else torch.as_tensor(t_rand, dtype=torch.float32)[b:b + chunk]
            pts, z = sample_points_along_rays(o, d, near, far, n_samples, tr)
            if z_in is not None:                      # explicit depths (fine pass of hierarchical sampling, ray_utils.py:139)
                z = torch.as_tensor(z_in, dtype=torch.float32)[b:b + chunk]
                pts = o[..., None, :] + d[..., None, :] * z[..., :, None]
            n = o.shape[0]
            pf = pts.reshape(-1, 3)
            df = d[:, None, :].expand(-1, n_samples, -1).reshape(-1, 3)
            if variant == "v1":
                out = mlp_v1(p, positional_encoding(pf, pos_freq))
                rgb, sig = out[:, :3], out[:, 3:4]
            elif variant == "v2":
                rgb, sig = mlp_v2(p, pf, df, pos_freq, dir_freq)
            elif variant == "v3":
                xy, _, _ = project_points_to_image(pf, dino["pose"], dino["focal"], dino["H"], dino["W"])
                feats = sample_features_at_points(dino["features"], xy)
                rgb, sig = mlp_v3(p, pf, df, feats, pos_freq, dir_freq)
            else:
                raise ValueError(variant)
            c, dep, w = volume_render(rgb.reshape(n, n_samples, 3), sig.reshape(n, n_samples, 1), z, d, white_bkgd)
            outs["rgb"].append(c); outs["depth"].append(dep); outs["weights"].append(w); outs["z_vals"].append(z)
    return {k: torch.cat(v, 0) for k, v in outs.items()}


# --------------------------------------------------------------------------
# deterministic synthetic inputs -- no torch RNG, so the
# container and the GPU box regenerate them bit-identically
# --------------------------------------------------------------------------

CAMERA_ANGLE_X = 0.6911112070083618
LEGO_LIKE_C2W = np.array([[-0.9999, 0.0042, -0.0133, -0.0538],
                          [-0.0140, -0.2997, 0.9539, 3.8455],
                          [0.0, 0.9540, 0.2997, 1.2081],
                          [0.0, 0.0, 0.0, 1.0]], dtype=np.float32)


def focal_for(width: int, focal_scale: float = 1.0) -> float:
    """src/models/data_loader.py:62: focal = .5*W/tan(.5*camera_angle_x)*focal_scale."""
    return 0.5 * width / math.tan(0.5 * CAMERA_ANGLE_X) * focal_scale


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform01(seed: int, n: int) -> np.ndarray:
    """n floats in [0,1) from a counter-based splitmix64 hash (24 mantissa bits)."""
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(32))
        bits = _splitmix64(_splitmix64(ctr))
    return ((bits >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def _linear_init(seed: int, out_f: int, in_f: int):
    """nn.Linear default init ranges: U(-1/sqrt(in), 1/sqrt(in)) for weight and bias."""
    k = 1.0 / math.sqrt(in_f)
    w = (uniform01(seed * 2 + 0, out_f * in_f).reshape(out_f, in_f) * 2 - 1) * k
    b = (uniform01(seed * 2 + 1, out_f) * 2 - 1) * k
    return torch.from_numpy(w.astype(np.float32)), torch.from_numpy(b.astype(np.float32))


def layer_shapes(variant: str, pos_freq=None, dir_freq=4, hidden=256, n_layers=8, dino_dim=64):
    """(name, out, in) for every Linear of a variant, in state_dict order."""
    if pos_freq is None:
        pos_freq = 12 if variant == "v3" else 10
    pe, de = encoded_dim(pos_freq), encoded_dim(dir_freq)
    L = []
    if variant == "v1":                      # src/models/nerf_model.py:6-14
        for i in range(n_layers):
            L.append((f"layers.{i}", hidden, pe if i == 0 else hidden))
        L += [("sigma_out", 1, hidden), ("rgb_out", 3, hidden)]
        return L
    if variant == "v3":                      # src/models/lora_dino.py:153-169
        L += [("dino_fusion.fusion.0", hidden, pe + dino_dim), ("dino_fusion.fusion.2", hidden, hidden),
              ("dino_fusion.attention.0", hidden // 4, hidden), ("dino_fusion.attention.2", 2, hidden // 4),
              ("dino_fusion.output_proj", hidden, hidden)]
    d_in = hidden if variant == "v3" else pe  # src/models/nerf_mlp.py:118-129
    for i in range(n_layers):
        L.append((f"density_mlp.density_layers.{2 * i}", hidden, d_in if i == 0 else hidden))
    L += [("density_mlp.density_head", 1, hidden), ("density_mlp.feature_head", hidden, hidden)]
    ch = hidden // 2
    L += [("color_mlp.color_layers.0", ch, hidden + de), ("color_mlp.color_layers.2", ch // 2, ch),
          ("color_mlp.color_layers.4", 3, ch // 2)]
    return L


def make_weights(variant: str, seed: int = 0, scene: str = "fog", **kw) -> Dict[str, torch.Tensor]:
    """Deterministic weights for a variant: nn.Linear default-init ranges for every layer, then
    rescaled so that the synthetic scene is not degenerate (raw default init shrinks the signal by
    ~sqrt(6) per layer: after 8 layers the outputs are constants, sigma <= 0 almost everywhere, a
    black frame):

    both scenes: every hidden Linear weight x sqrt(6) (variance-preserving for U(-1/sqrt(n),1/sqrt(n))
                 weights behind a ReLU), so the radiance field really varies with position at all
                 encoding frequencies; colour head weight x6 (colours spread over (0,1) instead of hugging 0.5);
    scene 'fog'  : density head bias +0.25 -> thin participating medium, no ray saturates
                   (the ERT-off roofline case);
    scene 'solid': density head weight x40 and bias +1.5 -> a sizeable share of rays saturate
                   early, but at unrelated depths from ray to ray (a random field).
    scene 'smooth': 'solid' with the first layer blind to encoding frequencies >= 2^2 -> a low-frequency field
                   with large coherent opaque regions, where neighbouring rays terminate together: the
                   early-ray-termination case (wave-level skipping only pays on coherent scenes).
    """
    p = {}
    for i, (name, o, n_in) in enumerate(layer_shapes(variant, **kw)):
        w, b = _linear_init(seed * 1000 + i, o, n_in)
        if o >= 32:
            w = w * math.sqrt(6.0)
        p[name + ".weight"], p[name + ".bias"] = w, b
    dens = "sigma_out" if variant == "v1" else "density_mlp.density_head"
    col = "rgb_out" if variant == "v1" else "color_mlp.color_layers.4"
    p[col + ".weight"] = p[col + ".weight"] * 6.0
    if scene in ("solid", "smooth"):
        p[dens + ".weight"] = p[dens + ".weight"] * 40.0
        p[dens + ".bias"] = p[dens + ".bias"] + 1.5
        if scene == "smooth":
            p[dens + ".bias"] = p[dens + ".bias"] + 5.0             # opaque: rays die after ~37 of 64 samples on average
            first = {"v1": "layers.0", "v2": "density_mlp.density_layers.0", "v3": "dino_fusion.fusion.0"}[variant]
            w = p[first + ".weight"].clone()
            w[:, 3 + 6 * 2: 3 + 6 * (12 if variant == "v3" else 10)] = 0.0        # sin/cos columns of frequencies 2^2 .. 2^(L-1)
            p[first + ".weight"] = w
    elif scene == "fog":
        p[dens + ".bias"] = p[dens + ".bias"] + 0.25
    else:
        raise ValueError(scene)
    return p


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    """-10 log10(mse), data range 1 (src/training/train_multiscale.py:294-295)."""
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float("inf") if mse == 0 else -10.0 * math.log10(mse)


# --------------------------------------------------------------------------
# f1  training path: gradients of the V1 MLP as the HIP kernels round them
# --------------------------------------------------------------------------

def quantize(x: torch.Tensor, mode: str) -> torch.Tensor:
    """Round to the MFMA operand type of `mode` ('bf16' | 'f16' | 'f32') and come back to fp32."""
    if mode == "bf16":
        return x.to(torch.bfloat16).to(torch.float32)
    if mode == "f16":
        return x.to(torch.float16).to(torch.float32)
    return x.to(torch.float32)


def mlp_v1_train_emulated(p: Dict[str, torch.Tensor], x_enc: torch.Tensor, g_out: torch.Tensor, mode: str = "f32"):
    """Forward + backward of src/models/nerf_model.py:16-24 with every MFMA operand (weights, activations,
    masked gradients) rounded to `mode`'s operand type and fp32 accumulation -- the arithmetic of
    csrc/train_impl.hpp.  With mode='f32' this IS autograd's result up to summation order.

    Returns (out4, grads {name: tensor}, acts [h_0..h_n], dzs [dz_1..dz_n, dz_head])."""
    q = lambda t: quantize(t, mode)
    n = 0
    while f"layers.{n}.weight" in p:
        n += 1
    h = [q(x_enc.to(torch.float32))]
    for i in range(n):
        h.append(q(F.relu(h[-1] @ q(p[f"layers.{i}.weight"]).T + p[f"layers.{i}.bias"])))
    sigma = h[-1] @ q(p["sigma_out.weight"]).T + p["sigma_out.bias"]
    rgb = torch.sigmoid(h[-1] @ q(p["rgb_out.weight"]).T + p["rgb_out.bias"])
    out = torch.cat([rgb, sigma], -1)
    g_out = g_out.to(torch.float32)
    d_head = q(torch.cat([g_out[:, :3] * rgb * (1.0 - rgb), g_out[:, 3:4]], -1))        # [d rgb logits, d sigma]
    grads = {
        "rgb_out.weight": d_head[:, :3].T @ h[-1], "rgb_out.bias": d_head[:, :3].sum(0),
        "sigma_out.weight": d_head[:, 3:4].T @ h[-1], "sigma_out.bias": d_head[:, 3:4].sum(0),
    }
    dh = d_head[:, :3] @ q(p["rgb_out.weight"]) + d_head[:, 3:4] @ q(p["sigma_out.weight"])
    dzs = []
    for i in range(n - 1, -1, -1):
        dz = q(dh * (h[i + 1] > 0).to(torch.float32))
        dzs.append(dz)
        grads[f"layers.{i}.weight"] = dz.T @ h[i]
        grads[f"layers.{i}.bias"] = dz.sum(0)
        if i > 0:
            dh = dz @ q(p[f"layers.{i}.weight"])
    return out, grads, h, dzs[::-1] + [d_head]


def relu_margin(p: Dict[str, torch.Tensor], variant: str, x, directions=None, dino=None) -> torch.Tensor:
    """Per sample, the smallest |pre-activation| over every ReLU of the network ('v1': x = encoded points; 'v2' / 'v3':
    x = positions; 'v3' also takes the per-sample DINO features).  A sample whose margin is within summation-order noise of 0 has an ill-defined ReLU mask: its
    gradient legitimately differs between any two correct implementations (tests exclude such samples)."""
    def stack(prefix, h, step):
        m = torch.full((h.shape[0],), float("inf"))
        i = 0
        while f"{prefix}{i}.weight" in p:
            z = _lin(p, f"{prefix}{i}", h)
            m = torch.minimum(m, z.abs().amin(-1))
            h = F.relu(z)
            i += step
        return h, m
    if variant == "v1":
        return stack("layers.", x, 1)[1]
    if variant == "v3":                                            # lora_dino.py:171-193: every ReLU of both fusion passes and the gate
        pe = positional_encoding(x, 12)
        m0 = torch.full((pe.shape[0],), float("inf"))

        def fusion(inp, m):
            z0 = _lin(p, "dino_fusion.fusion.0", inp)
            z1 = _lin(p, "dino_fusion.fusion.2", F.relu(z0))
            return F.relu(z1), torch.minimum(m, torch.minimum(z0.abs().amin(-1), z1.abs().amin(-1)))
        fused, m0 = fusion(torch.cat([pe, dino], -1), m0)
        za = _lin(p, "dino_fusion.attention.0", fused)
        m0 = torch.minimum(m0, za.abs().amin(-1))
        w = torch.softmax(_lin(p, "dino_fusion.attention.2", F.relu(za)), -1)
        final, m0 = fusion(torch.cat([pe * w[:, 0:1], dino * w[:, 1:2]], -1), m0)
        h, m = stack("density_mlp.density_layers.", _lin(p, "dino_fusion.output_proj", final), 2)
        m = torch.minimum(m, m0)
    else:
        pe = positional_encoding(x, 10)
        h, m = stack("density_mlp.density_layers.", pe, 2)
    dens = _lin(p, "density_mlp.density_head", h)
    feat = _lin(p, "density_mlp.feature_head", h)
    m = torch.minimum(m, dens.abs().amin(-1))
    c = torch.cat([feat, positional_encoding(directions, 4)], -1)
    for i in (0, 2):
        z = _lin(p, f"color_mlp.color_layers.{i}", c)
        m = torch.minimum(m, z.abs().amin(-1))
        c = F.relu(z)
    return m
