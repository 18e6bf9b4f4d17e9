"""A procedurally generated Blender-format scene with real 3-D structure (the lego data set is not available offline:
SURVEY.md section 8c/d): three textured Lambertian spheres on a checkered disc, ray-cast analytically (numpy, 3 x 3
supersampling) from cameras on a sphere around the origin -- the camera model of `get_rays` (ray_utils.py:4-37:
dirs = [(x - W/2)/f, -(y - H/2)/f, -1] rotated by c2w) and the on-disk layout `load_blender_data` reads
(data_loader.py:8-64: transforms_{train,test}.json with camera_angle_x + frames[file_path, transform_matrix], PNGs).

The images are the scene's GROUND TRUTH: a field trained on the train split is scored against them (tools/trained_scene.py).
Standalone numpy + PIL: neither the package nor the oracle is imported.

    python tools/synthetic_scene.py OUT_DIR [--size 128] [--train 8] [--test 4]
"""
import argparse
import json
import math
import os

import numpy as np

CAMERA_ANGLE_X = 0.6911112070083618            # lego's
RADIUS = 4.0311                                # lego's camera distance: the scene sits inside [near, far] = [2, 6]

SPHERES = [  # centre, radius, albedo, stripe frequency
    ((0.0, 0.0, 0.05), 0.62, (0.85, 0.25, 0.20), 9.0),
    ((0.78, 0.45, -0.22), 0.36, (0.20, 0.65, 0.30), 14.0),
    ((-0.55, -0.72, -0.28), 0.30, (0.22, 0.35, 0.85), 0.0),
]
DISC_Z, DISC_R = -0.58, 1.45
LIGHT = np.array([0.45, 0.35, 0.82]) / np.linalg.norm([0.45, 0.35, 0.82])
# An opaque, non-black backdrop: the loader drops alpha (data_loader.py:33) and baseline.yaml renders with white_bkgd false, so the
# field itself must emit the backdrop (through the last sample's alpha = 1 step).  On a black backdrop a randomly initialised
# field collapses to zero density within the first epochs (all-black frame, dead ReLU, 12.6 dB: measured) -- a training artefact
# unrelated to the arithmetic this scene is meant to probe.
BACKGROUND = np.array([0.55, 0.70, 0.90])


def look_at_pose(azimuth, elevation, radius=RADIUS):
    """camera-to-world (4,4): camera at (radius, azimuth, elevation) looking at the origin, -z forward, +y up (Blender / NeRF)."""
    p = radius * np.array([math.cos(elevation) * math.cos(azimuth), math.cos(elevation) * math.sin(azimuth), math.sin(elevation)])
    z = p / np.linalg.norm(p)
    x = np.cross([0.0, 0.0, 1.0], z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, y, z, p
    return m.astype(np.float32)


def split_poses(n, phase, elev_lo=0.25, elev_hi=0.95):
    """n cameras spread by the golden angle; `phase` offsets the test split from the train split."""
    out = []
    for i in range(n):
        az = 2.399963 * i + phase
        el = elev_lo + (elev_hi - elev_lo) * ((i * 0.618034 + 0.37 * phase) % 1.0)
        out.append(look_at_pose(az, el))
    return out


def shade(o, d):
    """Nearest hit of rays (o + t d) with the scene -> rgb (n,3); d need not be normalised."""
    n = o.shape[0]
    dn = d / np.linalg.norm(d, axis=-1, keepdims=True)
    best_t = np.full(n, np.inf)
    rgb = np.tile(BACKGROUND, (n, 1))
    for (c, r, albedo, freq) in SPHERES:
        c = np.asarray(c)
        oc = o - c
        b = (oc * dn).sum(-1)
        disc = b * b - ((oc * oc).sum(-1) - r * r)
        hit = disc > 0
        t = -b - np.sqrt(np.where(hit, disc, 0.0))
        hit &= (t > 0) & (t < best_t)
        p = o + dn * t[:, None]
        nrm = (p - c) / r
        lam = 0.30 + 0.70 * np.clip((nrm * LIGHT).sum(-1), 0.0, None)
        tex = np.ones(n)
        if freq > 0:
            tex = 0.72 + 0.28 * np.sign(np.sin(freq * np.arctan2(nrm[:, 1], nrm[:, 0])) * np.sin(freq * 0.5 * np.arcsin(np.clip(nrm[:, 2], -1, 1))))
        col = np.asarray(albedo)[None, :] * (lam * tex)[:, None]
        rgb = np.where(hit[:, None], col, rgb)
        best_t = np.where(hit, t, best_t)
    # checkered disc
    t = (DISC_Z - o[:, 2]) / np.where(np.abs(dn[:, 2]) > 1e-9, dn[:, 2], 1e-9)
    p = o + dn * t[:, None]
    hit = (t > 0) & (t < best_t) & ((p[:, 0] ** 2 + p[:, 1] ** 2) < DISC_R ** 2)
    check = ((np.floor(p[:, 0] * 2.5) + np.floor(p[:, 1] * 2.5)) % 2.0)
    # hard shadow of the spheres on the disc
    lit = np.ones(n)
    for (c, r, _, _) in SPHERES:
        oc = p - np.asarray(c)
        b = (oc * LIGHT).sum(-1)
        disc = b * b - ((oc * oc).sum(-1) - r * r)
        lit = np.where((disc > 0) & (-b + np.sqrt(np.where(disc > 0, disc, 0.0)) > 0), 0.45, lit)
    col = (0.30 + 0.45 * check)[:, None] * np.array([0.9, 0.85, 0.7])[None, :] * (lit * (0.30 + 0.70 * LIGHT[2]))[:, None]
    return np.where(hit[:, None], col, rgb)


def render_view(c2w, size, ss=3):
    """(size,size,3) float image of the scene from `c2w`; ss x ss sub-pixel samples around get_rays' pixel positions."""
    focal = 0.5 * size / math.tan(0.5 * CAMERA_ANGLE_X)
    ys, xs = np.mgrid[0:size, 0:size].astype(np.float64)
    acc = np.zeros((size * size, 3))
    rot, org = c2w[:3, :3].astype(np.float64), c2w[:3, 3].astype(np.float64)
    for sy in range(ss):
        for sx in range(ss):
            ox, oy = (sx + 0.5) / ss - 0.5, (sy + 0.5) / ss - 0.5
            dirs = np.stack([(xs + ox - size * 0.5) / focal, -(ys + oy - size * 0.5) / focal, -np.ones_like(xs)], -1).reshape(-1, 3)
            d = dirs @ rot.T
            acc += shade(np.broadcast_to(org, d.shape), d)
    return (acc / (ss * ss)).reshape(size, size, 3)


def write_scene(root, size=128, n_train=8, n_test=4):
    """Write the data set; returns {'train': [(png, pose)], 'test': [...]}.  8-bit PNGs, like the Blender sets."""
    from PIL import Image
    out = {}
    for split, count, phase in (("train", n_train, 0.0), ("test", n_test, 1.1)):
        os.makedirs(os.path.join(root, split), exist_ok=True)
        frames, recs = [], []
        for i, pose in enumerate(split_poses(count, phase)):
            img = render_view(pose, size)
            path = os.path.join(root, split, f"r_{i}.png")
            Image.fromarray((np.clip(img, 0, 1) * 255 + 0.5).astype(np.uint8), "RGB").save(path)
            frames.append({"file_path": f"./{split}/r_{i}", "transform_matrix": pose.tolist()})
            recs.append((path, pose))
        with open(os.path.join(root, f"transforms_{split}.json"), "w") as f:
            json.dump({"camera_angle_x": CAMERA_ANGLE_X, "frames": frames}, f)
        out[split] = recs
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--train", type=int, default=8)
    ap.add_argument("--test", type=int, default=4)
    a = ap.parse_args()
    write_scene(a.out, a.size, a.train, a.test)
    print(a.out)
