"""pytest configuration: the `gpu` marker + shared paths/fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def write_nerf_scene(root, fmt="blender", n_views=4, H=24, W=24, seed=0, radius=4.03):
    """A tiny multi-view scene on disk in one of the two formats the reference's trainers read (dataset.py /
    dataset_new.py): PNG frames rendered from the package's analytic solid (synthetic.solid_field) by the reference's
    compositing rule, `transforms_train.json` and `transforms_tmp.json` (the test pose file train_hash2.py:56 reads).
    fmt "blender": camera_angle_x, file_path without extension, `rotation`;  "colmap": fl_x/fl_y/cx/cy/w/h,
    file_path with extension, `sharpness`.  Returns the list of (image uint8 [H,W,3], c2w [4,4])."""
    import json
    import math
    import torch
    from PIL import Image
    from hbr_amd import synthetic
    from hbr_amd.helper import get_od
    os.makedirs(os.path.join(root, "train"), exist_ok=True)
    rng = np.random.default_rng(seed)
    angle_x = 0.6911
    focal = W / (2 * math.tan(angle_x / 2))
    K = torch.tensor([[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1]], dtype=torch.float32)
    frames, views = [], []
    t = torch.linspace(2.0, 6.0, 96)
    for i in range(n_views):
        az, pol = rng.uniform(0, 2 * np.pi), rng.uniform(0.3, 1.2)
        eye = radius * np.array([np.cos(az) * np.sin(pol), np.sin(az) * np.sin(pol), np.cos(pol)])
        fwd = -eye / np.linalg.norm(eye)                       # camera looks along -z of its frame
        right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        c2w = np.eye(4); c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, up, -fwd, eye
        o, d, _ = get_od(H, W, K, torch.tensor(c2w, dtype=torch.float32)[None])
        pts = o[0][:, None, :] + d[0][:, None, :] * t[None, :, None]
        sg, rgb = synthetic.solid_field(pts)
        img = (synthetic._composite_uniform(t, rgb, sg).clamp(0, 1).reshape(H, W, 3) * 255).round().byte().numpy()
        Image.fromarray(img, "RGB").save(os.path.join(root, "train", f"r_{i}.png"))
        views.append((img, c2w))
        fr = {"transform_matrix": c2w.tolist()}
        if fmt == "blender":
            fr.update(file_path=f"./train/r_{i}", rotation=0.01 * i)
        else:
            fr.update(file_path=f"./train/r_{i}.png", sharpness=50.0 + i)
        frames.append(fr)
    meta = {"camera_angle_x": angle_x, "frames": frames}
    if fmt != "blender":
        meta.update(fl_x=focal, fl_y=focal, cx=W / 2, cy=H / 2, w=W, h=H)
    for name in ("transforms_train.json", "transforms_tmp.json"):
        with open(os.path.join(root, name), "w") as f:
            json.dump(meta if name == "transforms_train.json" else dict(meta, frames=frames[:1]), f)
    return views
