"""CPU: dataset ingest (f1) on a tiny synthetic scene written in both on-disk formats."""
import json
import math
import os

import numpy as np
import torch

import ref_cpu


def _write_scene(root, n=3, H=6, W=8, rgba=True):
    from PIL import Image
    os.makedirs(os.path.join(root, "train"), exist_ok=True)
    rng = np.random.default_rng(0)
    frames, imgs = [], []
    for i in range(n):
        a = rng.integers(0, 256, (H, W, 4 if rgba else 3), dtype=np.uint8)
        Image.fromarray(a, "RGBA" if rgba else "RGB").save(os.path.join(root, "train", f"r_{i}.png"))
        imgs.append(a[..., :3])
        c2w = np.eye(4); c2w[:3, 3] = rng.uniform(-3, 3, 3)
        frames.append({"file_path": f"./train/r_{i}", "rotation": 0.01 * i, "transform_matrix": c2w.tolist()})
    with open(os.path.join(root, "transforms_train.json"), "w") as f:
        json.dump({"camera_angle_x": 0.6911, "frames": frames}, f)
    frames2 = [dict(fr, file_path=fr["file_path"] + ".png", sharpness=1.0) for fr in frames]
    with open(os.path.join(root, "transforms_new.json"), "w") as f:
        json.dump({"camera_angle_x": 0.6911, "fl_x": 9.5, "fl_y": 9.25, "cx": 4.0, "cy": 3.0, "w": W, "h": H, "frames": frames2}, f)
    return imgs, frames


def test_blender_and_colmap_formats(tmp_path):
    from hbr_amd.dataset import NeRF_DATA, NeRF_DATA_NEW, intrinsics, materialise_rays
    root = str(tmp_path)
    imgs, frames = _write_scene(root)
    ds = NeRF_DATA(json_path=os.path.join(root, "transforms_train.json"))
    assert len(ds) == 3 and (ds.H, ds.W) == (6, 8)
    assert abs(float(ds.focal1) - 8 / (2 * math.tan(0.6911 / 2))) < 1e-5 and ds.cx == 4.0 and ds.cy == 3.0
    img, c2w, rot = ds[1]
    assert img.shape == (3, 6, 8) and img.dtype == torch.float32
    assert torch.equal(img, torch.from_numpy(imgs[1]).permute(2, 0, 1).float() / 255)  # alpha dropped, RGB order, /255
    assert torch.allclose(c2w, torch.tensor(frames[1]["transform_matrix"], dtype=torch.float32)) and rot == 0.01
    ds2 = NeRF_DATA_NEW(json_path=os.path.join(root, "transforms_new.json"))
    assert (ds2.H, ds2.W, ds2.focal1, ds2.focal2, ds2.cx, ds2.cy) == (6, 8, 9.5, 9.25, 4.0, 3.0)
    assert torch.equal(ds2[2][0], ds[2][0]) and ds2[2][2] == 1.0
    # K as train_hash2.py:67-72 (integer matrix: the focal length is truncated)
    K = intrinsics(ds2)
    assert K.dtype == torch.int64 and K[0, 0] == 9 and K[1, 1] == 9 and K[0, 2] == 4
    # all-rays materialisation == per-image get_od of the oracle, ground truth in the same pixel order
    o, d, nrm, gt = materialise_rays(ds2, K, "cpu", images_per_batch=2)
    assert o.shape == (3 * 48, 3) and gt.shape == (3 * 48, 3)
    c = torch.stack([ds2[i][1] for i in range(3)])
    o_ref, d_ref, n_ref = ref_cpu.get_od(6, 8, K, c)
    assert torch.allclose(o, o_ref.reshape(-1, 3)) and torch.allclose(d, d_ref.reshape(-1, 3), atol=1e-6)
    assert torch.allclose(nrm, n_ref.reshape(-1, 1), rtol=1e-6)
    assert torch.equal(gt[48:96], ds2[1][0].permute(1, 2, 0).reshape(-1, 3))


def test_grid_lattice_equals_the_reference_recipe():
    """f2: rows of the query lattice built from the flat index on the device == the reference's host recipe
    (nerf2mesh.py:30-40): float64 np.linspace per axis, np.meshgrid (default 'xy' indexing), stack, cast to float16."""
    from hbr_amd.grid_query import grid_coordinates
    rng = np.random.default_rng(5)
    for res in (1, 2, 7, 33):
        mn, mx = rng.uniform(-5, -1, 3), rng.uniform(1, 6, 3)
        x, y, z = (np.linspace(mn[a], mx[a], res) for a in range(3))
        X, Y, Z = np.meshgrid(x, y, z)
        want = torch.stack([torch.tensor(X.reshape(-1)), torch.tensor(Y.reshape(-1)), torch.tensor(Z.reshape(-1))], dim=1).to(torch.float16)
        got = grid_coordinates(mn, mx, res, "cpu")
        assert torch.equal(got, want.float())
        if res == 33:  # a window of rows, as the batched query asks for them
            assert torch.equal(grid_coordinates(mn, mx, res, "cpu", 1000, 1500), want.float()[1000:1500])


def test_scene_writer_round_trips_through_both_loaders(tmp_path):
    from conftest import write_nerf_scene
    from hbr_amd.dataset import NeRF_DATA, NeRF_DATA_NEW
    vb = write_nerf_scene(str(tmp_path / "b"), "blender", n_views=2, H=8, W=10)
    vc = write_nerf_scene(str(tmp_path / "c"), "colmap", n_views=2, H=8, W=10)
    b = NeRF_DATA(json_path=str(tmp_path / "b" / "transforms_train.json"))
    c = NeRF_DATA_NEW(json_path=str(tmp_path / "c" / "transforms_train.json"))
    assert (b.H, b.W) == (8, 10) and (c.H, c.W) == (8, 10) and len(b) == len(c) == 2
    assert torch.equal(b[1][0], torch.from_numpy(vb[1][0]).permute(2, 0, 1).float() / 255)
    assert torch.equal(c[0][0], torch.from_numpy(vc[0][0]).permute(2, 0, 1).float() / 255)
    assert abs(float(b.focal1) - float(c.focal1)) < 1e-4
