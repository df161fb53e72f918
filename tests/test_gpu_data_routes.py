"""GPU (-m gpu): the callers on either side of the hot path with DATA-SHAPED inputs.

* f1: `hbr_amd.train_hash2.main` on scenes written to disk in the reference's two formats - Blender (`data/lego/`,
  dataset.py) and colmap2nerf (`--data_path`, dataset_new.py): PNG decode -> all-rays materialisation on the GPU ->
  bounding box -> HashNeRFTrainer -> test render + reference-format checkpoints (train_hash2.py:50-133,193-306).
* f2: `python -m hbr_amd.nerf2mesh` - bounds .npy + checkpoints in, `density_grid_w_rgb.npy` out (nerf2mesh.py:26-88).
No dataset ships with the reference and cv2/torchvision are absent, so the scenes are rendered on the fly from the
package's analytic solid; what is pinned is the file formats, the ray geometry and the field query - not lego's pixels.
"""
import os

import numpy as np
import pytest
import torch

import ref_cpu
from conftest import write_nerf_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("precision", ["fp32", "bf16"])   # bf16: the BASELINE dtype (bf16 MLP, bf16 feature buffers)
@pytest.mark.parametrize("fmt", ["blender", "colmap"])
def test_train_script_on_a_scene_on_disk(tmp_path, monkeypatch, fmt, precision):
    from hbr_amd import train_hash2
    monkeypatch.chdir(tmp_path)
    root = os.path.join("data", "lego") if fmt == "blender" else "scene"   # train_hash2.py:51: default path = Blender lego
    views = write_nerf_scene(root, fmt, n_views=6, H=32, W=32, seed=3)
    argv = ["--num_batch", "1024", "--num_samples", "64", "--hash_size", "12", "--num_epochs", "40", "--steps", "160",
            "--write", "--model_name", "m", "--out_dir", str(tmp_path / "res"), "--precision", precision]
    if fmt == "colmap":
        argv += ["--data_path", root + "/"]
    r = train_hash2.main(argv)
    # 6 x 32 x 32 = 6144 rays -> 6 batches of 1024 per epoch; the run stops at --steps
    assert r["steps"] == 160 and np.isfinite(r["loss"])
    # the scene is learnable from these views: PSNR of the held-out-pose render well above the untrained ~8 dB
    assert r["psnr"] > 14.0, r
    for f in ("m_Nerf_hash.pth", "m_encoder_hash.pth", "bounds_model.npy"):
        assert (tmp_path / f).exists(), f
    sd = torch.load(str(tmp_path / "m_Nerf_hash.pth"), weights_only=True)
    assert list(sd)[0] == "module.sig_model.0.weight"            # the reference's DataParallel-prefixed keys
    b = np.load(str(tmp_path / "bounds_model.npy"))
    assert b.shape == (2, 3) and (b[1] > b[0]).all()
    pngs = [p for p in os.listdir(tmp_path / "res") if p.endswith(".png")]
    assert pngs, "no test render written"


def test_nerf2mesh_entry_point(tmp_path, monkeypatch):
    """Train a few steps, then run the mesh-query CLI on the files the trainer wrote; its .npy must equal an in-process
    query of the same checkpoint, and the oracle's encoder + MLP on a sample of the lattice."""
    from hbr_amd import checkpoint, nerf2mesh, train_hash2
    from hbr_amd.grid_query import grid_coordinates
    from hbr_amd.trainer import build_default_model
    monkeypatch.chdir(tmp_path)
    train_hash2.main(["--synthetic", "8192", "--num_batch", "2048", "--num_samples", "32", "--hash_size", "12", "--num_epochs", "2",
                      "--steps", "8", "--write", "--model_name", "ck", "--out_dir", str(tmp_path / "res")])
    res = 20
    out = str(tmp_path / "density_grid_w_rgb.npy")
    r = nerf2mesh.main(["--bound_pth", "bounds_model.npy", "--ckpt_name", "ck", "--hash_size", "12", "--resolution", str(res),
                        "--batch", "3000", "--out", out])
    grid = np.load(out)
    assert grid.shape == (res, res, res, 4) and grid.dtype == np.float32 and r["resolution"] == res
    # oracle on the same checkpoint and lattice
    mn, mx, mu, sigma = checkpoint.load_bounds("bounds_model.npy")
    enc, _, mlp = build_default_model(mu, sigma, "cpu", T=2 ** 12)
    checkpoint.load_checkpoint("ck", torch.nn.DataParallel(mlp) if False else mlp, enc)
    pts = grid_coordinates(mn, mx, res, "cpu")
    sel = torch.arange(0, res ** 3, 7)
    tabs = [lv.weight.detach() for lv in enc.Embedding_list]
    prm = {f"{s}.{i}.{k}": getattr(getattr(mlp, s)[i], k).detach() for s in ("sig_model", "col_model") for i in (0, 2, 4)
           for k in ("weight", "bias")}
    sc = ref_cpu.level_scales(16, 2048.0, 16)
    feat = ref_cpu.hash_encode(pts[sel], tabs, sc, mu, sigma)
    pe = ref_cpu.dir_encode(torch.tensor([[0.0, 0.0, 1.0]]), 4).half().float().expand(sel.numel(), 24)
    want = ref_cpu.mlp_forward(feat, pe, prm).numpy()
    assert np.allclose(grid.reshape(-1, 4)[sel.numpy()], want, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("res", [256, 512])
def test_grid_query_256_and_512_cubed(res):
    """BASELINE config 5's query at the reference's default resolution (256, nerf2mesh.py:27) and at the config's stated
    512^3 = 1.34e8 lattice points (2.1 GB of output) through K1 + K3 (nerf2mesh.py:26-88): shape, finiteness, the oracle
    on a 4096-point subsample of the lattice, and the time of a second (warm) call."""
    import time
    from hbr_amd import synthetic
    from hbr_amd.grid_query import grid_coordinates, query_density_grid
    from hbr_amd.trainer import HashNeRFTrainer, build_default_model
    o, d, dn, gt = (a.to(DEV) for a in synthetic.scene_rays(4096, seed=5))
    mn, mx, sig = synthetic.ray_bbox(o.cpu(), d.cpu())
    enc, denc, mlp = build_default_model(mn, sig, DEV, T=2 ** 14, seed=1)
    tr = HashNeRFTrainer(enc, mlp, num_samples=64, total_steps=200)
    for _ in range(60):   # a field that is not the untrained constant
        tr.step(o, d, dn.reshape(-1), gt)
    lo, hi = torch.tensor([-1.2, -1.2, -1.2]), torch.tensor([1.2, 1.2, 1.2])
    grid = query_density_grid(enc, mlp, lo, hi, res)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    grid2 = query_density_grid(enc, mlp, lo, hi, res)
    torch.cuda.synchronize()
    warm = time.perf_counter() - t0
    print(f"grid query {res}^3 = {res ** 3:.3e} points: warm call {warm * 1e3:.1f} ms ({res ** 3 / warm:.3e} points/s)")
    assert grid.shape == (res, res, res, 4) and grid.dtype == torch.float32
    assert bool(torch.isfinite(grid).all()) and torch.equal(grid, grid2)
    del grid2
    assert float(grid[..., 3].max()) > 1.0 and float((grid[..., 3] > 0.5).float().mean()) > 1e-3   # the solid is there
    # tens (256^3) to a few hundred (512^3) milliseconds of kernels; a generous bound against a silent slow path
    assert warm < (2.0 if res == 256 else 8.0), warm
    # oracle on a subsample of the lattice (same fp16-rounded coordinates, direction (0, 0, 1) as nerf2mesh.py:69-70);
    # the stride is odd-ish so that the sample walks all three axes
    sel = torch.arange(0, res ** 3, res ** 3 // 4096 + 1)
    pts = grid_coordinates(lo, hi, res, "cpu", index=sel)
    tabs = [lv.weight.detach().cpu() for lv in enc.Embedding_list]
    prm = {f"{s}.{i}.{k}": getattr(getattr(mlp, s)[i], k).detach().cpu() for s in ("sig_model", "col_model") for i in (0, 2, 4)
           for k in ("weight", "bias")}
    feat = ref_cpu.hash_encode(pts, tabs, ref_cpu.level_scales(16, 2048.0, 16), mn, sig)
    pe = ref_cpu.dir_encode(torch.tensor([[0.0, 0.0, 1.0]]), 4).half().float().expand(sel.numel(), 24)
    want = ref_cpu.mlp_forward(feat, pe, prm).numpy()
    got = grid.reshape(-1, 4)[sel.to(DEV)].cpu().numpy()
    assert np.allclose(got, want, rtol=1e-4, atol=2e-5), float(np.abs(got - want).max())
