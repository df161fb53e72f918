"""GPU (-m gpu): SURVEY 8 row f4 - the hierarchical pass's resampler and the occupancy-grid update as kernels.

* hbr_hierarchical_resample == the reference's hierarchical_sampling recorded in G12 (hs_weights, hs_u, hs_samples01 ->
  hs_t, hs_rays; helper.py:23-51), exact; == the oracle at other shapes (S != n, unsorted z_vals, per-ray z_vals);
  device-side draws: reproducible, sorted, the right multiset
* hbr_occupancy_update == Volume_Renderer.update_grid called on the reference's object (G16; vol_renderer.py:116-131):
  three calls in a row incl. int8 wrap-around, the last-point-decides rule and the nothing-set => all-True branch;
  also through the drop-in Volume_Renderer.update_grid, and with ray-generated points
"""
import numpy as np
import pytest
import torch

import ref_cpu
from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_resample_vs_reference_golden():
    from hbr_amd import helper, ops
    g = load_golden("g12_hierarchical.npz")
    g8 = load_golden("g8_render_step.npz")
    w, u, s01 = T_(g["hs_weights"]), T_(g["hs_u"]), T_(g["hs_samples01"])
    R, S = u.shape
    t = T_(g8["t"])
    tf = ops.hierarchical_resample(w, t, S, 2.0, 6.0, u=u, samples01=s01)
    assert tf.shape == (R, 2 * S)
    assert np.array_equal(tf.cpu().numpy(), g["hs_t"])  # values are copies of z_vals / samples entries: exact
    rays, comb = helper.hierarchical_sampling(T_(g8["o"]), T_(g8["d"]), z_vals=t, weights=w, n_samples=S, tn=2.0, tf=6.0, u=u, samples01=s01)
    assert torch.equal(comb, tf) and np.allclose(rays.cpu().numpy(), g["hs_rays"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("R,S,n,mode", [(257, 128, 128, "shared"), (33, 64, 40, "shared"), (19, 48, 100, "per_ray"), (7, 37, 37, "unsorted"), (3, 300, 300, "shared")])
def test_resample_vs_oracle(R, S, n, mode):
    from hbr_amd import ops
    rng = np.random.default_rng(R * 1000 + S)
    w = torch.from_numpy(rng.normal(0.2, 0.5, (R, S)).astype(np.float32))
    w[:, ::7] = 0
    u = torch.from_numpy(rng.uniform(0, 1, (R, S)).astype(np.float32))
    s01 = torch.from_numpy(rng.uniform(0, 1, n).astype(np.float32))
    if mode == "shared":
        z = ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.from_numpy(rng.uniform(0, 1, S).astype(np.float32)))
    elif mode == "per_ray":
        z = torch.sort(torch.from_numpy(rng.uniform(2, 6, (R, S)).astype(np.float32)), dim=-1).values
    else:
        z = torch.from_numpy(rng.uniform(2, 6, S).astype(np.float32))
    # oracle: the reference's op sequence with the draws explicit (n_samples may differ from S: the clamp is to n - 1)
    ww = w.clone(); ww[ww < 0] = 0
    pdf = (ww + 1e-5) / torch.sum(ww + 1e-5, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    inds = torch.searchsorted(cdf, u.contiguous(), right=True).clamp(0, n - 1)
    smp = (s01 * (6.0 - 2.0) + 2.0)[inds]
    want, _ = torch.sort(torch.cat([z.expand(R, S) if z.dim() == 1 else z, smp], dim=-1), dim=-1)
    got = ops.hierarchical_resample(w.to(DEV), z.to(DEV), n, 2.0, 6.0, u=u.to(DEV), samples01=s01.to(DEV)).cpu()
    # torch.sum's association differs from the kernel's sequential sum: a draw within an ulp of a cdf value may pick the
    # neighbouring sample - allow a handful of rays to differ, the rest must be exact
    bad = (got != want).any(dim=-1)
    assert int(bad.sum()) <= max(1, R // 100), int(bad.sum())
    assert torch.equal(torch.sort(got, dim=-1).values, got)


def test_resample_device_draws():
    from hbr_amd import ops
    R, S = 512, 64
    w = torch.rand(R, S, device=DEV)
    z = torch.linspace(2, 6, S, device=DEV)
    a = ops.hierarchical_resample(w, z, S, 2.0, 6.0, seed=7, offset=100)
    b = ops.hierarchical_resample(w, z, S, 2.0, 6.0, seed=7, offset=100)
    c = ops.hierarchical_resample(w, z, S, 2.0, 6.0, seed=7, offset=104)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.equal(torch.sort(a, dim=-1).values, a)
    # every ray keeps its S first-pass depths and adds S values from ONE shared vector of S samples in [2, 6)
    new = []
    for r in range(4):
        row = a[r].tolist()
        for v in z.tolist():
            row.remove(v)
        new.append(row)
    pool = set(v for row in new for v in row)
    assert len(pool) <= S and all(2.0 <= v < 6.0 for v in pool)
    from hbr_amd import helper
    torch.manual_seed(3)
    r1 = helper.hierarchical_sampling(torch.zeros(R, 3, device=DEV), torch.ones(R, 3, device=DEV), z, w, S, 2.0, 6.0)[1]
    torch.manual_seed(3)
    r2 = helper.hierarchical_sampling(torch.zeros(R, 3, device=DEV), torch.ones(R, 3, device=DEV), z, w, S, 2.0, 6.0)[1]
    assert torch.equal(r1, r2)


def test_occupancy_update_vs_reference_golden():
    from hbr_amd import ops
    from hbr_amd.vol_renderer import Volume_Renderer
    g = load_golden("g16_update_grid.npz")
    G = int(g["grid_size"])
    mu, sv = g["mu"], float(g["sigma_val"])
    pts = T_(g["points"])
    grid = torch.zeros((G, G, G), dtype=torch.bool, device=DEV)
    tmp = torch.zeros((G, G, G), dtype=torch.int8, device=DEV)
    vr = Volume_Renderer(H=8, W=8, K=torch.eye(3), near=2.0, far=6.0, device=DEV, max_dim=4 * G, sigma_val=torch.tensor(sv), mu=torch.from_numpy(mu))
    vr.bool_grid[...] = False
    for k in range(3):
        if k == 2:
            grid[...] = False; tmp[...] = 0
            vr.bool_grid[...] = False; vr.tmp_arr[...] = 0
        ops.occupancy_update(grid, mu.tolist(), sv, T_(g[f"alpha{k}"]), x=pts, tmp_arr=tmp)
        assert np.array_equal(grid.cpu().numpy(), g[f"grid{k}"]), k
        assert np.array_equal(tmp.cpu().numpy(), g[f"tmp{k}"]), k
        vr.update_grid(pts, T_(g[f"alpha{k}"]))  # the drop-in method (vol_renderer.py:116)
        assert np.array_equal(vr.bool_grid.cpu().numpy(), g[f"grid{k}"]) and np.array_equal(vr.tmp_arr.cpu().numpy(), g[f"tmp{k}"])
    # ray-generated points == explicit points; no tmp_arr == zeros
    o, d, dn, _ = ref_cpu.synthetic_rays(200, seed=2)
    t = torch.linspace(2.0, 6.0, 24)
    x = (o[:, None, :] + d[:, None, :] * t[None, :, None]).reshape(-1, 3)
    mn, mx, sig = ref_cpu.bbox_mu_sigma(o, d)
    al = torch.from_numpy(np.random.default_rng(0).normal(0.1, 1, x.shape[0]).astype(np.float32))
    ga = torch.zeros((32, 32, 32), dtype=torch.bool, device=DEV)
    gb, gc = ga.clone(), torch.zeros((32, 32, 32), dtype=torch.bool)
    ops.occupancy_update(ga, mn.tolist(), float(sig), al.to(DEV), x=x.to(DEV))
    ops.occupancy_update(gb, mn.tolist(), float(sig), al.to(DEV), rays=(o.to(DEV), d.to(DEV), t.to(DEV)))
    ref_cpu.update_grid(x, al, gc, torch.zeros((32, 32, 32), dtype=torch.int8), mn, sig)
    assert torch.equal(ga, gb) and torch.equal(ga.cpu(), gc) and 0 < int(gc.sum()) < 32 ** 3
    # get_mask sees the updated grid (vol_renderer.py:133-140)
    keep = ops.occupancy_mask(ga, mn.tolist(), float(sig), x=x.to(DEV))
    assert torch.equal(keep.bool().cpu(), ref_cpu.occupancy_mask(x, gc, mn, sig))
