"""GPU (-m gpu): K0 - the stratified depth sampler and the occupancy-grid lookup on the device.

* hbr_strat_sample with the uniform draw made explicit == the reference's strat_sampler output recorded in G7
  (helper.py:234-235), and == the oracle for other S; with the device-side generator: a Philox4x32-10 known-answer,
  range, reproducibility.
* hbr_occupancy_mask == Volume_Renderer.get_mask (vol_renderer.py:133-140) on G13's mixed grid.
* vol_render's masked branch (vol_renderer.py:209-221) fused into the kernel pipeline == the reference's recorded
  render with that grid (G13): sigma/rgb, Cr, loss, every gradient.
"""
import numpy as np
import pytest
import torch

import ref_cpu
from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T_(a, dev=DEV):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_strat_sample_explicit_draw_vs_reference_golden_and_oracle():
    from hbr_amd import ops
    g = load_golden("g7_rays.npz")
    t = ops.strat_sample(2.0, 6.0, 16, DEV, u=T_(g["strat_u"])).cpu().numpy()
    assert np.allclose(t, g["strat_t"], rtol=0, atol=1e-6)  # torch's vectorised linspace may differ in the last bit
    assert t.max() > 6.0                                     # can exceed far
    rng = np.random.default_rng(3)
    for S, tn, tf in ((128, 2.0, 6.0), (1, 2.0, 6.0), (2, 0.5, 9.0), (257, 0.05, 1.0), (4096, 2.0, 6.0)):
        u = torch.from_numpy(rng.uniform(0, 1, S).astype(np.float32))
        want = ref_cpu.strat_jitter_to_t(tn, tf, S, u).numpy()
        got = ops.strat_sample(tn, tf, S, DEV, u=u.to(DEV)).cpu().numpy()
        assert np.allclose(got, want, rtol=0, atol=2e-6 * max(1.0, abs(tf))), S


def test_strat_sample_device_generator():
    from hbr_amd import ops
    S, tn, tf = 128, 2.0, 6.0
    a = ops.strat_sample(tn, tf, S, DEV, seed=1234, offset=7)
    b = ops.strat_sample(tn, tf, S, DEV, seed=1234, offset=7)
    c = ops.strat_sample(tn, tf, S, DEV, seed=1234, offset=8)
    d = ops.strat_sample(tn, tf, S, DEV, seed=1235, offset=7)
    assert torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a, d)
    lin = torch.linspace(tn, tf, S)
    jit = (a.cpu() - lin) * S / (tf - tn)  # the uniform that was drawn, up to rounding
    assert float(jit.min()) > -1e-4 and float(jit.max()) < 1 + 1e-4
    big = (ops.strat_sample(0.0, 1.0, 65536, DEV, seed=5, offset=0).cpu().double() - torch.linspace(0, 1, 65536).double()) * 65536
    assert abs(float(big.mean()) - 0.5) < 0.01 and abs(float(big.var()) - 1 / 12) < 0.005
    # Philox4x32-10 known answer (Random123 kat_vectors: counter 0, key 0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8):
    # sample 0 of (seed 0, offset 0) uses the first word's top 24 bits
    t0 = float(ops.strat_sample(0.0, 1.0, 1, DEV, seed=0, offset=0)[0])
    assert t0 == (0x6627E8D5 >> 8) * 2.0 ** -24


def test_occupancy_mask_vs_oracle():
    from hbr_amd import ops
    g8, g = load_golden("g8_render_step.npz"), load_golden("g13_masked_render.npz")
    grid = ref_cpu.block_pattern_grid(int(g["grid_size"]))
    o, d, t = (T_(g8[k]) for k in ("o", "d", "t"))
    keep = ops.occupancy_mask(grid.to(DEV), g8["mu"].tolist(), float(g8["sigma"]), rays=(o, d, t))
    assert keep.dtype == torch.uint8 and np.array_equal(keep.cpu().numpy().astype(bool), g["mask"])
    # explicit points, including cells that are negative (torch indexing counts them from the end) or out of range
    rng = np.random.default_rng(4)
    pts = rng.uniform(-0.3, 1.2, (5000, 3)).astype(np.float32)
    mu, sig = torch.zeros(3), torch.tensor(1.0)
    cells = (torch.from_numpy(pts) * 256).long()
    ok = ((cells >= -256) & (cells < 256)).all(dim=-1)
    want = torch.zeros(5000, dtype=torch.bool)
    want[ok] = ref_cpu.occupancy_mask(torch.from_numpy(pts)[ok], grid, mu, sig)
    got = ops.occupancy_mask(grid.to(DEV), [0.0, 0.0, 0.0], 1.0, x=T_(pts)).cpu().bool()
    assert torch.equal(got, want) and 0 < int(ok.sum()) < 5000


def _modules_from_g8(g8):
    from hbr_amd.encoder import PositionalEncoder
    from hbr_amd.hash_encoding import HashEncoder
    from hbr_amd.test_hash import MLP_3D
    from hbr_amd.vol_renderer import Volume_Renderer
    L, T = int(g8["L"]), int(g8["T"])
    mu, sigma = T_(g8["mu"]), torch.tensor(float(g8["sigma"]))
    enc = HashEncoder(N_max=2048.0, N_min=16, L=L, T=T, F=2, dim=3, mu=mu, sigma=sigma.to(DEV), device=DEV)
    mlp = MLP_3D(num_sig=2, num_col=2, L=L, F=2, d_view=24).to(DEV)
    with torch.no_grad():
        for l in range(L):
            enc.Embedding_list[l].weight.copy_(T_(g8["tables"][l]))
        for k in [k for k in g8 if k.startswith("p.")]:
            seq, idx, kind = k[2:].split(".")
            getattr(getattr(mlp, seq)[int(idx)], kind).copy_(T_(g8[k]))
    vr = Volume_Renderer(H=8, W=8, K=torch.eye(3), near=2.0, far=6.0, device=DEV, Pos_encode=enc, Dir_encode=PositionalEncoder(3, 4),
                         max_dim=2 ** 10, sigma_val=sigma, mu=mu)
    return enc, mlp, vr


def test_masked_render_vs_reference_golden():
    """Mixed occupancy grid through the FUSED path (the mask is one more kernel input, not a separate eager branch)."""
    g8, g = load_golden("g8_render_step.npz"), load_golden("g13_masked_render.npz")
    enc, mlp, vr = _modules_from_g8(g8)
    o, d, dn, gt, t = (T_(g8[k]) for k in ("o", "d", "dir_norm", "gt", "t"))
    S = t.shape[0]
    # 1) in-place edit of the default grid; 2) rebinding the attribute to a fresh tensor (version 0 again): both must be seen
    for rebind in (False, True):
        for p in list(enc.parameters()) + list(mlp.parameters()):
            p.grad = None
        if rebind:
            vr.bool_grid = torch.ones_like(vr.bool_grid)
            Cr_full, _, _ = vr.vol_render(mlp, d, o, num_samples=S, t=t, dir_norm=dn, hierarchical=False)
            assert np.allclose(Cr_full.detach().cpu().numpy(), g8["Cr"], rtol=1e-4, atol=1e-5)  # all-true again
            vr.bool_grid = ref_cpu.block_pattern_grid(int(g["grid_size"])).to(DEV)
        else:
            vr.bool_grid[...] = ref_cpu.block_pattern_grid(int(g["grid_size"])).to(DEV)
        Cr, Cf, _ = vr.vol_render(mlp, d, o, num_samples=S, t=t, update_mask=False, dir_norm=dn, hierarchical=False)
        assert Cf is Cr
        assert np.allclose(vr.last_sigma.detach().cpu().numpy(), g["sig_out"], rtol=1e-4, atol=1e-5)
        assert np.allclose(vr.last_rgb.detach().cpu().numpy(), g["rgb_out"], rtol=1e-4, atol=1e-5)
        assert float(vr.last_sigma.reshape(-1)[~T_(g["mask"])].abs().max()) == 0.0
        assert np.allclose(Cr.detach().cpu().numpy(), g["Cr"], rtol=1e-4, atol=1e-5)
        loss = torch.nn.functional.mse_loss(Cr, gt) + torch.nn.functional.mse_loss(Cf, gt)
        assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
        loss.backward()
        got = torch.stack([lv.weight.grad for lv in enc.Embedding_list]).cpu().numpy()
        assert np.allclose(got, g["dtables"], rtol=1e-3, atol=1e-5 * np.abs(g["dtables"]).max())
        for name, p in mlp.named_parameters():
            ref = g["g." + name]
            assert np.allclose(p.grad.cpu().numpy(), ref, rtol=1e-3, atol=1e-4 * np.abs(ref).max()), name
    # update_mask=True takes the unmasked branch whatever the grid holds (vol_renderer.py:199-208)
    with torch.no_grad():
        Cu, _, _ = vr.vol_render(mlp, d, o, num_samples=S, t=t, update_mask=True, dir_norm=dn, hierarchical=False)
    assert np.allclose(Cu.cpu().numpy(), g8["Cr"], rtol=1e-4, atol=1e-5)


def test_trainer_step_uses_no_host_side_sampling():
    """HashNeRFTrainer.step draws t[S] on the device from (seed, step): two trainers with the same seed walk the same
    depths; a different seed walks others; and the depths of step k do not depend on how many steps ran before."""
    from hbr_amd import synthetic
    from hbr_amd.trainer import HashNeRFTrainer, build_default_model
    o, d, dn, gt = (a.to(DEV) for a in synthetic.scene_rays(512, seed=71))
    mn, mx, sig = synthetic.ray_bbox(o, d)
    losses = []
    for seed in (11, 11, 12):
        enc, _, mlp = build_default_model(mn, sig, DEV, T=2 ** 12, seed=3)
        tr = HashNeRFTrainer(enc, mlp, num_samples=64, total_steps=100, seed=seed)
        losses.append([float(tr.step(o, d, dn.reshape(-1), gt)) for _ in range(3)])
        assert torch.equal(tr.sample_t(DEV), tr.sample_t(DEV))
    assert losses[0] == losses[1] and losses[0] != losses[2]


def test_render_fwd_one_call_vs_reference_golden():
    """hbr_render_fwd (direction encoding + K1 + K3 + K5 in one library call) against the reference's recorded renders:
    G8 (all cells occupied) and G13 (mixed grid through `keep`), sigma/rgb included; bf16 feature buffers give the
    same colours as fp32 ones in bf16 mode (the MLP rounds its inputs to bf16 either way)."""
    from hbr_amd import ops
    from hbr_amd._lib import BF16, F32
    g8, g = load_golden("g8_render_step.npz"), load_golden("g13_masked_render.npz")
    enc, mlp, vr = _modules_from_g8(g8)
    o, d, dn, t = (T_(g8[k]) for k in ("o", "d", "dir_norm", "t"))
    R, S = o.shape[0], t.shape[0]
    geom, tabs = enc.geometry(), enc.stacked_tables()
    flat, _ = mlp.flat_params()
    Cr, wts, out = ops.render_fwd(geom, tabs, flat, o, d, t, dn, want_wts=True, want_out=True)
    assert np.allclose(Cr.cpu().numpy(), g8["Cr"], rtol=1e-4, atol=1e-5)
    assert np.allclose(out[:, 3].reshape(R, S).cpu().numpy(), g8["sig_out"], rtol=1e-4, atol=1e-5)
    assert np.allclose(out[:, :3].reshape(R, S, 3).cpu().numpy(), g8["rgb_out"], rtol=1e-4, atol=1e-5)
    assert wts.shape == (R, S) and torch.allclose((wts[:, :, None] * out[:, :3].reshape(R, S, 3)).sum(1), Cr, rtol=1e-5, atol=1e-6)
    keep = ops.occupancy_mask(ref_cpu.block_pattern_grid(int(g["grid_size"])).to(DEV), g8["mu"].tolist(), float(g8["sigma"]), rays=(o, d, t))
    Cm, _, _ = ops.render_fwd(geom, tabs, flat, o, d, t, dn, keep=keep)
    assert np.allclose(Cm.cpu().numpy(), g["Cr"], rtol=1e-4, atol=1e-5)
    a, _, _ = ops.render_fwd(geom, tabs, flat, o, d, t, dn, precision=BF16, feat_dtype=F32)
    b, _, _ = ops.render_fwd(geom, tabs, flat, o, d, t, dn, precision=BF16, feat_dtype=BF16)
    assert torch.equal(a, b)
    e, _, _ = ops.render_fwd(geom, tabs, flat, o[:0], d[:0], t)
    assert tuple(e.shape) == (0, 3)
