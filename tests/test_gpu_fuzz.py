"""GPU (-m gpu): randomised shapes through the whole drop-in step and the fused trainer step, against the CPU oracle.

Twelve seeded configurations: R in 1..700 rays, S in 1..200 samples (incl. S = 1, S not a multiple of 64, more than one
64-sample chunk), T a power of two or not, dir_norm a tensor or the scalar default, an occupancy grid with holes or
without, fp32 throughout (the oracle's precision).  For each: vol_render's colours, loss and every gradient against
ref_cpu (vol_renderer.py:141-245 + train_hash2.py:221,226), then ONE fused HashNeRFTrainer.step on the same inputs must
leave the gradient buffer the drop-in backward produced.  What this guards: the glue between the kernels - prologue,
image re-use, planar buffers, the vector / generic compositing kernels, gradient views - at sizes no golden covers."""
import numpy as np
import pytest
import torch

import ref_cpu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _config(i):
    rng = np.random.default_rng(1000 + i)
    R = int(rng.choice([1, 3, 17, 64, 129, 300, 700]))
    S = int(rng.choice([1, 2, 5, 31, 64, 65, 128, 200]))
    T = int(rng.choice([2 ** 8, 2 ** 10, 1000, 2 ** 12, 5000]))
    return rng, R, S, T, bool(rng.integers(0, 2)), bool(rng.integers(0, 2))


@pytest.mark.parametrize("i", range(12))
def test_random_shape_step_vs_oracle(i):
    from hbr_amd._lib import F32
    from hbr_amd.encoder import PositionalEncoder
    from hbr_amd.hash_encoding import HashEncoder
    from hbr_amd.test_hash import MLP_3D
    from hbr_amd.trainer import HashNeRFTrainer
    from hbr_amd.vol_renderer import Volume_Renderer
    rng, R, S, T, scalar_norm, holes = _config(i)
    L = 16
    o, d, dn, gt = ref_cpu.synthetic_rays(R, seed=500 + i)
    mn, mx, sig = ref_cpu.bbox_mu_sigma(o, d)
    tables = torch.from_numpy(rng.uniform(-0.5, 0.5, (L, T, 2)).astype(np.float32))
    params = ref_cpu.mlp_init(700 + i)
    t = ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.from_numpy(rng.uniform(0, 1, S).astype(np.float32)))
    dnorm = 1 if scalar_norm else dn
    grid = ref_cpu.block_pattern_grid(256) if holes else torch.ones((256, 256, 256), dtype=torch.bool)
    sc = ref_cpu.level_scales(16, 2048.0, L)
    # ---- oracle
    tabs = [tables[l].clone().requires_grad_(True) for l in range(L)]
    pr = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    if holes:
        Cr_ref, _, _, mask = ref_cpu.render_masked(o, d, t, dnorm, tabs, sc, mn, sig, pr, grid, mn, sig)
    else:
        Cr_ref, _, _ = ref_cpu.render(o, d, t, dnorm, tabs, sc, mn, sig, pr)
    loss_ref = ref_cpu.train_loss(Cr_ref, gt)
    loss_ref.backward()
    g_tab_ref = torch.stack([x.grad if x.grad is not None else torch.zeros_like(x) for x in tabs])
    # ---- drop-in classes
    enc = HashEncoder(N_max=2048.0, N_min=16, L=L, T=T, F=2, dim=3, mu=mn.to(DEV), sigma=sig.to(DEV), device=DEV)
    mlp = MLP_3D(num_sig=2, num_col=2, L=L, F=2, d_view=24).to(DEV)
    with torch.no_grad():
        for l in range(L):
            enc.Embedding_list[l].weight.copy_(tables[l])
        for k, v in params.items():
            seq, idx, kind = k.split(".")
            getattr(getattr(mlp, seq)[int(idx)], kind).copy_(v)
    vr = Volume_Renderer(H=8, W=8, K=torch.eye(3), near=2.0, far=6.0, device=DEV, Pos_encode=enc, Dir_encode=PositionalEncoder(3, 4),
                         sigma_val=sig, mu=mn)
    if holes:
        vr.bool_grid[...] = grid.to(DEV)
    dn_dev = 1 if scalar_norm else dn.to(DEV)
    Cr, Cf, _ = vr.vol_render(mlp, d.to(DEV), o.to(DEV), num_samples=S, t=t.to(DEV), update_mask=False, dir_norm=dn_dev, hierarchical=False)
    crit = torch.nn.MSELoss()
    loss = crit(Cr, gt.to(DEV)) + crit(Cf, gt.to(DEV))
    loss.backward()
    tag = f"config {i}: R={R} S={S} T={T} scalar_norm={scalar_norm} holes={holes}"
    assert np.allclose(Cr.detach().cpu().numpy(), Cr_ref.detach().numpy(), rtol=1e-4, atol=2e-5), tag
    assert abs(float(loss) - float(loss_ref)) <= 1e-4 * abs(float(loss_ref)) + 1e-7, tag
    g_tab = torch.stack([lv.weight.grad if lv.weight.grad is not None else torch.zeros_like(lv.weight) for lv in enc.Embedding_list]).cpu()
    gs = float(g_tab_ref.abs().max())
    assert float((g_tab - g_tab_ref).abs().max()) <= 1e-3 * gs + 1e-12, tag
    for k, v in pr.items():
        seq, idx, kind = k.split(".")
        got = getattr(getattr(mlp, seq)[int(idx)], kind).grad.cpu()
        want = v.grad if v.grad is not None else torch.zeros_like(v)
        assert float((got - want).abs().max()) <= 1e-3 * float(want.abs().max()) + 1e-4 * gs + 1e-9, (tag, k)
    # ---- one fused trainer step on the same inputs: its gradient buffer == the drop-in backward's gradients
    if not holes and not scalar_norm:  # (the trainer's step takes the all-true grid and a per-ray dir_norm)
        flat_grads = torch.cat([g_tab.reshape(-1).to(DEV)] + [p.grad.reshape(-1) for p in mlp._ordered()])
        tr = HashNeRFTrainer(enc, mlp, num_samples=S, total_steps=10, precision=F32)
        l2 = tr.step(o.to(DEV), d.to(DEV), dn.to(DEV), gt.to(DEV), t=t.to(DEV))
        assert abs(float(l2) - float(loss)) <= 2e-6 * abs(float(loss)) + 1e-9, tag
        n = flat_grads.numel()
        assert float((tr.grad[:n] - flat_grads).abs().max()) <= 2e-4 * float(flat_grads.abs().max()) + 1e-12, tag
