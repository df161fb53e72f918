"""GPU (-m gpu): "PSNR within 0.1 dB of reference" (BASELINE.json north_star; metric = helper.py:301-304) at CONVERGENCE.

tests/golden/g15_converged_psnr.npz holds the held-out PSNR curves of the REFERENCE's own modules
(Volume_Renderer.vol_render + DataParallel(MLP_3D) + HashEncoder + torch.optim.Adam/AdamW + CosineAnnealingLR, the loop
of train_hash2.py:211-234, fp32 on CPU) trained by oracle/make_psnr_golden.py to the end of a cosine schedule - a
plateau - from five seeded initialisations on a fixed set of 16 x 1024 rays x 64 samples (65 536 points per step: the
LDS scatter kernel, i.e. the shipped path).  Here the SHIPPED bf16 path (HashNeRFTrainer: bf16 MFMA, bf16 feature
buffers, fixed-point scatter, fused Adam) and the DROP-IN route (vol_render under bf16 autocast + autograd +
torch.optim) train from the same initial parameters, rays and per-step jitter to the same horizon and are scored on the
same held-out rays.

Asserted: |mean over seeds of (HIP - reference)| <= 0.1 dB at the horizon for both routes, every seed within 0.5 dB,
and the HIP curves plateau like the reference's (last 20 % of the horizon moves < 0.1 dB).
"""
import os

import numpy as np
import pytest
import torch

import make_psnr_golden as MP
import ref_cpu
from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(seed, steps):
    tables0, u, params0 = MP.seeded_inputs(seed, steps)
    ts = torch.stack([ref_cpu.strat_jitter_to_t(MP.NEAR, MP.FAR, MP.S, torch.from_numpy(u[k])) for k in range(steps)]).to(DEV)
    return tables0, u, params0, ts


def _model(mn, sig, tables0, params0):
    from hbr_amd.trainer import build_default_model
    enc, denc, mlp = build_default_model(mn, sig, DEV, L=MP.L, T=MP.T, seed=0)
    with torch.no_grad():
        for l in range(MP.L):
            enc.Embedding_list[l].weight.copy_(torch.from_numpy(tables0[l]))
        for k, v in params0.items():
            seq, idx, kind = k.split(".")
            getattr(getattr(mlp, seq)[int(idx)], kind).copy_(v)
    return enc, denc, mlp


@pytest.fixture(scope="module")
def world():
    g = load_golden("g15_converged_psnr.npz")
    assert tuple(int(v) for v in g["config"]) == (MP.R, MP.S, MP.L, MP.T, MP.NB, MP.EVAL_RAYS, MP.EVAL_SEED, MP.BATCH_SEED0, MP.BBOX_SEED)
    mn, sig, batches, test = MP.scene()
    # the seeded inputs regenerate to the bytes the reference was trained on (numpy PCG64 streams)
    assert abs(MP.checksum(*[a.numpy() for b in batches[:2] for a in b], *[a.numpy() for a in test]) - float(g["scene_checksum"])) < 1e-6 * float(g["scene_checksum"])
    batches = [tuple(a.to(DEV) for a in b) for b in batches]
    test = tuple(a.to(DEV) for a in test)
    return g, mn, sig, batches, test


def _psnr(C, gt):
    from hbr_amd.helper import calc_psnr
    return float(calc_psnr(C, gt))


def _train_fused(world, seed, steps, eval_steps):
    from hbr_amd._lib import BF16
    from hbr_amd.trainer import HashNeRFTrainer
    g, mn, sig, batches, test = world
    tables0, u, params0, ts = _setup(seed, steps)
    enc, denc, mlp = _model(mn, sig, tables0, params0)
    tr = HashNeRFTrainer(enc, mlp, near=MP.NEAR, far=MP.FAR, num_samples=MP.S, total_steps=steps, precision=BF16)
    t_eval = torch.linspace(MP.NEAR, MP.FAR, MP.S, device=DEV)
    curve = []
    for k in range(steps):
        tr.step(*batches[k % MP.NB], t=ts[k])
        if k + 1 in eval_steps:
            curve.append(_psnr(tr.render(test[0], test[1], test[2], t=t_eval), test[3]))
    return curve


def _train_dropin(world, seed, steps, eval_steps):
    from hbr_amd.vol_renderer import Volume_Renderer
    g, mn, sig, batches, test = world
    tables0, u, params0, ts = _setup(seed, steps)
    enc, denc, mlp = _model(mn, sig, tables0, params0)
    nerf = torch.nn.DataParallel(mlp, device_ids=[0])
    vr = Volume_Renderer(H=8, W=8, K=torch.eye(3), near=MP.NEAR, far=MP.FAR, device=DEV, Pos_encode=enc, Dir_encode=denc,
                         max_dim=2 ** 10, sigma_val=sig.to(DEV), mu=mn.to(DEV))
    oe = torch.optim.Adam(enc.Embedding_list.parameters(), lr=0.05)
    om = torch.optim.AdamW(nerf.parameters(), lr=0.005)
    se = torch.optim.lr_scheduler.CosineAnnealingLR(oe, T_max=steps, eta_min=1e-4)
    sm = torch.optim.lr_scheduler.CosineAnnealingLR(om, T_max=steps, eta_min=1e-4)
    crit = torch.nn.MSELoss()
    t_eval = torch.linspace(MP.NEAR, MP.FAR, MP.S, device=DEV)
    curve = []
    for k in range(steps):
        o, d, dn, gt = batches[k % MP.NB]
        with torch.autocast("cuda", dtype=torch.bfloat16):
            Cr, Cf, _ = vr.vol_render(nerf, d, o, num_samples=MP.S, t=ts[k], update_mask=False, dir_norm=dn, hierarchical=False)
            loss = crit(Cr, gt) + crit(Cf, gt)
        loss.backward()
        oe.step(); om.step(); se.step(); sm.step()
        om.zero_grad(set_to_none=True); oe.zero_grad(set_to_none=True)
        if k + 1 in eval_steps:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                C = vr.vol_render(nerf, test[1], test[0], num_samples=MP.S, t=t_eval, update_mask=False, dir_norm=test[2], hierarchical=False)[0]
            curve.append(_psnr(C, test[3]))
    return curve


@pytest.mark.parametrize("route", ["fused", "dropin"])
def test_converged_psnr_within_a_tenth_of_a_db_of_the_reference(world, route):
    g = world[0]
    steps, ev = int(g["steps"]), [int(v) for v in g["eval_steps"]]
    train = _train_fused if route == "fused" else _train_dropin
    deltas, lines = [], []
    for i, seed in enumerate(int(s) for s in g["seeds"]):
        # the per-seed inputs regenerate to what the reference run used
        tables0, u, params0 = MP.seeded_inputs(seed, steps)
        assert abs(MP.checksum(tables0, u, *[v.numpy() for v in params0.values()]) - float(g["input_checksum"][i])) < 1e-6 * float(g["input_checksum"][i])
        curve = np.array(train(world, seed, steps, set(ev)))
        ref = g["psnr"][i]
        tail = curve[int(len(curve) * 0.8):]
        assert tail.max() - tail.min() < 0.1, f"seed {seed}: not on a plateau ({tail})"
        deltas.append(curve[-1] - ref[-1])
        lines.append(f"seed {seed}: reference {ref[-1]:.3f} dB, HIP {route} {curve[-1]:.3f} dB, delta {deltas[-1]:+.3f}; "
                     f"mid-run (step {ev[len(ev) // 4]}) delta {curve[len(ev) // 4] - ref[len(ev) // 4]:+.3f}")
    deltas = np.array(deltas)
    report = "\n".join(lines) + f"\nmean delta {deltas.mean():+.3f} dB, std {deltas.std():.3f}, max |delta| {np.abs(deltas).max():.3f}"
    print(report)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, f"psnr_converged_{route}.txt"), "w") as f:
            f.write(report + "\n")
    assert abs(deltas.mean()) <= 0.1, report
    assert np.abs(deltas).max() <= 0.5, report
