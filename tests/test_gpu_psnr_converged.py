"""GPU (-m gpu): "PSNR within 0.1 dB of reference" (BASELINE.json north_star; metric = helper.py:301-304), pinned by the
reference's own modules trained to a plateau (oracle/make_psnr_golden.py; loop of train_hash2.py:211-234, fp32 on CPU;
16 x 1024 rays x 64 samples = 65 536 points per step - the shipped LDS scatter kernel; cosine schedule ending at the
2000-step horizon; five seeded initialisations).

What the fixtures show, and what is therefore asserted:

1. SAME MODEL, both implementations (g15b: the reference's trained tables + MLP of one run, and its own render of the
   held-out rays): the HIP render of those weights has the reference's PSNR to 0.01 dB in fp32 and 0.1 dB in bf16 - the
   north star's 0.1 dB, two-sided, where it is well defined.
2. SAME LOOP, first steps (g15 `loss_head`): the drop-in route's first 16 losses equal the reference's to 1e-3 - the
   training step is the reference's before rounding-level differences have been amplified.
3. TRAINED TO THE PLATEAU from identical initial parameters, rays and jitter: the final PSNR of this problem is itself
   sensitive at the 1 dB level - the reference re-run with every initial table entry moved by ONE fp32 ulp (g15
   `psnr_ulp`) ends -0.2 ... +0.9 dB from its own unperturbed run (mean +0.3, sd 0.35), and the exact-fp32 HIP drop-in
   route, whose arithmetic differs from the reference's only in summation order, -0.4 ... +1.7 dB.  So no implementation
   that is not bit-identical can be held to 0.1 dB per trajectory; what is asserted is the one-sided statement that
   matters - no quality loss: mean over the seeds of (HIP - reference) >= -0.3 dB (measured: fused bf16 +0.65, drop-in
   bf16 +1.12), every seed within 3 dB, and the HIP curves plateau like the reference's (last 20 % of the horizon moves
   < 0.2 dB).  The measured table is in BASELINE.md / profiles/r03_psnr_converged.txt.
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
            enc.Embedding_list[l].weight.copy_(torch.as_tensor(tables0[l]))
        for k, v in params0.items():
            seq, idx, kind = k.split(".")
            getattr(getattr(mlp, seq)[int(idx)], kind).copy_(torch.as_tensor(v))
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


def _dropin(world, enc, denc, mlp, steps):
    from hbr_amd.vol_renderer import Volume_Renderer
    g, mn, sig, batches, test = world
    nerf = torch.nn.DataParallel(mlp, device_ids=[0])
    vr = Volume_Renderer(H=8, W=8, K=torch.eye(3), near=MP.NEAR, far=MP.FAR, device=DEV, Pos_encode=enc, Dir_encode=denc,
                         max_dim=2 ** 10, sigma_val=sig.to(DEV), mu=mn.to(DEV))
    oe = torch.optim.Adam(enc.Embedding_list.parameters(), lr=0.05)
    om = torch.optim.AdamW(nerf.parameters(), lr=0.005)
    se = torch.optim.lr_scheduler.CosineAnnealingLR(oe, T_max=steps, eta_min=1e-4)
    sm = torch.optim.lr_scheduler.CosineAnnealingLR(om, T_max=steps, eta_min=1e-4)
    return nerf, vr, (oe, om, se, sm), torch.nn.MSELoss()


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


def _train_dropin(world, seed, steps, eval_steps, autocast=True, losses=None, stop=None):
    g, mn, sig, batches, test = world
    tables0, u, params0, ts = _setup(seed, steps)
    enc, denc, mlp = _model(mn, sig, tables0, params0)
    nerf, vr, (oe, om, se, sm), crit = _dropin(world, enc, denc, mlp, steps)
    t_eval = torch.linspace(MP.NEAR, MP.FAR, MP.S, device=DEV)
    curve = []
    for k in range(stop or steps):
        o, d, dn, gt = batches[k % MP.NB]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            Cr, Cf, _ = vr.vol_render(nerf, d, o, num_samples=MP.S, t=ts[k], update_mask=False, dir_norm=dn, hierarchical=False)
            loss = crit(Cr, gt) + crit(Cf, gt)
        if losses is not None:
            losses.append(float(loss.detach()))
        loss.backward()
        oe.step(); om.step(); se.step(); sm.step()
        om.zero_grad(set_to_none=True); oe.zero_grad(set_to_none=True)
        if k + 1 in eval_steps:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                C = vr.vol_render(nerf, test[1], test[0], num_samples=MP.S, t=t_eval, update_mask=False, dir_norm=test[2], hierarchical=False)[0]
            curve.append(_psnr(C, test[3]))
    return curve


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_same_trained_model_renders_to_the_reference_psnr(world, precision):
    """The reference's own trained weights (one of its 2000-step runs), rendered on the held-out rays by the HIP path."""
    from hbr_amd.vol_renderer import Volume_Renderer
    gw = load_golden("g15b_trained_weights.npz")
    g, mn, sig, batches, test = world
    enc, denc, mlp = _model(mn, sig, gw["tables"], {k[2:]: v for k, v in gw.items() if k.startswith("p.")})
    vr = Volume_Renderer(H=8, W=8, K=torch.eye(3), near=MP.NEAR, far=MP.FAR, device=DEV, Pos_encode=enc, Dir_encode=denc,
                         max_dim=2 ** 10, sigma_val=sig.to(DEV), mu=mn.to(DEV))
    t_eval = torch.linspace(MP.NEAR, MP.FAR, MP.S, device=DEV)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=precision == "bf16"):
        C = vr.vol_render(mlp, test[1], test[0], num_samples=MP.S, t=t_eval, update_mask=False, dir_norm=test[2], hierarchical=False)[0]
    p, p_ref = _psnr(C, test[3]), float(gw["psnr"])
    err = float((C.cpu() - torch.from_numpy(gw["Cr_eval"])).abs().max())
    print(f"{precision}: reference {p_ref:.4f} dB, HIP {p:.4f} dB, max |dC| {err:.2e}")
    assert abs(p - p_ref) <= (0.01 if precision == "fp32" else 0.1)
    assert err <= (2e-4 if precision == "fp32" else 3e-2)


def test_first_steps_follow_the_reference_loss_curve(world):
    g = world[0]
    steps = int(g["steps"])
    for i, seed in enumerate(int(s) for s in g["seeds"][:2]):
        losses = []
        _train_dropin(world, seed, steps, set(), autocast=False, losses=losses, stop=16)
        ref = g["loss_head"][i]
        assert np.allclose(losses, ref, rtol=1e-3), (seed, np.abs(np.array(losses) / ref - 1).max())


@pytest.mark.parametrize("route", ["fused", "dropin"])
def test_converged_psnr_is_not_below_the_reference(world, route):
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
        assert tail.max() - tail.min() < 0.2, f"seed {seed}: not on a plateau ({tail})"
        deltas.append(curve[-1] - ref[-1])
        lines.append(f"seed {seed}: reference {ref[-1]:.3f} dB (re-run with 1-ulp-moved tables {g['psnr_ulp'][i][-1]:.3f}), HIP {route} {curve[-1]:.3f} dB, "
                     f"delta {deltas[-1]:+.3f}; at step {ev[len(ev) // 4]} delta {curve[len(ev) // 4] - ref[len(ev) // 4]:+.3f}")
    deltas = np.array(deltas)
    ulp = g["psnr_ulp"][:, -1] - g["psnr"][:, -1]
    report = "\n".join(lines) + (f"\nHIP {route} - reference: mean {deltas.mean():+.3f} dB, sd {deltas.std():.3f}, max |delta| {np.abs(deltas).max():.3f}"
                                 f"\nreference (1 ulp) - reference: mean {ulp.mean():+.3f} dB, sd {ulp.std():.3f}, max |delta| {np.abs(ulp).max():.3f}")
    print(report)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, f"psnr_converged_{route}.txt"), "w") as f:
            f.write(report + "\n")
    assert deltas.mean() >= -0.3, report
    assert np.abs(deltas).max() <= 3.0, report
