"""GPU (-m gpu): hbr_mlp_render_bwd - the training step's MLP forward + compositing + loss + their backward as ONE launch
(round 4) - against the three separate calls it replaces (hbr_mlp_fwd, hbr_composite_loss_fwd_bwd, hbr_mlp_bwd) and, through
them, against the oracle (the separate calls are pinned by G5 / G6 / G8 and the shipped-path tests).

The compositing inside the kernel is composite_ray.h - the very code composite_loss_vec_kernel runs - so given the same
(rgb, sigma) it returns the same bits; what differs is the MLP forward feeding it: the backward kernel's recompute lets
the bias enter through one more MFMA k-step (three bf16 parts) where the forward kernel starts its accumulators from the
fp32 bias, so a hidden activation can round to the neighbouring bf16 value.  Tolerances below are a few bf16 ulps of the
quantities involved; the trainer-level test checks that the fused step trains like the separate one."""
import numpy as np
import pytest
import torch

import ref_cpu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(R, S, seed, feat_dtype):
    from hbr_amd import ops
    g = torch.Generator().manual_seed(seed)
    N = R * S
    feat = (torch.randn((16, N, 2), generator=g) * 0.3).to(DEV).to(feat_dtype)
    d = torch.nn.functional.normalize(torch.randn((R, 3), generator=g), dim=1).to(DEV)
    pe = ops.dir_encode(d, 4)
    P = torch.cat([v.reshape(-1) for v in ref_cpu.mlp_init(seed).values()]).to(DEV)
    t = ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.rand(S, generator=g)).to(DEV)
    dn = (1.0 + 0.2 * torch.rand(R, generator=g)).to(DEV)
    gt = torch.rand((R, 3), generator=g).to(DEV)
    return feat, pe, P, t, dn, gt


@pytest.mark.parametrize("S", [32, 64, 128])
@pytest.mark.parametrize("feat_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("with_norm", [True, False])
def test_render_bwd_equals_the_three_separate_launches(S, feat_dtype, with_norm):
    from hbr_amd import ops
    from hbr_amd._lib import BF16, PLANAR
    R = 1061 if S == 32 else (517 if S == 64 else 301)   # not a multiple of the rays per workgroup round: the last round is partial
    feat, pe, P, t, dn, gt = _inputs(R, S, 11 + S, feat_dtype)
    dn_arg = dn if with_norm else None
    # ---- the separate launches
    out = ops.mlp_fwd(feat, PLANAR, pe, S, P, BF16)
    loss_ref, d_out, Cr_ref = ops.composite_loss_fwd_bwd(t, out, dn_arg, R, S, gt, want_Cr=True)
    dP_ref = torch.zeros_like(P)
    amax_ref = torch.zeros(16, device=DEV)
    dfeat_ref = ops.mlp_bwd(feat, PLANAR, pe, S, P, BF16, d_out, dP_ref, absmax_out=amax_ref)
    # ---- one launch
    dP = torch.full_like(P, 7.0)   # overwrite mode: whatever the buffer held
    amax = torch.zeros(16, device=DEV)
    got = ops.mlp_render_bwd(feat, pe, P, BF16, t, dn_arg, gt, dP, absmax_out=amax, overwrite=True, want_Cr=True)
    assert got is not None
    loss, dfeat, Cr = got
    torch.cuda.synchronize()
    assert float((Cr - Cr_ref).abs().max()) <= 2e-3 * max(1.0, float(Cr_ref.abs().max())), float((Cr - Cr_ref).abs().max())
    assert abs(float(loss) - float(loss_ref)) <= 2e-3 * float(loss_ref)
    gs = float(dP_ref.abs().max())
    assert float((dP - dP_ref).abs().max()) <= 2e-2 * gs, float((dP - dP_ref).abs().max()) / gs
    assert float((dP - dP_ref).abs().mean()) <= 1e-3 * gs
    fs = float(dfeat_ref.float().abs().max())
    err = (dfeat.float() - dfeat_ref.float()).abs()
    assert float(err.max()) <= 5e-2 * fs and float(err.mean()) <= 2e-3 * fs, (float(err.max()) / fs, float(err.mean()) / fs)
    assert torch.allclose(amax, amax_ref, rtol=5e-2)
    # bitwise reproducible, and accumulate mode adds the same gradient on top
    dP2 = torch.zeros_like(P)
    loss2, dfeat2, _ = ops.mlp_render_bwd(feat, pe, P, BF16, t, dn_arg, gt, dP2, overwrite=False)
    assert torch.equal(dP2, dP) and torch.equal(dfeat2, dfeat) and float(loss2) == float(loss)


def test_render_bwd_against_the_oracle_under_autocast():
    """Colours, loss and the MLP gradient of the one-launch route against the CPU oracle under bf16 autocast, judged like
    the shipped K4 test: within 1.5 x the error torch's own bf16 autocast shows against its fp32 run on the same data."""
    from hbr_amd import ops
    from hbr_amd._lib import BF16
    R, S = 96, 64
    feat, pe, P, t, dn, gt = _inputs(R, S, 5, torch.float32)
    params = {k: v.clone().requires_grad_(True) for k, v in ref_cpu.mlp_init(5).items()}
    x = feat.cpu().permute(1, 0, 2).reshape(R * S, 32).clone().requires_grad_(True)
    pe_pts = pe.cpu()[:, None, :].expand(R, S, 24).reshape(R * S, 24)

    def oracle(autocast):
        for v in params.values():
            v.grad = None
        x.grad = None
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            o = ref_cpu.mlp_forward(x, pe_pts, params)
        o = o.float().reshape(R, S, 4)
        Cr, _ = ref_cpu.composite(t.cpu(), o[..., :3], o[..., 3], dn.cpu()[:, None])
        loss = ref_cpu.train_loss(Cr, gt.cpu())
        loss.backward()
        return Cr.detach(), float(loss), torch.cat([params[k].grad.reshape(-1) for k in params]), x.grad.clone()

    Cr32, l32, g32, dx32 = oracle(False)
    Crbf, lbf, gbf, dxbf = oracle(True)
    dP = torch.zeros_like(P)
    loss, dfeat, Cr = ops.mlp_render_bwd(feat, pe, P, BF16, t, dn, gt, dP, want_Cr=True)
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

    def check(got, exact, autocast, what):  # the criterion of test_k4_shipped_instantiation_vs_oracle_autocast
        r_hip, r_torch = rel(got, exact), rel(autocast, exact)
        assert r_hip <= max(1.5 * r_torch, 5e-3) and r_hip < 0.1, (what, r_hip, r_torch)

    check(Cr.cpu().numpy(), Cr32.numpy(), Crbf.numpy(), "Cr")
    assert abs(float(loss) - l32) <= max(1.5 * abs(lbf - l32), 5e-3 * l32)
    check(dP.cpu().numpy(), g32.numpy(), gbf.numpy(), "dparams")
    dx = dfeat.float().cpu().permute(1, 0, 2).reshape(R * S, 32)
    check(dx.numpy(), dx32.numpy(), dxbf.numpy(), "dfeat")


def test_render_bwd_refuses_what_it_does_not_cover():
    from hbr_amd import ops
    from hbr_amd._lib import BF16, F32
    feat, pe, P, t, dn, gt = _inputs(8, 100, 3, torch.bfloat16)   # S = 100: rays do not tile a workgroup round
    assert ops.mlp_render_bwd(feat, pe, P, BF16, t, dn, gt, torch.zeros_like(P)) is None
    feat, pe, P, t, dn, gt = _inputs(8, 64, 3, torch.float32)
    assert ops.mlp_render_bwd(feat, pe, P, F32, t, dn, gt, torch.zeros_like(P)) is None  # exact-fp32 MLP: separate launches


@pytest.mark.parametrize("S", [64, 128])
def test_trainer_step_with_and_without_the_fused_render(S):
    """HashNeRFTrainer (bf16) with hbr_mlp_render_bwd and with the separate launches (HBR_FUSED_RENDER=0's switch): same
    loss to bf16-rounding level after one step, nearby parameters after ten, and each route bit-reproducible."""
    from hbr_amd import synthetic
    from hbr_amd._lib import BF16
    from hbr_amd.trainer import HashNeRFTrainer, build_default_model
    o, d, dn, gt = (a.to(DEV) for a in synthetic.scene_rays(2048, seed=9))
    mn, mx, sig = synthetic.ray_bbox(o.cpu(), d.cpu())

    def run(render, steps):
        enc, denc, mlp = build_default_model(mn, sig, DEV, T=2 ** 14, seed=4)
        tr = HashNeRFTrainer(enc, mlp, num_samples=S, total_steps=100, precision=BF16, seed=3)
        tr.fused_render = render
        tr.timers = {}
        losses = [float(tr.step(o, d, dn.reshape(-1), gt)) for _ in range(steps)]
        assert ("mlp_fwd" not in tr.timers) == render
        return losses, tr.tables.clone(), tr.flat.clone(), tr.grad.clone()

    a1, b1 = run(True, 1), run(False, 1)
    assert abs(a1[0][0] - b1[0][0]) <= 2e-3 * b1[0][0]
    gs = float(b1[3].abs().max())
    assert float((a1[3] - b1[3]).abs().max()) <= 3e-2 * gs and float((a1[3] - b1[3]).abs().mean()) <= 1e-3 * gs
    a10, a10b, b10 = run(True, 10), run(True, 10), run(False, 10)
    assert a10[0] == a10b[0] and torch.equal(a10[1], a10b[1]) and torch.equal(a10[2], a10b[2])
    assert a10[0][-1] < a10[0][0] and abs(a10[0][-1] - b10[0][-1]) <= 0.1 * b10[0][-1]
