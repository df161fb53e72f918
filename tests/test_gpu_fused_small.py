"""GPU (-m gpu): the training step's small launches folded together (round 3) against the separate launches they replace.

* hbr_render_prologue == hbr_strat_sample + hbr_dir_encode + the weight packing of hbr_mlp_fwd, bit for bit
  (helper.py:234-235, encoder.py:25-32)
* hbr_composite_loss_fwd_bwd == hbr_composite_fwd + hbr_mse2_loss_fwd_bwd + hbr_composite_bwd (helper.py:53-107,
  train_hash2.py:221): colours bit-identical; gradients bit-identical on the generic kernel (strided sigma / rgb) and
  within 2e-5 of the largest on the vector kernel of the [N,4] layout (it keeps the forward's transmittance where the
  separate backward recomputes it from a differently associated sum); loss to fp32 summation order; vs the CPU oracle too
* hbr_adam_step_multi == two hbr_adam_step launches, bit for bit (train_hash2.py:227-228)
* HashNeRFTrainer with and without the folded launches: identical parameters after several steps
* drop-in vol_render with t=None draws its depths inside the prologue from torch's CUDA generator: manual_seed replays
"""
import numpy as np
import pytest
import torch

import ref_cpu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scene(R=96, seed=3):
    o, d, dn, gt = ref_cpu.synthetic_rays(R, seed=seed)
    mn, mx, sig = ref_cpu.bbox_mu_sigma(o, d)
    return (o.to(DEV), d.to(DEV), dn.to(DEV), gt.to(DEV)), mn, sig


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_prologue_equals_separate_launches(precision):
    from hbr_amd import ops
    from hbr_amd._lib import BF16, F32, PLANAR
    prec = BF16 if precision == "bf16" else F32
    (o, d, dn, gt), mn, sig = _scene(R=333)
    S = 77
    flat = torch.cat([v.reshape(-1) for v in _ordered(ref_cpu.mlp_init(5))]).to(DEV)
    t_ref = ops.strat_sample(2.0, 6.0, S, DEV, seed=11, offset=5)
    pe_ref = ops.dir_encode(d, 4)
    t, pe = ops.render_prologue(DEV, prec, params=flat, rays_d=d, strat=(2.0, 6.0, S, None, 11, 5))
    assert torch.equal(t, t_ref) and torch.equal(pe, pe_ref)
    u = torch.rand(S, device=DEV)
    t2, pe2 = ops.render_prologue(DEV, prec, rays_d=d, strat=(2.0, 6.0, S, u, 0, 0))
    assert torch.equal(t2, ops.strat_sample(2.0, 6.0, S, DEV, u=u)) and torch.equal(pe2, pe_ref)
    t3, pe3 = ops.render_prologue(DEV, prec, params=flat)  # the image alone
    assert t3 is None and pe3 is None
    # the image it leaves is the one mlp_fwd packs: forward with image_ready == forward that packs
    N = 333 * S
    feat = (torch.randn(16, N, 2, device=DEV) * 0.3).to(torch.bfloat16 if prec == BF16 else torch.float32)
    a = ops.mlp_fwd(feat, PLANAR, pe_ref, S, flat, prec, image_ready=False)
    ops.render_prologue(DEV, prec, params=flat)
    b = ops.mlp_fwd(feat, PLANAR, pe_ref, S, flat, prec, image_ready=None)
    assert torch.equal(a, b)
    # ... and the record behind image_ready=None notices other parameters
    flat2 = flat * 1.5
    c = ops.mlp_fwd(feat, PLANAR, pe_ref, S, flat2, prec, image_ready=None)
    assert torch.equal(c, ops.mlp_fwd(feat, PLANAR, pe_ref, S, flat2, prec, image_ready=False)) and not torch.equal(c, a)
    flat2.mul_(0.5)  # in-place update (an optimiser step): version counter moves, repack
    e = ops.mlp_fwd(feat, PLANAR, pe_ref, S, flat2, prec, image_ready=None)
    assert torch.equal(e, ops.mlp_fwd(feat, PLANAR, pe_ref, S, flat2, prec, image_ready=False))


def _ordered(params):
    return [params[f"{seq}.{i}.{k}"] for seq in ("sig_model", "col_model") for i in (0, 2, 4) for k in ("weight", "bias")]


@pytest.mark.parametrize("R,S,per_ray_t,masked", [(64, 32, False, False), (1001, 128, False, True), (37, 300, True, False), (3, 5, False, False)])
def test_composite_loss_fused_equals_three_launches_and_oracle(R, S, per_ray_t, masked):
    from hbr_amd import ops
    rng = np.random.default_rng(R + S)
    out = torch.from_numpy(rng.normal(0, 1, (R * S, 4)).astype(np.float32))
    out[:, 3] = torch.from_numpy(rng.normal(0.5, 4.0, R * S).astype(np.float32))  # sigma: negative and < -10 values included
    out[::17, 3] = -12.0
    gt = torch.from_numpy(rng.uniform(0, 1, (R, 3)).astype(np.float32))
    dn = torch.from_numpy(rng.uniform(0.9, 1.3, R).astype(np.float32))
    if per_ray_t:
        t = torch.sort(torch.from_numpy(rng.uniform(2, 6, (R, S)).astype(np.float32)), dim=-1).values
    else:
        t = ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.from_numpy(rng.uniform(0, 1, S).astype(np.float32)))
    keep = torch.from_numpy((rng.uniform(0, 1, R * S) > 0.3).astype(np.uint8)) if masked else None
    o_d, gt_d, dn_d, t_d = out.to(DEV), gt.to(DEV), dn.to(DEV), t.to(DEV).contiguous()
    k_d = keep.to(DEV) if masked else None
    # the three separate launches
    Cr, _ = ops.composite_fwd(t_d, o_d.data_ptr(), 4, o_d.data_ptr() + 12, 4, dn_d, R, S, want_wts=False)
    loss_ref, dCr = ops.mse2_loss(Cr, gt_d)
    d_ref = torch.empty_like(o_d)
    ops.composite_bwd(t_d, o_d.data_ptr(), 4, o_d.data_ptr() + 12, 4, dn_d, R, S, dCr, d_ref.data_ptr(), d_ref.data_ptr() + 12, keep=k_d)
    # one launch
    loss, d_out, Cr2 = ops.composite_loss_fwd_bwd(t_d, o_d, dn_d, R, S, gt_d, keep=k_d, want_Cr=True)
    assert torch.equal(Cr2, Cr)
    if S <= 256:  # the vector kernel
        assert float((d_out - d_ref).abs().max()) <= 2e-5 * float(d_ref.abs().max())
        assert torch.equal(d_out == 0, d_ref == 0) or float(((d_out == 0) != (d_ref == 0)).float().mean()) < 1e-4
    else:         # the generic kernel repeats the separate kernels' operations exactly
        assert torch.equal(d_out, d_ref)
    assert abs(float(loss) - float(loss_ref)) <= 2e-6 * abs(float(loss_ref))
    # the generic kernel through its own door: rgb / sigma in separate buffers (calc_color's layout)
    rgb_s, sg_s = o_d[:, 0:3].contiguous(), o_d[:, 3].contiguous()
    d_rgb, d_sg = torch.empty_like(rgb_s), torch.empty_like(sg_s)
    from hbr_amd._lib import check, lib
    l2 = torch.empty((), device=DEV)
    ws = torch.zeros(lib().hbr_composite_loss_workspace_bytes(R), dtype=torch.uint8, device=DEV)
    check(lib().hbr_composite_loss_fwd_bwd(t_d.data_ptr(), 0 if t_d.dim() == 1 else S, rgb_s.data_ptr(), 3, sg_s.data_ptr(), 1, dn_d.data_ptr(), R, S,
                                           gt_d.data_ptr(), 1.0, l2.data_ptr(), None, d_rgb.data_ptr(), d_sg.data_ptr(),
                                           k_d.data_ptr() if masked else None, ws.data_ptr(), torch.cuda.current_stream().cuda_stream), "closs")
    assert torch.equal(d_rgb, d_ref[:, 0:3]) and torch.equal(d_sg, d_ref[:, 3])
    assert abs(float(l2) - float(loss_ref)) <= 2e-6 * abs(float(loss_ref))
    loss_b, d_b, none = ops.composite_loss_fwd_bwd(t_d, o_d, dn_d, R, S, gt_d, keep=k_d)  # again: bitwise reproducible, ticket reset
    assert none is None and torch.equal(loss_b, loss) and torch.equal(d_b, d_out)
    # the oracle (autograd through its compositing + loss)
    oo = out.clone().requires_grad_(True)
    rgb, sg = oo[:, 0:3].reshape(R, S, 3), oo[:, 3].reshape(R, S)
    if masked:  # masked samples are constants (zeros), not model outputs (vol_renderer.py:213-221)
        m = keep.bool().reshape(R, S)
        rgb, sg = rgb * m[..., None], sg * m
    C = (ref_cpu.composite_per_ray(t, rgb, sg, dn[:, None]) if per_ray_t else ref_cpu.composite(t, rgb, sg, dn[:, None]))[0]
    lo = ref_cpu.train_loss(C, gt)
    lo.backward()
    if not masked:  # (the kernel composites sigma/rgb as given: the mask only gates the gradients)
        assert abs(float(loss) - float(lo)) <= 1e-4 * abs(float(lo)) + 1e-7
        scale = float(oo.grad.abs().max())
        assert float((d_out.cpu() - oo.grad).abs().max()) <= 2e-4 * scale + 1e-9


def test_adam_multi_equals_two_launches():
    from hbr_amd import ops
    torch.manual_seed(0)
    n1, n2 = 2 * 4096 * 16, 14227
    def mk(n):
        n4 = (n + 3) // 4 * 4
        return [torch.randn(n4, device=DEV)[:n] * s for s in (0.1, 1e-3, 1e-3, 1e-6)]
    (p1, g1, m1, v1), (p2, g2, m2, v2) = mk(n1), mk(n2)
    v1.abs_(); v2.abs_()
    ref = [x.clone() for x in (p1, m1, v1, p2, m2, v2)]
    common = dict(beta1=0.9, beta2=0.999, eps=1e-8, step=7, grad_scale=0.5)
    ops.adam_step(ref[0], g1, ref[1], ref[2], lr=0.05, weight_decay=0.0, **common)
    ops.adam_step(ref[3], g2, ref[4], ref[5], lr=0.005, weight_decay=0.01, **common)
    ops.adam_step_multi([dict(p=p1, g=g1, m=m1, v=v1, lr=0.05, weight_decay=0.0, **common),
                         dict(p=p2, g=g2, m=m2, v=v2, lr=0.005, weight_decay=0.01, **common)])
    for a, b in zip(ref, (p1, m1, v1, p2, m2, v2)):
        assert torch.equal(a, b)
    assert not torch.equal(p2, mk(n2)[0])


def test_trainer_folded_launches_equal_separate_launches():
    """One step of HashNeRFTrainer with the folded launches against the separate ones: same loss, same gradient buffer
    (to the vector compositing kernel's last-bit difference in the transmittance), same parameters after the step."""
    from hbr_amd._lib import BF16
    from hbr_amd.trainer import HashNeRFTrainer, build_default_model
    (o, d, dn, gt), mn, sig = _scene(R=2048, seed=9)
    states = []
    for fused in (True, False):
        enc, denc, mlp = build_default_model(mn, sig, DEV, T=2 ** 12, seed=0)
        with torch.no_grad():
            for lvl in enc.Embedding_list:
                lvl.weight.mul_(3000.0)  # features of order 0.3: a field whose gradients are not all ~0
        tr = HashNeRFTrainer(enc, mlp, num_samples=64, total_steps=100, precision=BF16, seed=3)
        tr.fused_small = fused
        tr.fused_render = False  # (round 4's one-launch MLP + compositing has its own comparison: tests/test_gpu_render_bwd.py)
        p0 = tr.tables.clone()
        loss = float(tr.step(o, d, dn, gt))
        states.append((tr.grad.clone(), tr.tables.clone() - p0, tr.flat.clone(), loss))
    (ga, da, fa, la), (gb, db, fb, lb) = states
    assert abs(la - lb) <= 2e-6 * abs(lb)
    assert float((ga - gb).abs().max()) <= 1e-4 * float(gb.abs().max())
    assert float(((da - db).abs() > 1e-6).float().mean()) < 1e-3  # Adam's first step is +-lr: only ~zero gradients can flip
    assert float((fa - fb).abs().max()) <= 1e-5


def test_dropin_depths_follow_torch_generator():
    """vol_render(t=None): the depths are drawn inside the prologue launch from (seed, offset) of torch's CUDA generator
    and the generator advances as it would for torch.rand - manual_seed replays, consecutive calls differ."""
    from hbr_amd.encoder import PositionalEncoder
    from hbr_amd.trainer import build_default_model
    from hbr_amd.vol_renderer import Volume_Renderer
    (o, d, dn, gt), mn, sig = _scene(R=128, seed=4)
    enc, denc, mlp = build_default_model(mn, sig, DEV, T=2 ** 10, seed=0)
    with torch.no_grad():
        for lvl in enc.Embedding_list:
            lvl.weight.uniform_(-0.5, 0.5)
    vr = Volume_Renderer(H=8, W=8, K=torch.eye(3), near=2.0, far=6.0, device=DEV, Pos_encode=enc, Dir_encode=PositionalEncoder(3, 4),
                         sigma_val=sig, mu=mn)
    def render():
        with torch.no_grad():
            return vr.vol_render(mlp, d, o, num_samples=48, dir_norm=dn, hierarchical=False)[0]
    torch.manual_seed(5)
    a, b = render(), render()
    torch.manual_seed(5)
    a2 = render()
    st = torch.cuda.get_rng_state(0)
    c = render()
    torch.cuda.set_rng_state(st, 0)
    c2 = render()
    assert torch.equal(a, a2) and torch.equal(c, c2) and not torch.equal(a, b)
    from hbr_amd import helper
    torch.manual_seed(5)
    t = helper.strat_sampler(2.0, 6.0, 48, device=DEV)
    with torch.no_grad():
        assert torch.equal(vr.vol_render(mlp, d, o, num_samples=48, t=t, dir_norm=dn, hierarchical=False)[0], a)
