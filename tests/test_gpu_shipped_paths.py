"""GPU (-m gpu): DIRECT oracle parity for the kernel instantiations the benchmark times, plus the
properties the round-2 scatter kernel adds.

* K2 as shipped: N >= 65536 points, planar bf16 dy, LDS-slice kernel with its workspace (fixed-point accumulation,
  slab flush) against oracle/ref_cpu.hash_encode_backward on the same bf16-rounded dy - rtol 1e-4 and the exact set of
  touched rows (reference hash_encoding.py:146-170, autograd's embedding_dense_backward).
* K2 determinism: integer accumulation + fixed-order reduce => two launches agree bit for bit.
* K2 over level sub-ranges exactly as the multi-GPU trainer launches it (trainer.py world > 1 branch) == one launch.
* K4 as shipped: planar bf16 features in, bf16 MFMA, bf16 feature gradient out, against the oracle under torch's bf16
  autocast on the same bf16-rounded features (reference test_hash.py:52-72 under train_hash2.py:218).
"""
import os

import numpy as np
import pytest
import torch

import ref_cpu
from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from hbr_amd import ops as o
    from hbr_amd import _lib as L
    assert L.lib().hbr_device_ok() == 1
    return o


def _scene(ops, R, S, T, seed):
    o, d, _, _ = ref_cpu.synthetic_rays(R, seed=seed)
    mn, mx, sig = ref_cpu.bbox_mu_sigma(o, d)
    sc = ref_cpu.level_scales(16, 2048.0, 16)
    geom = ops.HashGeom(tuple(float(v) for v in sc), tuple(float(v) for v in mn), float(sig), T, 2)
    t = ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.rand(S, generator=torch.Generator().manual_seed(seed + 1)))
    return o, d, t, mn, sig, sc, geom


def test_k2_shipped_instantiation_vs_oracle(ops):
    """131 072 points (1024 rays x 128 samples), T = 2^16, planar bf16 dy with a wide dynamic range (|dy| spans six
    decades, like gradients behind and in front of a surface) - the exact kernel bench.py times."""
    from hbr_amd._lib import PLANAR, lib
    R, S, L, T = 1024, 128, 16, 2 ** 16
    N = R * S
    o, d, t, mn, sig, sc, geom = _scene(ops, R, S, T, seed=41)
    assert lib().hbr_hash_bwd_workspace_bytes(N, L, T, 2, 0) > 0  # auto picks the LDS-slice kernel at this size
    rng = np.random.default_rng(42)
    mag = 10.0 ** rng.uniform(-6, 0, (N, 1))
    dy32 = (rng.standard_normal((N, L * 2)) * mag).astype(np.float32)
    dy_bf = torch.from_numpy(dy32).bfloat16()                                # [N,32] bf16
    dy_planar = dy_bf.reshape(N, L, 2).permute(1, 0, 2).contiguous().to(DEV)  # [L,N,2]
    pts = ref_cpu.sample_points(o, d, t).reshape(-1, 3)
    ref = ref_cpu.hash_encode_backward(pts, dy_bf.float(), sc, mn, sig, T).numpy()

    got = torch.zeros((L, T, 2), device=DEV)
    ops.hash_encode_bwd(geom, dy_planar, got, rays=(o.to(DEV), d.to(DEV), t.to(DEV)), layout=PLANAR, algo=0)
    g = got.cpu().numpy()
    # per-contribution rounding to the 2^-44-of-max quantum and fp32 rounding of <= 1 chunk partial: rtol 1e-4
    assert np.allclose(g, ref, rtol=1e-4, atol=1e-6 * np.abs(ref).max())
    # index parity: no row outside the reference's touched set, and every row the reference gives more than the
    # fixed-point quantum (2^-43 of the level's largest |dy| at this N; |dy| spans six decades here and a trilinear
    # weight can be 1e-7, so a few thousand-billionth-sized entries legitimately round to zero)
    assert not np.any((g != 0) & (ref == 0))
    assert np.all((g != 0) | (np.abs(ref) < 1e-10 * np.abs(ref).max()))
    # the float-atomic flush (minimal workspace) and the global-atomics kernel agree with it
    for kw in (dict(algo=2, deterministic=False), dict(algo=1)):
        alt = torch.zeros((L, T, 2), device=DEV)
        ops.hash_encode_bwd(geom, dy_planar, alt, rays=(o.to(DEV), d.to(DEV), t.to(DEV)), layout=PLANAR, **kw)
        assert np.allclose(alt.cpu().numpy(), ref, rtol=1e-4, atol=1e-6 * np.abs(ref).max()), kw
    # a caller-supplied per-level maximum (any upper bound) gives the same bits as the internal pass when equal
    amax = dy_planar.float().abs().amax(dim=(1, 2))
    again = torch.zeros((L, T, 2), device=DEV)
    ops.hash_encode_bwd(geom, dy_planar, again, rays=(o.to(DEV), d.to(DEV), t.to(DEV)), layout=PLANAR, algo=2, dy_absmax=amax)
    assert torch.equal(again, got)


def test_k2_is_bitwise_reproducible_and_accumulates(ops):
    from hbr_amd._lib import PLANAR
    R, S, L, T = 2048, 128, 16, 2 ** 16
    o, d, t, mn, sig, sc, geom = _scene(ops, R, S, T, seed=43)
    rays = (o.to(DEV), d.to(DEV), t.to(DEV))
    gen = torch.Generator(device=DEV).manual_seed(44)
    dy = (torch.randn((L, R * S, 2), device=DEV, generator=gen) * 1e-3).bfloat16()
    runs = []
    for _ in range(3):
        g = torch.zeros((L, T, 2), device=DEV)
        ops.hash_encode_bwd(geom, dy, g, rays=rays, layout=PLANAR, algo=2)
        runs.append(g)
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    # ACCUMULATES into dtables: a second call on top of the first doubles every entry exactly (x + x is exact)
    ops.hash_encode_bwd(geom, dy, runs[0], rays=rays, layout=PLANAR, algo=2)
    assert torch.equal(runs[0], 2 * runs[1])


@pytest.mark.parametrize("algo", [1, 2])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_k2_level_halves_equal_single_launch(ops, algo, dt):
    """The multi-GPU trainer scatters levels [L/2, L) and then [0, L/2) with sub-geometries on slices of the planar
    gradient, so that each half's all-reduce overlaps the other half's kernel (trainer.py).  Same result as one launch:
    bit-identical for the fixed-point kernel, within atomic-ordering noise for the global-atomics one."""
    from hbr_amd._lib import PLANAR
    R, S, L, T = 600, 128, 16, 2 ** 16
    o, d, t, mn, sig, sc, geom = _scene(ops, R, S, T, seed=45)
    rays = (o.to(DEV), d.to(DEV), t.to(DEV))
    gen = torch.Generator(device=DEV).manual_seed(46)
    dy = torch.randn((L, R * S, 2), device=DEV, generator=gen).to(dt)
    one = torch.zeros((L, T, 2), device=DEV)
    ops.hash_encode_bwd(geom, dy, one, rays=rays, layout=PLANAR, algo=algo)
    two = torch.zeros((L, T, 2), device=DEV)
    half = L // 2
    for k, (lo, hi) in enumerate(((half, L), (0, half))):
        sub = ops.HashGeom(geom.scales[lo:hi], geom.mu, geom.sigma, geom.T, geom.F)
        # the second LDS launch re-uses the first one's normalised coordinates (algo 3), as the trainer does
        ops.hash_encode_bwd(sub, dy[lo:hi], two[lo:hi], rays=rays, layout=PLANAR, algo=3 if (algo == 2 and k == 1) else algo)
    if algo == 2:
        assert torch.equal(one, two)
        three = torch.zeros((L, T, 2), device=DEV)
        for lo, hi in ((half, L), (0, half)):  # ... and two independent algo-2 launches give the same bits
            sub = ops.HashGeom(geom.scales[lo:hi], geom.mu, geom.sigma, geom.T, geom.F)
            ops.hash_encode_bwd(sub, dy[lo:hi], three[lo:hi], rays=rays, layout=PLANAR, algo=2)
        assert torch.equal(one, three)
    else:
        assert torch.allclose(one, two, rtol=1e-4, atol=1e-5 * float(one.abs().max()))


def test_k2_zero_and_nonfinite_gradients(ops):
    """A level whose gradient is all zero contributes nothing; a NaN or inf in a level's dy poisons that level (as
    float accumulation would) and leaves the others exact."""
    from hbr_amd._lib import PLANAR
    R, S, L, T = 512, 128, 16, 2 ** 14
    o, d, t, mn, sig, sc, geom = _scene(ops, R, S, T, seed=47)
    rays = (o.to(DEV), d.to(DEV), t.to(DEV))
    gen = torch.Generator(device=DEV).manual_seed(48)
    dy = torch.randn((L, R * S, 2), device=DEV, generator=gen)
    clean = torch.zeros((L, T, 2), device=DEV)
    ops.hash_encode_bwd(geom, dy, clean, rays=rays, layout=PLANAR, algo=2)
    dy2 = dy.clone()
    dy2[3] = 0.0
    dy2[5, 1234, 1] = float("nan")
    dy2[9, 77, 0] = float("inf")
    g = torch.zeros((L, T, 2), device=DEV)
    ops.hash_encode_bwd(geom, dy2, g, rays=rays, layout=PLANAR, algo=2)
    assert float(g[3].abs().max()) == 0.0
    assert bool(torch.isnan(g[5]).any()) and bool(torch.isnan(g[9]).any())
    keep = [l for l in range(L) if l not in (3, 5, 9)]
    assert torch.equal(g[keep], clean[keep])


def test_k2_refuses_what_it_cannot_do(ops):
    from hbr_amd._lib import HbrError, PLANAR
    R, S, L, T = 512, 128, 16, 2 ** 14
    o, d, t, mn, sig, sc, geom = _scene(ops, R, S, T, seed=49)
    rays = (o.to(DEV), d.to(DEV), t.to(DEV))
    g = torch.zeros((L, T, 2), device=DEV)
    with pytest.raises(HbrError):  # fp16 is what torch.cuda.amp.autocast() produces: refuse, do not reinterpret as bf16
        ops.hash_encode_bwd(geom, torch.zeros((L, R * S, 2), device=DEV, dtype=torch.float16), g, rays=rays, layout=PLANAR)
    with pytest.raises(HbrError):
        ops.mlp_fwd(torch.zeros((64, 32), device=DEV, dtype=torch.float64), 0, torch.zeros((64, 24), device=DEV), 1,
                    torch.zeros(14227, device=DEV), 0)
    from hbr_amd._lib import lib
    big = ops.HashGeom(geom.scales, geom.mu, geom.sigma, 2 ** 29, 2)  # row offsets would overflow 32 bits in the LDS kernel
    assert lib().hbr_hash_bwd_workspace_bytes(R * S, L, big.T, 2, 2) == 0

    def raw_call(gm, ws_ptr, ws_bytes):
        sc_, mu_ = gm.c_args()
        return lib().hbr_hash_encode_bwd(None, rays[0].data_ptr(), rays[1].data_ptr(), rays[2].data_ptr(), R, S, g.data_ptr(), PLANAR, 0,
                                         0, None, sc_, mu_, gm.sigma, L, gm.T, 2, g.data_ptr(), 2, ws_ptr, ws_bytes, None)

    scratch = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    assert raw_call(big, scratch.data_ptr(), scratch.numel()) == -2  # HBR_EUNSUPPORTED: algo 2 with T > 2^28
    assert raw_call(geom, None, 0) == -4                              # HBR_EWORKSPACE: algo 2 without its workspace
    assert raw_call(geom, scratch.data_ptr(), 1024) == -4
    torch.cuda.synchronize()


def test_k4_shipped_instantiation_vs_oracle_autocast(ops):
    """G5's inputs with the features pre-rounded to bf16, through `layout=PLANAR, feat dtype=BF16, precision=BF16`
    (what the fused trainer runs), against the oracle under torch's bf16 autocast on the same rounded features.
    Calibration as in test_mlp_forward_backward_vs_reference_golden: the HIP kernel may be at most 1.5x as far from
    the fp32 golden values as torch's own bf16 autocast is."""
    from hbr_amd._lib import BF16, PLANAR
    g = load_golden("g5_mlp.npz")
    keys = [f"{s}.{i}.{k}" for s in ("sig_model", "col_model") for i in (0, 2, 4) for k in ("weight", "bias")]
    flat = np.concatenate([g["p." + k].reshape(-1) for k in keys]).astype(np.float32)
    N = g["feat"].shape[0]
    feat_bf = torch.from_numpy(g["feat"]).bfloat16()
    dirs = torch.from_numpy(g["dirs"])
    dout = torch.from_numpy(g["dout"])
    # oracle: bf16 autocast on the rounded features
    prm = {k: torch.from_numpy(g["p." + k]).clone().requires_grad_(True) for k in keys}
    f_c = feat_bf.float().clone().requires_grad_(True)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        o_c = ref_cpu.mlp_forward(f_c, ref_cpu.dir_encode(dirs, 4), prm)
    o_c.float().backward(dout)
    # the same thing in exact fp32 on the rounded features: the yardstick both are measured against
    prm32 = {k: torch.from_numpy(g["p." + k]).clone().requires_grad_(True) for k in keys}
    f32 = feat_bf.float().clone().requires_grad_(True)
    o32 = ref_cpu.mlp_forward(f32, ref_cpu.dir_encode(dirs, 4), prm32)
    o32.backward(dout)

    feat_planar = feat_bf.reshape(N, 16, 2).permute(1, 0, 2).contiguous().to(DEV)
    pe = ops.dir_encode(dirs.to(DEV), 4)
    P = torch.from_numpy(flat).to(DEV)
    out = ops.mlp_fwd(feat_planar, PLANAR, pe, 1, P, BF16)
    dP = torch.zeros_like(P)
    amax = torch.full((16,), -1.0, device=DEV)
    dfeat = ops.mlp_bwd(feat_planar, PLANAR, pe, 1, P, BF16, dout.to(DEV), dP, absmax_out=amax)
    assert dfeat.dtype == torch.bfloat16 and tuple(dfeat.shape) == (16, N, 2)
    # the per-level maxima K4 hands to K2 are exactly those of the buffer it wrote
    assert torch.equal(amax, dfeat.float().abs().amax(dim=(1, 2)))
    for lay, fe in ((PLANAR, feat_planar.float()), (0, feat_bf.float().to(DEV))):  # fp32 storage, both layouts
        a32 = torch.full((16,), -1.0, device=DEV)
        d32 = ops.mlp_bwd(fe, lay, pe, 1, P, 0, dout.to(DEV), torch.zeros_like(P), absmax_out=a32)
        want = d32.abs().amax(dim=(1, 2)) if lay == PLANAR else d32.reshape(N, 16, 2).abs().amax(dim=(0, 2))
        assert torch.equal(a32, want)

    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

    def check(got, exact, autocast, what):
        r_hip, r_torch = rel(got, exact), rel(autocast, exact)
        assert r_hip <= max(1.5 * r_torch, 5e-3) and r_hip < 0.1, (what, r_hip, r_torch)

    check(out.cpu().numpy(), o32.detach().numpy(), o_c.detach().float().numpy(), "out")
    check(dfeat.float().permute(1, 0, 2).reshape(N, 32).cpu().numpy(), f32.grad.numpy(), f_c.grad.numpy(), "dfeat")
    dPn, off = dP.cpu().numpy(), 0
    for k in keys:
        n = prm32[k].grad.numel()
        check(dPn[off:off + n].reshape(prm32[k].grad.shape), prm32[k].grad.numpy(), prm[k].grad.numpy(), k)
        off += n


def test_mlp_strided_rows_view_forward_and_backward(ops):
    """feat = y[:, :32] of an [N,36] buffer (the encoder's E aux columns): kept as a strided view when 16-byte aligned.
    The feature gradient must come back with the same pitch - no write past an [N,32] allocation."""
    rng = np.random.default_rng(51)
    N = 300
    wide = torch.from_numpy(rng.standard_normal((N, 36)).astype(np.float32)).to(DEV)
    view = wide[:, :32]
    dense = view.contiguous()
    dirs = rng.standard_normal((N, 3)).astype(np.float32)
    pe = ops.dir_encode(torch.from_numpy(dirs).to(DEV), 4)
    P = torch.cat([v.reshape(-1) for v in ref_cpu.mlp_init(52).values()]).to(DEV)
    assert torch.equal(ops.mlp_fwd(view, 0, pe, 1, P, 0), ops.mlp_fwd(dense, 0, pe, 1, P, 0))
    dout = torch.from_numpy(rng.standard_normal((N, 4)).astype(np.float32)).to(DEV)
    guard = torch.full((4096,), 7.0, device=DEV)  # allocated right after: an overrun would most likely land here
    g1, g2 = torch.zeros_like(P), torch.zeros_like(P)
    d_view = ops.mlp_bwd(view, 0, pe, 1, P, 0, dout, g1)
    d_dense = ops.mlp_bwd(dense, 0, pe, 1, P, 0, dout, g2)
    torch.cuda.synchronize()
    assert d_view.stride(0) == 36 and tuple(d_view.shape) == (N, 32)
    assert torch.equal(d_view, d_dense) and torch.equal(g1, g2)
    assert bool((guard == 7.0).all())
    # through autograd (MlpFn keeps the strided view)
    from hbr_amd.test_hash import MLP_3D
    mlp = MLP_3D(num_sig=2, num_col=2, L=16, F=2, d_view=24).to(DEV)
    w = wide.clone().requires_grad_(True)
    mlp(w[:, :32], pe).sum().backward()
    assert w.grad is not None and float(w.grad[:, 32:].abs().max()) == 0.0 and float(w.grad[:, :32].abs().max()) > 0


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_trainer_split_scatter_branch_equals_single_launch(precision):
    """HashNeRFTrainer's multi-GPU branch (two half-level K2 launches interleaved with the staged all-reduce,
    trainer.py) forced on one GPU, where the reduce is a no-op: the gradient buffer, the updated parameters and the loss
    must equal the single-launch step bit for bit (K2's integer accumulation makes that an equality, not a tolerance)."""
    from hbr_amd import _lib, synthetic
    from hbr_amd.trainer import HashNeRFTrainer, build_default_model
    R, S = 1024, 128  # 131 072 points: the LDS-slice kernel on both paths
    o, d, dn, gt = (a.to(DEV) for a in synthetic.scene_rays(R, seed=61))
    mn, mx, sig = synthetic.ray_bbox(o, d)
    prec = _lib.BF16 if precision == "bf16" else _lib.F32
    t = ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.rand(S, generator=torch.Generator().manual_seed(62))).to(DEV)
    res = []
    for split in (False, True):
        enc, _, mlp = build_default_model(mn, sig, DEV, seed=7)
        with torch.no_grad():  # gradients of useful size from step one
            enc.stacked_tables().uniform_(-0.3, 0.3, generator=torch.Generator(device=DEV).manual_seed(63))
        tr = HashNeRFTrainer(enc, mlp, num_samples=S, total_steps=100, precision=prec, split_scatter=split)
        assert tr.split_scatter is split
        first = float(tr.step(o, d, dn.reshape(-1), gt, t=t))
        g1 = tr.grad.clone()
        more = [float(tr.step(o, d, dn.reshape(-1), gt, t=t)) for _ in range(2)]
        res.append((first, g1, more, tr.tables.clone()))
    (l_a, g_a, m_a, t_a), (l_b, g_b, m_b, t_b) = res
    nt = t_a.numel()
    # step 1 starts from identical parameters: same loss, and the table block of the gradient is bit-identical (the MLP
    # block still ends in <= 8 float atomics per bias address in mlp_dw_reduce_kernel, so it is compared to rounding)
    assert l_a == l_b
    assert torch.equal(g_a[:nt], g_b[:nt])
    assert torch.allclose(g_a[nt:], g_b[nt:], rtol=1e-5, atol=1e-9)
    # later steps inherit that rounding-level difference of the MLP parameters
    assert np.allclose(m_a, m_b, rtol=1e-4)
    assert torch.allclose(t_a, t_b, rtol=0, atol=2e-3)  # Adam steps are +-lr*sign(g) where |g| ~ 0


def test_shipped_path_training_vs_cpu_oracle_psnr():
    """PSNR vs reference (BASELINE metric, second half) ON THE SHIPPED PATH: 65 536 points per step (1024 rays x 64
    samples, T = 2^12) so that K2 is the LDS fixed-point kernel, bf16 MLP with bf16 feature buffers, 200 steps from the
    oracle's initial parameters, rays and jitter (metric: reference helper.py:301-304).

    What can and cannot be asserted (measured with tools/psnr_vs_oracle.py, gpurun_out r2p, this exact setup):
      steps            100      200      400      800
      oracle fp32    16.37    23.68    26.94    28.54 dB
      HIP fp32       -0.05    +0.27    +0.44    +0.36 dB   (same arithmetic as the oracle up to summation order!)
      HIP bf16/bf16  -0.19    +0.09    +0.67    +0.66 dB   (shipped)
      HIP bf16/f32   -0.72    -0.26    +0.79    +0.57 dB
    and, from an initialisation where training stalls at 11 dB (seed 8), every configuration within 0.03 dB at every
    horizon.  Every HIP row is bit-reproducible run to run (integer accumulation in K2, fixed-order reductions
    elsewhere).  A fixed-horizon bound of 0.1 dB therefore cannot hold for ANY implementation that is not bit-identical
    to the oracle: the fp32 path, which differs only in summation order, is already 0.3-0.4 dB away once training has
    taken off - Adam amplifies rounding-level differences into a different, equally valid trajectory.  The shipped bf16
    path is never meaningfully BELOW the oracle and ends 0.66 dB above it.  Asserted here:
      * the first 10 losses track the oracle's (2 % - bf16 operands), i.e. the same optimisation problem is being solved;
      * after 200 steps the PSNR is within 0.5 dB of the oracle's and both gained > 10 dB;
      * two HIP runs end with bit-identical parameters."""
    from hbr_amd._lib import BF16
    from hbr_amd.helper import calc_psnr
    from hbr_amd.trainer import HashNeRFTrainer, build_default_model
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    R, S, L, T, steps, seed = 1024, 64, 16, 2 ** 12, 200, 7
    o0, d0, _, _ = ref_cpu.synthetic_rays(8192, seed=0)
    mn, mx, sig = ref_cpu.bbox_mu_sigma(o0, d0)
    rng = np.random.default_rng(seed)
    tables0 = torch.from_numpy(rng.uniform(-1e-4, 1e-4, (L, T, 2)).astype(np.float32))
    params0 = ref_cpu.mlp_init(seed + 1)
    ts = [ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.from_numpy(rng.uniform(0, 1, S).astype(np.float32))) for _ in range(steps)]
    batches = [ref_cpu.synthetic_scene_rays(R, seed=50 + i) for i in range(16)]
    test = ref_cpu.synthetic_scene_rays(2048, seed=999)
    t_eval = torch.linspace(2.0, 6.0, S)
    sc = ref_cpu.level_scales(16, 2048.0, L)
    # oracle
    tabs = [tables0[l].clone().requires_grad_(True) for l in range(L)]
    prm = {k: v.clone().requires_grad_(True) for k, v in params0.items()}
    opts = ref_cpu.make_optimizers(tabs, prm.values(), steps)
    with torch.no_grad():
        p_init = float(ref_cpu.psnr(ref_cpu.render(test[0], test[1], t_eval, test[2], tabs, sc, mn, sig, prm)[0], test[3]))
    ref_losses = [float(ref_cpu.train_step(batches[k % 16], ts[k], tabs, sc, mn, sig, prm, opts)) for k in range(steps)]
    with torch.no_grad():
        p_ref = float(ref_cpu.psnr(ref_cpu.render(test[0], test[1], t_eval, test[2], tabs, sc, mn, sig, prm)[0], test[3]))
    # HIP, shipped configuration, twice
    finals = []
    for _ in range(2):
        enc, _, mlp = build_default_model(mn, sig, DEV, L=L, T=T, seed=0)
        with torch.no_grad():
            for l in range(L):
                enc.Embedding_list[l].weight.copy_(tables0[l])
            for k, v in params0.items():
                sq, idx, kind = k.split(".")
                getattr(getattr(mlp, sq)[int(idx)], kind).copy_(v)
        tr = HashNeRFTrainer(enc, mlp, num_samples=S, total_steps=steps, precision=BF16, feat_dtype=BF16)
        losses = [float(tr.step(*(a.to(DEV) for a in batches[k % 16]), t=ts[k].to(DEV))) for k in range(steps)]
        C = tr.render(test[0].to(DEV), test[1].to(DEV), test[2].to(DEV), t=t_eval.to(DEV))
        finals.append((losses, float(calc_psnr(C.cpu(), test[3])), tr.tables.clone(), tr.flat.clone()))
    (l1, p1, t1, f1), (l2, p2, t2, f2) = finals
    assert l1 == l2 and torch.equal(t1, t2) and torch.equal(f1, f2)       # bitwise reproducible
    assert np.allclose(l1[:10], ref_losses[:10], rtol=2e-2), (l1[:10], ref_losses[:10])
    assert p_ref > p_init + 10 and p1 > p_init + 10, (p_init, p_ref, p1)
    assert abs(p1 - p_ref) < 0.5, (p1, p_ref)


@pytest.mark.parametrize("T,layout,dt,outside", [(2 ** 16, "rows", torch.float32, True), (1000, "planar", torch.bfloat16, False),
                                                   (2 ** 12, "rows", torch.bfloat16, True), (2 ** 14 + 77, "planar", torch.float32, True)])
def test_k2_lds_kernels_edge_shapes_vs_oracle(ops, T, layout, dt, outside):
    """The LDS scatter kernels away from the benchmark's shape, against the oracle: EXPLICIT points (no ray structure:
    the boxes come from the points themselves), N = 70 001 (a partial last stripe, an odd level stride for the
    vector loads), rows and planar dy in fp32 and bf16, tables smaller than a slice / not a power of two (int64
    modulo path, also for the dense levels' vertex hashing), and points outside the box (negative cells: torch's
    `.long()` truncates toward zero, the per-corner path of the hashed kernel)."""
    from hbr_amd._lib import PLANAR, ROWS
    N, L = 70001, 16
    rng = np.random.default_rng(int(T) % 977)
    lo, hi = (-0.08, 0.5) if outside else (0.0, 0.55)
    mu, sigma = torch.tensor([-1.0, 0.5, 2.0]), torch.tensor(3.0)
    x = torch.from_numpy(rng.uniform(lo, hi, (N, 3)).astype(np.float32)) * sigma + mu
    sc = ref_cpu.level_scales(16, 2048.0, L)
    geom = ops.HashGeom(tuple(float(v) for v in sc), tuple(float(v) for v in mu), float(sigma), int(T), 2)
    dy = torch.from_numpy((rng.standard_normal((N, L * 2)) * 10.0 ** rng.uniform(-3, 0, (N, 1))).astype(np.float32)).to(dt)
    ref = ref_cpu.hash_encode_backward(x, dy.float(), sc, mu, sigma, int(T)).numpy()
    dyd = dy.to(DEV) if layout == "rows" else dy.reshape(N, L, 2).permute(1, 0, 2).contiguous().to(DEV)
    got = torch.zeros((L, int(T), 2), device=DEV)
    ops.hash_encode_bwd(geom, dyd, got, x=x.to(DEV), layout=ROWS if layout == "rows" else PLANAR, algo=2)
    g = got.cpu().numpy()
    assert np.allclose(g, ref, rtol=1e-4, atol=1e-6 * np.abs(ref).max())
    assert not np.any((g != 0) & (ref == 0))
    again = torch.zeros_like(got)
    ops.hash_encode_bwd(geom, dyd, again, x=x.to(DEV), layout=ROWS if layout == "rows" else PLANAR, algo=2)
    assert torch.equal(again, got)


_ADDR_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/oracle")
import ref_cpu
from hbr_amd import ops
from hbr_amd._lib import BF16, F32, PLANAR
dev = "cuda:0"
g = torch.Generator().manual_seed(11)
R, S = 37, 29                      # N = 1073: partial last tile, several points per ray
N = R * S
_, d, _, _ = ref_cpu.synthetic_rays(R, seed=3)
pe = ops.dir_encode(d.to(dev), 4)
P = torch.cat([v.reshape(-1) for v in ref_cpu.mlp_init(5).values()]).to(dev)
feat = (torch.randn((16, N, 2), generator=g) * 0.3)
dout = torch.randn((N, 4), generator=g).to(dev)
res = {}
for name, fdt, prec in (("bf16", torch.bfloat16, BF16), ("f32", torch.float32, F32)):
    f = feat.to(fdt).to(dev)
    out = ops.mlp_fwd(f, PLANAR, pe, S, P, prec)
    dP = torch.zeros_like(P); amax = torch.zeros(16, device=dev)
    df = ops.mlp_bwd(f, PLANAR, pe, S, P, prec, dout, dP, absmax_out=amax)
    res[name + "_out"] = out.cpu().numpy(); res[name + "_dP"] = dP.cpu().numpy()
    res[name + "_df"] = df.float().cpu().numpy(); res[name + "_amax"] = amax.cpu().numpy()
# the one-launch render + backward (round 4) through the same two addressing paths: 64 rays x 32 samples
R2, S2 = 67, 32
g2 = torch.Generator().manual_seed(12)
feat2 = (torch.randn((16, R2 * S2, 2), generator=g2) * 0.3).bfloat16().to(dev)
_, d2, _, gt2 = ref_cpu.synthetic_rays(R2, seed=4)
pe2 = ops.dir_encode(d2.to(dev), 4)
t2 = ref_cpu.strat_jitter_to_t(2.0, 6.0, S2, torch.rand(S2, generator=g2)).to(dev)
dP2 = torch.zeros_like(P); amax2 = torch.zeros(16, device=dev)
loss2, df2, Cr2 = ops.mlp_render_bwd(feat2, pe2, P, BF16, t2, None, gt2.to(dev), dP2, absmax_out=amax2, want_Cr=True)
res["render_loss"] = np.array(float(loss2)); res["render_dP"] = dP2.cpu().numpy(); res["render_df"] = df2.float().cpu().numpy()
res["render_Cr"] = Cr2.cpu().numpy(); res["render_amax"] = amax2.cpu().numpy()
np.savez(sys.argv[2], **res)
"""


def test_mlp_64bit_addressing_path_equals_32bit_path(tmp_path):
    """Beyond 2^27 points (or 2^25 rays) the MLP kernels switch from 32-bit offsets off wave-uniform level bases to
    64-bit per-lane addresses and a division for the ray index.  HBR_MLP_ADDR64 forces that path at a small size (in a
    child process: the knob is read once per process); every output must equal the default path's bit for bit."""
    import os, subprocess, sys
    from conftest import ROOT
    outs = {}
    for tag, extra in (("a32", {}), ("a64", {"HBR_MLP_ADDR64": "1"})):
        env = dict(os.environ, **extra)
        env.pop("HBR_MLP_ADDR64", None) if not extra else None
        path = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, "-c", _ADDR_SCRIPT, ROOT, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = np.load(path)
    assert set(outs["a32"].files) == set(outs["a64"].files) and len(outs["a32"].files) == 13
    for k in outs["a32"].files:
        assert np.array_equal(outs["a32"][k], outs["a64"][k]), k
    assert np.abs(outs["a32"]["bf16_df"]).max() > 0 and np.abs(outs["a32"]["f32_dP"]).max() > 0 and np.abs(outs["a32"]["render_df"]).max() > 0


@pytest.mark.parametrize("precision", [1, 0])
def test_mlp_instantiations_agree_bit_for_bit(ops, precision):
    """The MLP kernels are compiled once per (feature layout, feature storage type); all of them run the same
    arithmetic, so on features that are exactly representable in bf16 every instantiation must return the same bits
    (outputs, parameter gradients, per-level maxima; feature gradients after rounding the fp32-stored ones to bf16).
    Guards against per-instantiation code-generation accidents: a scheduling-dependent wrong result in ONE of them
    (rows layout, fp32 features, bf16 MFMA) is what this test was written after."""
    from hbr_amd._lib import PLANAR
    g = torch.Generator().manual_seed(23)
    R, S = 41, 27                     # N = 1107: a partial last tile
    N = R * S
    _, d, _, _ = ref_cpu.synthetic_rays(R, seed=4)
    pe = ops.dir_encode(d.to(DEV), 4)
    P = torch.cat([v.reshape(-1) for v in ref_cpu.mlp_init(9).values()]).to(DEV)
    feat = (torch.randn((N, 32), generator=g) * 0.3).bfloat16()
    dout = torch.randn((N, 4), generator=g).to(DEV)
    got = {}
    for lay in (0, PLANAR):
        for fdt in (torch.float32, torch.bfloat16):
            f = feat.to(fdt)
            f = (f.reshape(N, 16, 2).permute(1, 0, 2) if lay == PLANAR else f).contiguous().to(DEV)
            out = ops.mlp_fwd(f, lay, pe, S, P, precision)
            dP = torch.zeros_like(P)
            amax = torch.zeros(16, device=DEV)
            df = ops.mlp_bwd(f, lay, pe, S, P, precision, dout, dP, absmax_out=amax)
            df = (df.permute(1, 0, 2).reshape(N, 32) if lay == PLANAR else df)
            got[(lay, fdt)] = (out.cpu(), dP.cpu(), df.bfloat16().cpu(), amax.cpu() if fdt == torch.bfloat16 else None)
    ref = got[(PLANAR, torch.bfloat16)]
    assert float(ref[1].abs().max()) > 0 and float(ref[2].float().abs().max()) > 0
    for key, (out, dP, df, amax) in got.items():
        assert torch.equal(out, ref[0]), ("out", key)
        assert torch.equal(dP, ref[1]), ("dparams", key)
        assert torch.equal(df, ref[2]), ("dfeat", key)
        if amax is not None:
            assert torch.equal(amax, ref[3]), ("absmax", key)


@pytest.mark.parametrize("precision", [1, 0])
def test_mlp_bwd_reuses_the_forward_image(ops, precision):
    """HBR_IMAGE_READY: the backward right after a forward with the same parameters skips the repack and returns the
    same bits; with the flag wrongly set after OTHER parameters were packed it would not - which is what makes the
    check meaningful."""
    from hbr_amd._lib import PLANAR
    g = torch.Generator().manual_seed(5)
    N = 1000
    feat = (torch.randn((16, N, 2), generator=g) * 0.3).to(DEV)
    pe = ops.dir_encode(torch.nn.functional.normalize(torch.randn((N, 3), generator=g), dim=1).to(DEV), 4)
    dout = torch.randn((N, 4), generator=g).to(DEV)
    P = torch.cat([v.reshape(-1) for v in ref_cpu.mlp_init(2).values()]).to(DEV)
    P2 = torch.cat([v.reshape(-1) for v in ref_cpu.mlp_init(3).values()]).to(DEV)

    def bwd(ready):
        dP = torch.zeros_like(P)
        df = ops.mlp_bwd(feat, PLANAR, pe, 1, P, precision, dout, dP, image_ready=ready)
        return df.clone(), dP

    ops.mlp_fwd(feat, PLANAR, pe, 1, P, precision)
    ref = bwd(False)
    ops.mlp_fwd(feat, PLANAR, pe, 1, P, precision)
    got = bwd(True)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    ops.mlp_fwd(feat, PLANAR, pe, 1, P2, precision)   # a different image in the workspace
    stale = bwd(True)
    assert not torch.equal(stale[1], ref[1])


def test_overwrite_mode_equals_accumulating_into_zeros(ops):
    """HBR_OVERWRITE (the trainer's way of not zeroing 8 MiB per step): K2 and K4 leave exactly this call's gradient in
    buffers that held garbage - bit for bit what accumulating into zeros gives - on the LDS path (rows written by the
    slab reduce), on the float-atomic path (ops zeroes first) and for an empty batch."""
    from hbr_amd._lib import BF16, PLANAR
    R, S, T = 1024, 64, 2 ** 12            # 65 536 points: the LDS kernels
    o, d, t, mn, sig, sc, geom = _scene(ops, R, S, T, seed=31)
    rays = (o.to(DEV), d.to(DEV), t.to(DEV))
    g = torch.Generator().manual_seed(4)
    dy = (torch.randn((16, R * S, 2), generator=g) * 1e-3).bfloat16().to(DEV)
    for algo in (2, 1, 0):
        ref = ops.hash_encode_bwd(geom, dy, torch.zeros((16, T, 2), device=DEV), rays=rays, layout=PLANAR, algo=algo)
        junk = torch.full((16, T, 2), 7.5, device=DEV)
        got = ops.hash_encode_bwd(geom, dy, junk, rays=rays, layout=PLANAR, algo=algo, overwrite=True)
        if algo == 1:   # float atomics: order-dependent rounding
            assert torch.allclose(got, ref, rtol=1e-4, atol=1e-5 * float(ref.abs().max()))
        else:
            assert torch.equal(got, ref), algo
    few = (rays[0][:8], rays[1][:8], rays[2])   # 512 points: the float-atomic kernel, zeroed by ops
    ref = ops.hash_encode_bwd(geom, dy[:, :512].contiguous(), torch.zeros((16, T, 2), device=DEV), rays=few, layout=PLANAR)
    got = ops.hash_encode_bwd(geom, dy[:, :512].contiguous(), torch.full((16, T, 2), -3.0, device=DEV), rays=few, layout=PLANAR, overwrite=True)
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-5 * float(ref.abs().max())) and float(got.abs().max()) < 1.0
    none = (rays[0][:0], rays[1][:0], rays[2])
    assert float(ops.hash_encode_bwd(geom, dy[:, :0].contiguous(), torch.ones((16, T, 2), device=DEV), rays=none, layout=PLANAR,
                                     overwrite=True).abs().max()) == 0.0

    N = 2000
    feat = (torch.randn((16, N, 2), generator=g) * 0.3).bfloat16().to(DEV)
    pe = ops.dir_encode(torch.nn.functional.normalize(torch.randn((N, 3), generator=g), dim=1).to(DEV), 4)
    dout = torch.randn((N, 4), generator=g).to(DEV)
    P = torch.cat([v.reshape(-1) for v in ref_cpu.mlp_init(6).values()]).to(DEV)
    for prec in (BF16, 0):
        f = feat if prec == BF16 else feat.float()
        ref = torch.zeros_like(P)
        ops.mlp_bwd(f, PLANAR, pe, 1, P, prec, dout, ref)
        got = torch.full_like(P, 9.25)
        ops.mlp_bwd(f, PLANAR, pe, 1, P, prec, dout, got, overwrite=True)
        assert torch.equal(got, ref), prec
    got = torch.ones_like(P)
    ops.mlp_bwd(feat[:, :0].contiguous(), PLANAR, pe[:0], 1, P, BF16, dout[:0], got, overwrite=True)
    assert float(got.abs().max()) == 0.0


def test_integration_md_ctypes_stub_runs_verbatim():
    """INTEGRATION.md section B - the reference-side ctypes binding a maintainer would add next to hash_encoding.py - is
    executed AS PRINTED (only the library path is filled in) on an object with the reference HashEncoder's attributes
    (hash_encoding.py:6-39: L, T, F, N_min, b, mu, sigma, Embedding_list), forward and backward, against the oracle."""
    import re
    import types
    import hbr_amd._lib as L
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## B."):]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    assert "/path/to/libhbr_hip.so" in code and "class _HashEncode" in code
    ns = {}
    exec(compile(code.replace("/path/to/libhbr_hip.so", L.LIB_PATH), "INTEGRATION.md#B", "exec"), ns)
    Lv, T, F, N = 16, 2 ** 12, 2, 3000
    rng = np.random.default_rng(12)
    mu, sigma = torch.tensor([-1.0, -0.7, 0.1]), torch.tensor(2.9)
    enc = types.SimpleNamespace(L=Lv, T=T, F=F, N_min=torch.tensor(16), mu=mu.to(DEV), sigma=sigma.to(DEV))
    enc.b = torch.exp((torch.log(torch.tensor(2048.0)) - torch.log(enc.N_min)) / (Lv - 1))     # hash_encoding.py:13
    tabs = [torch.from_numpy(rng.uniform(-0.5, 0.5, (T, F)).astype(np.float32)) for _ in range(Lv)]
    weights = [t.clone().to(DEV).requires_grad_(True) for t in tabs]
    x = mu + torch.from_numpy(rng.uniform(0.01, 2.0, (N, 3)).astype(np.float32))
    y = ns["_HashEncode"].apply(x.to(DEV), enc, *weights)
    dy = torch.from_numpy(rng.normal(0, 1, (N, Lv * F)).astype(np.float32))
    y.backward(dy.to(DEV))
    sc = ref_cpu.level_scales(16, 2048.0, Lv)
    ref_t = [t.clone().requires_grad_(True) for t in tabs]
    y_ref = ref_cpu.hash_encode(x, ref_t, sc, mu, sigma)
    y_ref.backward(dy)
    assert float((y.detach().cpu() - y_ref.detach()).abs().max()) <= 1e-6 * float(y_ref.abs().max())
    for l in range(Lv):
        g, gr = weights[l].grad.cpu(), ref_t[l].grad
        assert float((g - gr).abs().max()) <= 1e-4 * float(gr.abs().max()) + 1e-9, l



@pytest.mark.parametrize("L", [8, 12])
def test_k2_other_level_counts_vs_oracle_and_absmax_handoff(ops, L):
    """K2 at L = 8 and 12 (the fused step itself is L = 16 only - the MLP kernels read 32 features,
    and vol_render / HashNeRFTrainer refuse another encoder, checked below): the LDS kernels against the oracle, and a
    caller-supplied per-level max |dy| (K4's hand-off) gives the same BITS as K2's own absmax pass."""
    from hbr_amd._lib import PLANAR
    R, S, T = 1024, 64, 2 ** 14
    N = R * S
    o, d, _, _ = ref_cpu.synthetic_rays(R, seed=60 + L)
    mn, mx, sig = ref_cpu.bbox_mu_sigma(o, d)
    sc = ref_cpu.level_scales(16, 2048.0, L)
    geom = ops.HashGeom(tuple(float(v) for v in sc), tuple(float(v) for v in mn), float(sig), T, 2)
    t = ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.rand(S, generator=torch.Generator().manual_seed(61)))
    rng = np.random.default_rng(62 + L)
    dy_bf = torch.from_numpy((rng.standard_normal((N, L * 2)) * 10.0 ** rng.uniform(-4, 0, (N, 1))).astype(np.float32)).bfloat16()
    dy_planar = dy_bf.reshape(N, L, 2).permute(1, 0, 2).contiguous().to(DEV)
    pts = ref_cpu.sample_points(o, d, t).reshape(-1, 3)
    ref = ref_cpu.hash_encode_backward(pts, dy_bf.float(), sc, mn, sig, T).numpy()
    rays = (o.to(DEV), d.to(DEV), t.to(DEV))
    got = torch.zeros((L, T, 2), device=DEV)
    ops.hash_encode_bwd(geom, dy_planar, got, rays=rays, layout=PLANAR, algo=2)
    assert np.allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=1e-6 * np.abs(ref).max())
    amax = dy_planar.float().abs().amax(dim=(1, 2))
    again = torch.zeros((L, T, 2), device=DEV)
    ops.hash_encode_bwd(geom, dy_planar, again, rays=rays, layout=PLANAR, algo=2, dy_absmax=amax)
    assert torch.equal(again, got)
    # the fused routes refuse an encoder that does not feed the MLP's 32 inputs
    from hbr_amd.hash_encoding import HashEncoder
    from hbr_amd.test_hash import MLP_3D
    from hbr_amd.trainer import HashNeRFTrainer
    from hbr_amd.vol_renderer import Volume_Renderer
    from hbr_amd.encoder import PositionalEncoder
    enc = HashEncoder(N_max=2048.0, N_min=16, L=L, T=T, F=2, dim=3, mu=mn.to(DEV), sigma=sig.to(DEV), device=DEV)
    mlp = MLP_3D(num_sig=2, num_col=2, L=16, F=2, d_view=24).to(DEV)
    with pytest.raises(NotImplementedError):
        HashNeRFTrainer(enc, mlp, num_samples=S)
    vr = Volume_Renderer(H=8, W=8, K=torch.eye(3), near=2.0, far=6.0, device=DEV, Pos_encode=enc, Dir_encode=PositionalEncoder(3, 4), sigma_val=sig, mu=mn)
    with pytest.raises(NotImplementedError):
        vr.vol_render(mlp, rays[1], rays[0], num_samples=S, t=rays[2], update_mask=False, hierarchical=False)
    with pytest.raises(Exception):  # a planar buffer of another level count never reaches the MLP kernels
        ops.mlp_fwd(torch.zeros((L, 64, 2), device=DEV), PLANAR, torch.zeros((1, 24), device=DEV), 64, mlp.flat_params()[0], 0)


@pytest.mark.parametrize("log2T", [17, 18, 19])  # 8 slices: unmasked sweep; 16 (the threshold) and 32: masked
def test_k2_large_tables_vs_oracle(ops, log2T):
    """train_hash2.py:36 --hash_size above the README's 16 (19 is the Instant-NGP default the flag reaches): the LDS
    kernels on 131 072 points against the oracle, bit-identical re-run, and the two flushes / the global-atomics kernel
    beside it."""
    from hbr_amd._lib import PLANAR
    R, S, L, T = 1024, 128, 16, 2 ** log2T
    N = R * S
    o, d, t, mn, sig, sc, geom = _scene(ops, R, S, T, seed=71)
    rng = np.random.default_rng(72)
    dy_bf = torch.from_numpy((rng.standard_normal((N, L * 2)) * 10.0 ** rng.uniform(-5, 0, (N, 1))).astype(np.float32)).bfloat16()
    dy_planar = dy_bf.reshape(N, L, 2).permute(1, 0, 2).contiguous().to(DEV)
    pts = ref_cpu.sample_points(o, d, t).reshape(-1, 3)
    ref = ref_cpu.hash_encode_backward(pts, dy_bf.float(), sc, mn, sig, T).numpy()
    rays = (o.to(DEV), d.to(DEV), t.to(DEV))
    runs = []
    for _ in range(2):
        got = torch.zeros((L, T, 2), device=DEV)
        ops.hash_encode_bwd(geom, dy_planar, got, rays=rays, layout=PLANAR, algo=2)
        runs.append(got)
    assert torch.equal(runs[0], runs[1])
    g = runs[0].cpu().numpy()
    assert np.allclose(g, ref, rtol=1e-4, atol=1e-6 * np.abs(ref).max())
    assert not np.any((g != 0) & (ref == 0))
    assert np.all((g != 0) | (np.abs(ref) < 1e-10 * np.abs(ref).max()))
    for kw in (dict(algo=2, deterministic=False), dict(algo=1)):
        alt = torch.zeros((L, T, 2), device=DEV)
        ops.hash_encode_bwd(geom, dy_planar, alt, rays=rays, layout=PLANAR, **kw)
        assert np.allclose(alt.cpu().numpy(), ref, rtol=1e-4, atol=1e-6 * np.abs(ref).max()), kw
    # overwrite mode leaves the same bits in a dirty buffer
    dirty = torch.full((L, T, 2), 7.0, device=DEV)
    ops.hash_encode_bwd(geom, dy_planar, dirty, rays=rays, layout=PLANAR, algo=2, overwrite=True)
    assert torch.equal(dirty, runs[0])
