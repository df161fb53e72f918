"""Generate tests/golden/*.npz by running the REFERENCE's own library modules on CPU.

TEST INFRASTRUCTURE ONLY; runs in the build container (needs /root/reference, which never
travels to the GPU box).  The committed .npz files are data: seeded inputs + the reference's
outputs.  No reference source is copied.

Harness-side shims (no edits to the reference):
  1. hash_encoding.py:24 builds np.array([1, 2654435761, ...], dtype=np.int32), which raises
     OverflowError on numpy>=2 (the pinned numpy 1.23 wrapped silently).  We hand the module a
     numpy proxy whose `array(..., dtype=int32)` wraps via int64.
  2. h5py / cv2 are imported but unused on the path -> empty stub modules.
  3. test_hash.py:25-26 calls `.to('cuda')` on max_bound/min_bound -> an object whose .to()
     returns the CPU tensor.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [--ref /root/reference]
"""
from __future__ import annotations

import argparse
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


class _NumpyProxy:
    def __init__(self, real):
        self._real = real

    def __getattr__(self, k):
        return getattr(self._real, k)

    def array(self, obj, dtype=None, **kw):
        if dtype is self._real.int32:
            return self._real.array(obj, dtype=self._real.int64).astype(self._real.int32)
        return self._real.array(obj, dtype=dtype, **kw)


class _CpuBound:
    def __init__(self, t):
        self.t = t

    def to(self, *_a, **_k):
        return self.t


def import_reference(ref_dir: str):
    sys.dont_write_bytecode = True
    for name in ("h5py", "cv2"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, ref_dir)
    with contextlib.redirect_stdout(io.StringIO()):
        import hash_encoding
        hash_encoding.np = _NumpyProxy(np)
        import encoder
        import test_hash
        import helper
        import vol_renderer
    return types.SimpleNamespace(hash_encoding=hash_encoding, encoder=encoder, test_hash=test_hash,
                                 helper=helper, vol_renderer=vol_renderer)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def make_tables(rng, L, T, F, scale=1e-4):
    return f32(rng.uniform(-scale, scale, (L, T, F)))


def build_encoder(ref, tables, N_max, N_min, mu, sigma):
    L, T, F = tables.shape
    enc = quiet(ref.hash_encoding.HashEncoder, N_max=N_max, N_min=N_min, L=L, T=T, F=F, dim=3,
                mu=torch.from_numpy(mu), sigma=torch.tensor(float(sigma)), device="cpu")
    with torch.no_grad():
        for l in range(L):
            enc.Embedding_list[l].weight.copy_(torch.from_numpy(tables[l]))
    return enc


def build_mlp(ref, params):
    one = _CpuBound(torch.ones(3))
    m = ref.test_hash.MLP_3D(num_sig=2, num_col=2, L=16, F=2, d_view=24, max_bound=one, min_bound=one)
    with torch.no_grad():
        for k, v in params.items():
            seq, idx, kind = k.split(".")
            getattr(getattr(m, seq)[int(idx)], kind).copy_(v)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    ref = import_reference(args.ref)
    sys.path.insert(0, HERE)
    import ref_cpu  # only for mlp_init (numpy-RNG weights) and synthetic inputs
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)

    # ---------------- G2: level scales --------------------------------------------------
    g2 = {}
    for tag, (nmax, nmin, L) in {"f2048_16": (2048.0, 16, 16), "i2048_16": (2048, 16, 16),
                                 "f512_8": (512.0, 16, 8), "f4096_16": (4096.0, 16, 16)}.items():
        enc = quiet(ref.hash_encoding.HashEncoder, N_max=nmax, N_min=nmin, L=L, T=16, F=2, dim=3, device="cpu")
        g2[tag] = np.stack([(enc.N_min * enc.b ** i).to(torch.float32).numpy() for i in range(L)])
    np.savez_compressed(os.path.join(OUT, "g2_level_scales.npz"), **g2)

    # ---------------- G1: integer corner ids + hash KAT ---------------------------------
    rng = np.random.default_rng(101)
    enc = quiet(ref.hash_encoding.HashEncoder, N_max=2048.0, N_min=16, L=16, T=2 ** 16, F=2, dim=3, device="cpu")
    coords = rng.integers(-40, 2100, size=(256, 3)).astype(np.int64)
    coords[:8] = [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, -1, -1], [2047, 2047, 2047], [-3, 5, -7], [65536, 1, 2]]
    idx = torch.from_numpy(coords)[:, None, :]  # [N,1,3] -> broadcast against pis[None,None,:]
    kat = {"coords": coords}
    for T in (2 ** 16, 2 ** 19, 2 ** 10, 1000, 92681):
        kat[f"hash_T{T}"] = enc.hash_func(idx, T)[:, 0].numpy().astype(np.int64)
    # corner ordering: run fast_get_indices on x0/x1 pairs
    x0 = torch.from_numpy(coords[:16])
    x_val = torch.stack([x0, x0 + 1], dim=-1)[..., None, :, :]
    bm = enc.bin_mask.reshape(1, 8, 3)
    kat["corner_ids"] = torch.where(bm, x_val[..., 0], x_val[..., 1]).numpy()
    np.savez_compressed(os.path.join(OUT, "g1_hash_kat.npz"), **kat)

    # ---------------- G3: encoder fwd + table grads -------------------------------------
    def encoder_case(tag, L, T, F, N, seed, N_max=2048.0):
        rng = np.random.default_rng(seed)
        mu = f32([-1.5, -1.25, -1.0])
        sigma = np.float32(5.5)
        tables = make_tables(rng, L, T, F, scale=1.0)  # O(1) values so errors are visible
        x = f32(rng.uniform(-1.4, 1.9, (N, 3)))
        x[0] = mu  # exactly at the origin of the grid
        x[1] = mu + np.float32(5.5) * np.float32(0.25)  # on cell boundaries at coarse levels
        x[2] = mu - np.float32(0.3)  # below mu: negative cells, trunc toward zero
        x[3] = [0.0, 0.0, 0.0]
        enc = build_encoder(ref, tables, N_max, 16, mu, sigma)
        xt = torch.from_numpy(x)
        y = enc(xt)
        dy = f32(rng.standard_normal(y.shape))
        y.backward(torch.from_numpy(dy))
        grads = np.stack([enc.Embedding_list[l].weight.grad.numpy() for l in range(L)])
        out = dict(x=x, mu=mu, sigma=sigma, y=y.detach().numpy(), dy=dy, L=L, T=T, F=F, N_max=N_max, seed=seed)
        if tables.nbytes <= (1 << 20):
            out["tables"] = tables
            out["dtables"] = grads
        else:  # big table: regenerate tables from the seed in the test; store grads sparsely
            nz = np.nonzero(np.abs(grads).sum(-1))
            out["dtab_l"] = nz[0].astype(np.int32)
            out["dtab_row"] = nz[1].astype(np.int32)
            out["dtab_val"] = grads[nz]
        np.savez_compressed(os.path.join(OUT, f"g3_encoder_{tag}.npz"), **out)

    encoder_case("T10", 16, 2 ** 10, 2, 2048, 303)
    encoder_case("T16", 16, 2 ** 16, 2, 384, 304)
    encoder_case("T1000", 4, 1000, 2, 512, 305, N_max=512.0)  # non power-of-two table size

    # ---------------- G4: direction encoding -------------------------------------------
    rng = np.random.default_rng(404)
    d = f32(rng.standard_normal((64, 3)))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    pe = quiet(ref.encoder.PositionalEncoder, d_model=3, num_freq=4)
    pe.sinus_in = pe.sinus_in.cpu()
    pe10 = quiet(ref.encoder.PositionalEncoder, d_model=3, num_freq=10)
    pe10.sinus_in = pe10.sinus_in.cpu()
    np.savez_compressed(os.path.join(OUT, "g4_dir_pe.npz"), d=d, pe4=pe(torch.from_numpy(d)).numpy(),
                        pe10=pe10(torch.from_numpy(d)).numpy())

    # ---------------- G5: MLP_3D fwd + all grads ---------------------------------------
    rng = np.random.default_rng(505)
    params = ref_cpu.mlp_init(505)
    mlp = build_mlp(ref, params)
    N = 1024
    feat = f32(rng.standard_normal((N, 32)) * 0.5)
    dirs = f32(rng.standard_normal((N, 3)))
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    dirs_enc = pe(torch.from_numpy(dirs))
    ft = torch.from_numpy(feat).requires_grad_(True)
    out = quiet(mlp, ft, dirs_enc)
    dout = f32(rng.standard_normal(out.shape))
    out.backward(torch.from_numpy(dout))
    g5 = dict(feat=feat, dirs=dirs, out=out.detach().numpy(), dout=dout, dfeat=ft.grad.numpy())
    for k, v in params.items():
        g5["p." + k] = v.numpy()
    for name, p in mlp.named_parameters():
        g5["g." + name] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g5_mlp.npz"), **g5)

    # ---------------- G6: calc_color fwd/bwd -------------------------------------------
    rng = np.random.default_rng(606)
    R, S = 48, 40
    t = np.sort(f32(rng.uniform(2.0, 6.2, S)))
    sigma = f32(rng.standard_normal((R, S)) * 4.0)
    sigma[0, :] = 0.0
    sigma[1, 3] = -20.0; sigma[1, 4] = -10.0; sigma[1, 5] = -10.5  # clamp edge
    sigma[2, :] = np.abs(sigma[2, :]) * 10  # opaque ray
    sigma[3, :] = -np.abs(sigma[3, :])  # all negative
    rgb = f32(rng.uniform(-0.5, 1.2, (R, S, 3)))
    dn = f32(rng.uniform(0.9, 1.3, (R, 1)))
    st = torch.from_numpy(sigma.copy()).requires_grad_(True)
    rt = torch.from_numpy(rgb).requires_grad_(True)
    # calc_color clamps in place on its argument -> hand it a non-leaf view like vol_render does
    Cr, wts, _ = ref.helper.calc_color(torch.from_numpy(t), rt, st * 1.0, torch.from_numpy(dn), device="cpu")
    dC = f32(rng.standard_normal((R, 3)))
    Cr.backward(torch.from_numpy(dC))
    np.savez_compressed(os.path.join(OUT, "g6_composite.npz"), t=t, sigma=sigma, rgb=rgb, dir_norm=dn,
                        Cr=Cr.detach().numpy(), wts=wts.detach().numpy(), dC=dC,
                        dsigma=st.grad.numpy(), drgb=rt.grad.numpy())
    # scalar dir_norm (vol_render default dir_norm=1)
    Cr1, w1, _ = ref.helper.calc_color(torch.from_numpy(t), torch.from_numpy(rgb), torch.from_numpy(sigma.copy()), 1, device="cpu")
    np.savez_compressed(os.path.join(OUT, "g6b_composite_scalar_norm.npz"), Cr=Cr1.numpy(), wts=w1.numpy())

    # ---------------- G7: get_od + strat_sampler ---------------------------------------
    rng = np.random.default_rng(707)
    H, W = 12, 20
    K = torch.tensor([[30.5, 0.0, 9.75], [0.0, 28.25, 6.5], [0.0, 0.0, 1.0]])
    A = rng.standard_normal((2, 3, 3))
    Q = np.stack([np.linalg.qr(a)[0] for a in A])
    c2w = np.zeros((2, 4, 4), np.float32)
    c2w[:, :3, :3] = Q
    c2w[:, :3, 3] = rng.uniform(-4, 4, (2, 3))
    c2w[:, 3, 3] = 1
    o, dd, nn = ref.helper.get_od(H, W, K, torch.from_numpy(c2w))
    torch.manual_seed(7)
    u = torch.rand(16)
    torch.manual_seed(7)
    tt = ref.helper.strat_sampler(torch.tensor(2.0), torch.tensor(6.0), 16, device="cpu")
    np.savez_compressed(os.path.join(OUT, "g7_rays.npz"), H=H, W=W, K=K.numpy(), c2w=c2w, o=o.numpy(),
                        d=dd.numpy(), n=nn.numpy(), strat_u=u.numpy(), strat_t=tt.numpy())

    # ---------------- G8: full vol_render + G9 one optimiser step ----------------------
    rng = np.random.default_rng(808)
    R, S, L, T, F = 64, 32, 16, 2 ** 11, 2
    o, dvec, dn, gt = ref_cpu.synthetic_rays(R, seed=808)
    mn, mx, sig = ref_cpu.bbox_mu_sigma(o, dvec, 2.0, 6.0)
    mu = mn.numpy()
    sigma = np.float32(sig.item())
    tables = make_tables(rng, L, T, F, scale=0.5)
    params = ref_cpu.mlp_init(809)
    t = f32(np.linspace(2.0, 6.0, S) + rng.uniform(0, 1, S) * 4.0 / S)
    enc = build_encoder(ref, tables, 2048.0, 16, mu, sigma)
    mlp = torch.nn.DataParallel(build_mlp(ref, params))
    Kd = torch.eye(3)
    vr = ref.vol_renderer.Volume_Renderer(H=8, W=8, K=Kd, near=2.0, far=6.0, device="cpu", Pos_encode=enc,
                                          Dir_encode=pe, max_dim=2 ** 10, sigma_val=torch.tensor(float(sigma)),
                                          mu=torch.from_numpy(mu))
    # capture sigma/rgb handed to calc_color
    cap = {}
    orig_cc = ref.vol_renderer.calc_color

    def spy(**kw):  # (`cap` is looked up at call time: G13 rebinds it)
        cap["sigma"] = kw["sigma"].detach().clone().numpy()
        cap["rgb"] = kw["rgb"].detach().clone().numpy()
        return orig_cc(**kw)

    ref.vol_renderer.calc_color = spy
    total_steps = 10
    oe = torch.optim.Adam(list(enc.Embedding_list.parameters()), lr=0.05)
    om = torch.optim.AdamW(mlp.parameters(), lr=0.005)
    se = torch.optim.lr_scheduler.CosineAnnealingLR(oe, T_max=total_steps, eta_min=1e-4)
    sm = torch.optim.lr_scheduler.CosineAnnealingLR(om, T_max=total_steps, eta_min=1e-4)
    crit = torch.nn.MSELoss()
    Cr, Cf, _ = quiet(vr.vol_render, mlp, dvec, o, num_samples=S, t=torch.from_numpy(t), update_mask=False,
                      dir_norm=dn, hierarchical=False)
    ref.vol_renderer.calc_color = orig_cc
    loss = crit(Cr, gt) + crit(Cf, gt)
    loss.backward()
    g8 = dict(o=o.numpy(), d=dvec.numpy(), dir_norm=dn.numpy(), gt=gt.numpy(), t=t, mu=mu, sigma=sigma,
              tables=tables, Cr=Cr.detach().numpy(), sig_out=cap["sigma"], rgb_out=cap["rgb"], loss=loss.item(),
              L=L, T=T, F=F, total_steps=total_steps)
    for k, v in params.items():
        g8["p." + k] = v.numpy()
    g8["dtables"] = np.stack([enc.Embedding_list[l].weight.grad.numpy() for l in range(L)])
    for name, p in mlp.module.named_parameters():
        g8["g." + name] = p.grad.numpy()
    oe.step(); om.step(); se.step(); sm.step()
    g8["tables_after"] = np.stack([enc.Embedding_list[l].weight.detach().numpy() for l in range(L)])
    for name, p in mlp.module.named_parameters():
        g8["a." + name] = p.detach().numpy()
    g8["lr_embed_after"] = se.get_last_lr()[0]
    g8["lr_mlp_after"] = sm.get_last_lr()[0]
    # unmasked render branch (update_mask=True, vol_renderer.py:199-208) on the updated weights
    with torch.no_grad():
        _, Cf2, _ = quiet(vr.vol_render, mlp, dvec, o, num_samples=S, t=torch.from_numpy(t), update_mask=True,
                          dir_norm=dn, hierarchical=False)
    g8["Cr_after_unmasked"] = Cf2.numpy()
    g8["state_keys_mlp"] = np.array(list(mlp.state_dict().keys()))
    g8["state_keys_enc"] = np.array(list(enc.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, "g8_render_step.npz"), **g8)

    # ---------------- G11/G12: hierarchical second pass (helper.py:23-51, vol_renderer.py:225-242) ----------
    # The reference draws u = rand(R,S) then samples = rand(S) from torch's global CPU generator; seeding it and
    # replaying the two draws gives the explicit random inputs the build's implementation takes as arguments.
    enc = build_encoder(ref, tables, 2048.0, 16, mu, sigma)          # original (pre-step) weights
    mlp = torch.nn.DataParallel(build_mlp(ref, params))
    vr = ref.vol_renderer.Volume_Renderer(H=8, W=8, K=Kd, near=2.0, far=6.0, device="cpu", Pos_encode=enc,
                                          Dir_encode=pe, max_dim=2 ** 10, sigma_val=torch.tensor(float(sigma)),
                                          mu=torch.from_numpy(mu))
    torch.manual_seed(1212)
    u_draw = torch.rand(R, S)
    smp_draw = torch.rand(S)
    torch.manual_seed(1212)
    Cr_h, Cf_h, _ = quiet(vr.vol_render, mlp, dvec, o, num_samples=S, t=torch.from_numpy(t), update_mask=False,
                          dir_norm=dn, hierarchical=True)
    loss_h = crit(Cr_h, gt) + crit(Cf_h, gt)
    loss_h.backward()
    g12 = dict(u=u_draw.numpy(), samples01=smp_draw.numpy(), Cr=Cr_h.detach().numpy(), Cf=Cf_h.detach().numpy(),
               loss=loss_h.item(), dtables=np.stack([enc.Embedding_list[l].weight.grad.numpy() for l in range(L)]))
    for name, p in mlp.module.named_parameters():
        g12["g." + name] = p.grad.numpy()
    # the sampler on its own, with weights that include negatives (clamped to 0 in place, helper.py:36)
    wts_in = torch.from_numpy(f32(rng.standard_normal((R, S, 1)) * 0.3 + 0.2))
    torch.manual_seed(1111)
    u11 = torch.rand(R, S)
    s11 = torch.rand(S)
    torch.manual_seed(1111)
    rays_f, t_f = quiet(ref.helper.hierarchical_sampling, o, dvec, z_vals=torch.from_numpy(t), weights=wts_in.clone(),
                        n_samples=S, tn=2.0, tf=6.0, device="cpu")
    g12.update(hs_weights=wts_in.numpy(), hs_u=u11.numpy(), hs_samples01=s11.numpy(), hs_rays=rays_f.numpy(), hs_t=t_f.numpy())
    np.savez_compressed(os.path.join(OUT, "g12_hierarchical.npz"), **g12)

    # ---------------- G13: masked branch with a MIXED occupancy grid (vol_renderer.py:133-140,209-221) ----------
    # The shipped trainer never clears grid cells, so the masked assign is exercised here with a grid that
    # is False in a third of its 256^3 cells: blocks of 8^3 cells with (bx + 2*by + 3*bz) % 3 == 0.  The pattern is a
    # formula so that the 16 MiB grid does not have to travel; the fixture records what the reference renders with it.
    enc = build_encoder(ref, tables, 2048.0, 16, mu, sigma)          # original (pre-step) weights
    mlp = torch.nn.DataParallel(build_mlp(ref, params))
    vr = ref.vol_renderer.Volume_Renderer(H=8, W=8, K=Kd, near=2.0, far=6.0, device="cpu", Pos_encode=enc,
                                          Dir_encode=pe, max_dim=2 ** 10, sigma_val=torch.tensor(float(sigma)),
                                          mu=torch.from_numpy(mu))
    gi = torch.arange(vr.grid_size) // 8
    vr.bool_grid[...] = ((gi[:, None, None] + 2 * gi[None, :, None] + 3 * gi[None, None, :]) % 3) != 0
    cap = {}
    ref.vol_renderer.calc_color = spy
    Cr_m, Cf_m, _ = quiet(vr.vol_render, mlp, dvec, o, num_samples=S, t=torch.from_numpy(t), update_mask=False,
                          dir_norm=dn, hierarchical=False)
    ref.vol_renderer.calc_color = orig_cc
    loss_m = crit(Cr_m, gt) + crit(Cf_m, gt)
    loss_m.backward()
    pts13 = (o[:, None, :] + dvec[:, None, :] * torch.from_numpy(t)[None, :, None]).reshape(-1, 3)
    g13 = dict(mask=vr.get_mask(pts13).numpy(), Cr=Cr_m.detach().numpy(), sig_out=cap["sigma"], rgb_out=cap["rgb"],
               loss=loss_m.item(), dtables=np.stack([enc.Embedding_list[l].weight.grad.numpy() for l in range(L)]),
               grid_size=vr.grid_size)
    for name, p in mlp.module.named_parameters():
        g13["g." + name] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g13_masked_render.npz"), **g13)

    # ---------------- G14: BASELINE config 1 - vanilla positional-encoding NeRF through vol_render ---------
    # (train.py:16-19's objects with a narrow trunk so that the weights fit a fixture: d_filter 32 instead of 256)
    p14 = ref_cpu.vanilla_nerf_init(1414, d_filter=32)
    net = ref.vol_renderer.NeRF(d_input=60, n_layers=8, d_filter=32, skip=(4,), d_viewdirs=60)
    net.load_state_dict(p14)
    pe10 = quiet(ref.encoder.PositionalEncoder, 3, 10)
    vr14 = ref.vol_renderer.Volume_Renderer(H=8, W=8, K=Kd, near=2.0, far=6.0, device="cpu", Pos_encode=pe10, Dir_encode=pe10,
                                            max_dim=2 ** 10, sigma_val=torch.tensor(float(sigma)), mu=torch.from_numpy(mu))
    cap = {}
    ref.vol_renderer.calc_color = spy
    with torch.no_grad():
        Cr14, _, _ = quiet(vr14.vol_render, net, dvec, o, num_samples=S, t=torch.from_numpy(t), update_mask=False, dir_norm=dn,
                           hierarchical=False)
    ref.vol_renderer.calc_color = orig_cc
    g14 = dict(Cr=Cr14.numpy(), sig_out=cap["sigma"], rgb_out=cap["rgb"], state_keys=np.array(list(net.state_dict().keys())))
    for k, v in p14.items():
        g14["p." + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "g14_vanilla_nerf.npz"), **g14)

    # ---------------- G16: Volume_Renderer.update_grid (vol_renderer.py:116-131) --------------------------------
    # Called directly on the reference's object, grid 16^3 (max_dim 64), bool_grid cleared first so that what the update
    # SETS is visible.  Three calls in a row on the same object (tmp_arr carries state): (1) a mix of alpha <= 0,
    # fractional, ordinary and > 127 values (ceil(alpha) wraps in the int8 scratch) with many points per cell (the
    # non-accumulating index_put: the LAST point of a cell decides); (2) the same points, new alphas - cells whose count
    # wrapped negative in (1) start from that value; (3) nothing positive at all -> the whole grid becomes True.
    rg = np.random.default_rng(1616)
    mu16 = np.array([-1.0, -0.5, 0.25], dtype=np.float32)
    vr16 = ref.vol_renderer.Volume_Renderer(H=8, W=8, K=Kd, near=2.0, far=6.0, device="cpu", Pos_encode=None, Dir_encode=None,
                                            max_dim=64, sigma_val=torch.tensor(3.0), mu=torch.from_numpy(mu16))
    pts16 = f32(mu16 + rg.uniform(0.02, 2.9, (3000, 3)))
    def alphas(k):
        a = rg.normal(0.3, 1.0, 3000)
        a[rg.uniform(0, 1, 3000) < 0.1] = 0.0
        big = rg.uniform(0, 1, 3000) < 0.06
        a[big] = rg.uniform(100, 400, int(big.sum()))
        return f32(a)
    g16 = dict(points=pts16, mu=mu16, sigma_val=np.float32(3.0), grid_size=vr16.grid_size)
    vr16.bool_grid[...] = False
    for k in range(3):
        a = alphas(k) if k < 2 else f32(-np.abs(rg.normal(0, 1, 3000)))
        g16[f"alpha{k}"] = a.copy()
        if k == 2:
            vr16.bool_grid[...] = False
            vr16.tmp_arr[...] = 0
        vr16.update_grid(torch.from_numpy(pts16.copy()), torch.from_numpy(a.copy()))
        g16[f"grid{k}"] = vr16.bool_grid.numpy().copy()
        g16[f"tmp{k}"] = vr16.tmp_arr.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g16_update_grid.npz"), **g16)

    # ---------------- G10: PSNR + bounding box -----------------------------------------
    a = torch.from_numpy(f32(rng.uniform(0, 1, (50, 3))))
    b = torch.from_numpy(f32(rng.uniform(0, 1, (50, 3))))
    np.savez_compressed(os.path.join(OUT, "g10_psnr.npz"), a=a.numpy(), b=b.numpy(),
                        psnr=ref.helper.calc_psnr(a, b).numpy())
    print("golden vectors written to", OUT)
    for fn in sorted(os.listdir(OUT)):
        print(f"  {fn}: {os.path.getsize(os.path.join(OUT, fn)) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
