"""CPU: pin oracle/ref_cpu.py (the restatement) to the golden vectors produced by the
reference's own modules (oracle/make_golden.py).  Tolerances are stated per check."""
import numpy as np
import torch

import ref_cpu
from conftest import load_golden


def T_(a):
    return torch.from_numpy(np.asarray(a))


def test_g2_level_scales_bit_exact():
    g = load_golden("g2_level_scales.npz")
    for tag, (nmax, nmin, L) in {"f2048_16": (2048.0, 16, 16), "i2048_16": (2048, 16, 16),
                                 "f512_8": (512.0, 16, 8), "f4096_16": (4096.0, 16, 16)}.items():
        s = ref_cpu.level_scales(nmin, nmax, L).numpy()
        assert np.array_equal(s.view(np.uint32), g[tag].view(np.uint32)), tag
    # the top level is NOT exactly 2048 (SURVEY 7 hard part 2.i)
    assert g["f2048_16"][-1] != 2048.0


def test_g1_hash_kat_exact():
    g = load_golden("g1_hash_kat.npz")
    c = g["coords"]
    for T in (2 ** 16, 2 ** 19, 2 ** 10, 1000, 92681):
        h = ref_cpu.spatial_hash_np(c[:, 0], c[:, 1], c[:, 2], T)
        assert np.array_equal(h, g[f"hash_T{T}"]), T
        assert h.min() >= 0 and h.max() < T
    for log2T in (16, 19, 10):  # uint32 form == int64 floor-mod form for power-of-two T
        h = ref_cpu.spatial_hash_u32_np(c[:, 0], c[:, 1], c[:, 2], log2T)
        assert np.array_equal(h, g[f"hash_T{1 << log2T}"])
    # corner n takes +1 on axis d iff bit d of n is set
    ids = g["corner_ids"]  # [16,8,3]
    for n in range(8):
        off = np.array([(n >> 0) & 1, (n >> 1) & 1, (n >> 2) & 1])
        assert np.array_equal(ids[:, n, :], c[:16] + off)


def _tables_for(g):
    if "tables" in g:
        return g["tables"]
    rng = np.random.default_rng(int(g["seed"]))
    g["mu"]  # noqa
    return rng.uniform(-1.0, 1.0, (int(g["L"]), int(g["T"]), int(g["F"]))).astype(np.float32)


def _dense_grads(g):
    if "dtables" in g:
        return g["dtables"]
    out = np.zeros((int(g["L"]), int(g["T"]), int(g["F"])), np.float32)
    out[g["dtab_l"], g["dtab_row"]] = g["dtab_val"]
    return out


def _encoder_case(name):
    g = load_golden(name)
    tables = _tables_for(g)
    L = int(g["L"])
    scales = ref_cpu.level_scales(16, float(g["N_max"]), L)
    tabs = [T_(tables[l]).clone().requires_grad_(True) for l in range(L)]
    x = T_(g["x"])
    y = ref_cpu.hash_encode(x, tabs, scales, T_(g["mu"]), torch.tensor(float(g["sigma"])))
    # features: |d| <= 1e-6*max|y| (sum order over 8 corners differs)
    tol = 1e-6 * np.abs(g["y"]).max() + 1e-9
    assert np.abs(y.detach().numpy() - g["y"]).max() <= tol
    y.backward(T_(g["dy"]))
    got = np.stack([t.grad.numpy() for t in tabs])
    ref = _dense_grads(g)
    assert np.allclose(got, ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())
    # explicit scatter-add form
    got2 = ref_cpu.hash_encode_backward(x, T_(g["dy"]), scales, T_(g["mu"]), torch.tensor(float(g["sigma"])),
                                        int(g["T"]), int(g["F"])).numpy()
    assert np.allclose(got2, ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())
    # untouched rows have exactly zero gradient
    assert np.array_equal(got2 == 0, ref == 0)


def test_g3_encoder_T10():
    _encoder_case("g3_encoder_T10.npz")


def test_g3_encoder_T16():
    _encoder_case("g3_encoder_T16.npz")


def test_g3_encoder_non_pow2_T():
    _encoder_case("g3_encoder_T1000.npz")


def test_g4_dir_encoding():
    g = load_golden("g4_dir_pe.npz")
    d = T_(g["d"])
    assert np.allclose(ref_cpu.dir_encode(d, 4).numpy(), g["pe4"], rtol=0, atol=1e-7)
    assert np.allclose(ref_cpu.dir_encode(d, 10).numpy(), g["pe10"], rtol=0, atol=1e-7)
    # k=0 terms are the constants sin 0 = 0, cos 0 = 1
    assert np.all(g["pe4"][:, 0] == 0) and np.all(g["pe4"][:, 4] == 1)


def test_g5_mlp_forward_backward():
    g = load_golden("g5_mlp.npz")
    params = {k[2:]: T_(v).clone().requires_grad_(True) for k, v in g.items() if k.startswith("p.")}
    feat = T_(g["feat"]).clone().requires_grad_(True)
    out = ref_cpu.mlp_forward(feat, ref_cpu.dir_encode(T_(g["dirs"]), 4), params)
    assert np.allclose(out.detach().numpy(), g["out"], rtol=1e-5, atol=1e-6)
    out.backward(T_(g["dout"]))
    assert np.allclose(feat.grad.numpy(), g["dfeat"], rtol=1e-4, atol=1e-6)
    for k, p in params.items():
        assert np.allclose(p.grad.numpy(), g["g." + k], rtol=1e-4, atol=1e-5), k


def test_g6_composite_forward_backward():
    g = load_golden("g6_composite.npz")
    sig = T_(g["sigma"]).clone().requires_grad_(True)
    rgb = T_(g["rgb"]).clone().requires_grad_(True)
    Cr, w = ref_cpu.composite(T_(g["t"]), rgb, sig, T_(g["dir_norm"]))
    scale = np.abs(g["wts"]).max(axis=(1, 2), keepdims=True)  # negative sigma => huge weights on some rays
    assert np.all(np.abs(w.detach().numpy() - g["wts"]) <= 1e-5 * scale + 1e-7)
    cscale = np.abs(g["wts"]).max(axis=(1, 2))[:, None] * 2
    assert np.all(np.abs(Cr.detach().numpy() - g["Cr"]) <= 1e-5 * cscale + 1e-6)
    Cr.backward(T_(g["dC"]))
    ds, dr = sig.grad.numpy(), rgb.grad.numpy()
    assert np.allclose(dr, g["drgb"], rtol=1e-4, atol=1e-6 * np.abs(g["drgb"]).max())
    rs = np.abs(g["dsigma"]).max(axis=1, keepdims=True)
    assert np.all(np.abs(ds - g["dsigma"]) <= 2e-4 * rs + 1e-6)
    # sigma < -10 is clamped with ZERO gradient (helper.py:76); last sample has delta = 0
    assert g["dsigma"][1, 3] == 0 and ds[1, 3] == 0 and ds[1, 5] == 0
    assert np.all(g["wts"][:, -1, 0] == 0) and np.all(w.detach().numpy()[:, -1, 0] == 0)
    g1 = load_golden("g6b_composite_scalar_norm.npz")
    Cr1, _ = ref_cpu.composite(T_(g["t"]), T_(g["rgb"]), T_(g["sigma"]), 1)
    assert np.all(np.abs(Cr1.numpy() - g1["Cr"]) <= 1e-5 * np.abs(g1["wts"]).max(axis=(1, 2))[:, None] * 2 + 1e-6)


def test_g7_rays_and_sampler():
    g = load_golden("g7_rays.npz")
    o, d, n = ref_cpu.get_od(int(g["H"]), int(g["W"]), T_(g["K"]), T_(g["c2w"]))
    assert np.allclose(o.numpy(), g["o"], atol=1e-6)
    assert np.allclose(d.numpy(), g["d"], atol=1e-6)
    assert np.allclose(n.numpy(), g["n"], rtol=1e-6)
    t = ref_cpu.strat_jitter_to_t(2.0, 6.0, 16, T_(g["strat_u"]))
    assert np.allclose(t.numpy(), g["strat_t"], rtol=0, atol=1e-6)
    assert g["strat_t"].max() > 6.0  # can exceed far


def _g8_state(g):
    L = int(g["L"])
    tabs = [T_(g["tables"][l]).clone().requires_grad_(True) for l in range(L)]
    params = {k[2:]: T_(v).clone().requires_grad_(True) for k, v in g.items() if k.startswith("p.")}
    scales = ref_cpu.level_scales(16, 2048.0, L)
    return tabs, params, scales


def test_g8_full_render_and_train_step():
    g = load_golden("g8_render_step.npz")
    tabs, params, scales = _g8_state(g)
    mu, sigma = T_(g["mu"]), torch.tensor(float(g["sigma"]))
    o, d, dn, gt, t = (T_(g[k]) for k in ("o", "d", "dir_norm", "gt", "t"))
    Cr, sig, rgb = ref_cpu.render(o, d, t, dn, tabs, scales, mu, sigma, params)
    assert np.allclose(sig.detach().numpy(), g["sig_out"], rtol=1e-4, atol=1e-5)
    assert np.allclose(rgb.detach().numpy(), g["rgb_out"], rtol=1e-4, atol=1e-5)
    assert np.allclose(Cr.detach().numpy(), g["Cr"], rtol=1e-4, atol=1e-5)
    opts = ref_cpu.make_optimizers(tabs, params.values(), int(g["total_steps"]))
    # grads
    loss = ref_cpu.train_loss(Cr, gt)
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    loss.backward()
    got = np.stack([t_.grad.numpy() for t_ in tabs])
    assert np.allclose(got, g["dtables"], rtol=1e-3, atol=1e-5 * np.abs(g["dtables"]).max())
    for k, p in params.items():
        assert np.allclose(p.grad.numpy(), g["g." + k], rtol=1e-3, atol=1e-5 * np.abs(g["g." + k]).max()), k
    oe, om, se, sm = opts
    oe.step(); om.step(); se.step(); sm.step()
    after = np.stack([t_.detach().numpy() for t_ in tabs])
    # Adam's first step moves every touched row by ~lr*sign(g): compare loosely where |g| is tiny
    assert np.allclose(after, g["tables_after"], rtol=0, atol=2e-3)
    assert np.mean(np.abs(after - g["tables_after"]) < 1e-5) > 0.99
    for k, p in params.items():
        assert np.allclose(p.detach().numpy(), g["a." + k], rtol=0, atol=5e-4), k
    assert abs(se.get_last_lr()[0] - float(g["lr_embed_after"])) < 1e-9
    assert abs(sm.get_last_lr()[0] - float(g["lr_mlp_after"])) < 1e-9
    # state-dict key names the build must reproduce (SURVEY 5 checkpoint row)
    assert list(g["state_keys_enc"]) == [f"Embedding_list.{i}.weight" for i in range(16)]
    assert list(g["state_keys_mlp"])[0] == "module.sig_model.0.weight"


def test_g10_psnr():
    g = load_golden("g10_psnr.npz")
    assert np.allclose(ref_cpu.psnr(T_(g["a"]), T_(g["b"])).numpy(), g["psnr"], rtol=1e-6)


def test_g12_hierarchical_sampler_and_render():
    g = load_golden("g12_hierarchical.npz")
    g8 = load_golden("g8_render_step.npz")
    o, d, dn, gt, t = (T_(g8[k]) for k in ("o", "d", "dir_norm", "gt", "t"))
    S = t.shape[0]
    pts, tf_ = ref_cpu.hierarchical_sample(o, d, t, T_(g["hs_weights"]), S, 2.0, 6.0, T_(g["hs_u"]), T_(g["hs_samples01"]))
    assert np.array_equal(tf_.numpy(), g["hs_t"])           # same bins, same sort: exact
    assert np.allclose(pts.numpy(), g["hs_rays"], atol=1e-6)
    assert tf_.shape[1] == 2 * S and bool((tf_[:, 1:] >= tf_[:, :-1]).all())
    tabs, params, scales = _g8_state(g8)
    mu, sigma = T_(g8["mu"]), torch.tensor(float(g8["sigma"]))
    Cr, Cf = ref_cpu.render_hierarchical(o, d, t, dn, tabs, scales, mu, sigma, params, T_(g["u"]), T_(g["samples01"]))
    assert np.allclose(Cr.detach().numpy(), g["Cr"], rtol=1e-4, atol=1e-5)
    assert np.allclose(Cf.detach().numpy(), g["Cf"], rtol=1e-4, atol=1e-5)
    loss = torch.mean((Cr - gt) ** 2) + torch.mean((Cf - gt) ** 2)
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * float(g["loss"])
    loss.backward()
    got = np.stack([x.grad.numpy() for x in tabs])
    assert np.allclose(got, g["dtables"], rtol=1e-3, atol=1e-5 * np.abs(g["dtables"]).max())
    for k, p in params.items():
        assert np.allclose(p.grad.numpy(), g["g." + k], rtol=1e-3, atol=1e-5 * np.abs(g["g." + k]).max()), k


def test_g13_masked_render_with_mixed_occupancy_grid():
    """vol_render's masked branch with a grid that is False in a third of its cells (reference vol_renderer.py:133-140,
    209-221): mask, sigma/rgb handed to calc_color, Cr, loss and every gradient."""
    g = load_golden("g13_masked_render.npz")
    g8 = load_golden("g8_render_step.npz")
    tabs, params, scales = _g8_state(g8)
    mu, sigma = T_(g8["mu"]), torch.tensor(float(g8["sigma"]))
    o, d, dn, gt, t = (T_(g8[k]) for k in ("o", "d", "dir_norm", "gt", "t"))
    grid = ref_cpu.block_pattern_grid(int(g["grid_size"]))
    Cr, sig, rgb, mask = ref_cpu.render_masked(o, d, t, dn, tabs, scales, mu, sigma, params, grid, mu, sigma)
    assert np.array_equal(mask.numpy(), g["mask"]) and 0.3 < 1 - mask.float().mean().item() < 0.4
    assert np.allclose(sig.detach().numpy(), g["sig_out"], rtol=1e-4, atol=1e-5)
    assert np.allclose(rgb.detach().numpy(), g["rgb_out"], rtol=1e-4, atol=1e-5)
    assert np.array_equal(sig.detach().numpy().reshape(-1)[~g["mask"]], np.zeros((~g["mask"]).sum(), np.float32))
    assert np.allclose(Cr.detach().numpy(), g["Cr"], rtol=1e-4, atol=1e-5)
    loss = ref_cpu.train_loss(Cr, gt)
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    loss.backward()
    got = np.stack([t_.grad.numpy() for t_ in tabs])
    assert np.allclose(got, g["dtables"], rtol=1e-3, atol=1e-5 * np.abs(g["dtables"]).max())
    for k, p in params.items():
        assert np.allclose(p.grad.numpy(), g["g." + k], rtol=1e-3, atol=1e-5 * np.abs(g["g." + k]).max()), k


def test_g14_config1_vanilla_positional_encoding_nerf():
    """BASELINE config 1 (train.py's positional-encoding NeRF; CPU plumbing, outside the accelerated path): the oracle's
    restatement of NeRF.forward + vol_render against the reference's own modules (vol_renderer.py:12-86,141-223)."""
    g = load_golden("g14_vanilla_nerf.npz")
    g8 = load_golden("g8_render_step.npz")
    p = {k[2:]: T_(v) for k, v in g.items() if k.startswith("p.")}
    assert [k for k in g["state_keys"]] == list(p.keys())  # same names and order as the reference's state dict
    o, d, dn, t = (T_(g8[k]) for k in ("o", "d", "dir_norm", "t"))
    Cr, sig, rgb = ref_cpu.render_vanilla(o, d, t, dn, p)
    assert np.allclose(sig.numpy(), g["sig_out"], rtol=1e-4, atol=1e-5)
    assert np.allclose(rgb.numpy(), g["rgb_out"], rtol=1e-4, atol=1e-5)
    assert np.allclose(Cr.numpy(), g["Cr"], rtol=1e-4, atol=1e-5)


def test_config1_plumbing_at_baseline_size():
    """4096 rays x 64 samples through PositionalEncoder(3,10) x 2 + NeRF(d_input=60, d_viewdirs=60),
    near 2 / far 6, seed 0 - shapes and finite values (this configuration is plumbing on the CPU; no kernel serves it)."""
    torch.manual_seed(0)
    R, S = 4096, 64
    o, d, dn, _ = ref_cpu.synthetic_rays(R, seed=0)
    t = ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.rand(S))
    with torch.no_grad():
        Cr, sig, rgb = ref_cpu.render_vanilla(o, d, t, dn, ref_cpu.vanilla_nerf_init(0))
    assert Cr.shape == (R, 3) and sig.shape == (R, S) and rgb.shape == (R, S, 3)
    assert bool(torch.isfinite(Cr).all()) and float(sig.min()) >= 0.0 and float(sig.max()) <= 1.0 and float(rgb.min()) >= 0.0


def test_update_grid_vs_reference(golden):
    """G16: Volume_Renderer.update_grid called directly on the reference's object, three calls in a row (state in tmp_arr)."""
    g = golden("g16_update_grid.npz")
    G = int(g["grid_size"])
    grid, tmp = torch.zeros((G, G, G), dtype=torch.bool), torch.zeros((G, G, G), dtype=torch.int8)
    pts, mu, sv = torch.from_numpy(g["points"]), torch.from_numpy(g["mu"]), torch.tensor(float(g["sigma_val"]))
    for k in range(3):
        if k == 2:
            grid[...] = False
            tmp[...] = 0
        ref_cpu.update_grid(pts, torch.from_numpy(g[f"alpha{k}"]), grid, tmp, mu, sv)
        assert np.array_equal(grid.numpy(), g[f"grid{k}"]), k
        assert np.array_equal(tmp.numpy(), g[f"tmp{k}"]), k
    assert (g["tmp0"] < 0).sum() > 10 and g["grid2"].all() and 0 < g["grid0"].sum() < g["grid1"].sum() < G ** 3


def test_oracle_on_the_references_trained_model_and_first_steps(golden):
    """G15 / G15b (oracle/make_psnr_golden.py: the reference's own modules trained for 2000 steps): the oracle renders the
    reference's trained weights to the reference's own held-out colours and PSNR, and its train step reproduces the
    first losses of the reference's loop (train_hash2.py:211-234) from the same seeded inputs."""
    import make_psnr_golden as MP
    g, gw = golden("g15_converged_psnr.npz"), golden("g15b_trained_weights.npz")
    mn, sig, batches, test = MP.scene()
    sc = ref_cpu.level_scales(16, 2048.0, MP.L)
    tabs = [torch.from_numpy(gw["tables"][l]) for l in range(MP.L)]
    prm = {k[2:]: torch.from_numpy(v) for k, v in gw.items() if k.startswith("p.")}
    t_eval = torch.linspace(MP.NEAR, MP.FAR, MP.S)
    with torch.no_grad():
        C, _, _ = ref_cpu.render(test[0][:512], test[1][:512], t_eval, test[2][:512], tabs, sc, mn, sig, prm)
    assert np.allclose(C.numpy(), gw["Cr_eval"][:512], rtol=1e-4, atol=1e-5)
    assert abs(float(ref_cpu.psnr(torch.from_numpy(gw["Cr_eval"]), test[3])) - float(gw["psnr"])) < 1e-3
    # first steps of seed g["seeds"][0]
    seed, steps = int(g["seeds"][0]), int(g["steps"])
    tables0, u, params0 = MP.seeded_inputs(seed, steps)
    assert abs(MP.checksum(tables0, u, *[v.numpy() for v in params0.values()]) - float(g["input_checksum"][0])) < 1e-6 * float(g["input_checksum"][0])
    tt = [torch.from_numpy(tables0[l]).clone().requires_grad_(True) for l in range(MP.L)]
    pp = {k: v.clone().requires_grad_(True) for k, v in params0.items()}
    opts = ref_cpu.make_optimizers(tt, pp.values(), steps)
    for k in range(3):
        t = ref_cpu.strat_jitter_to_t(MP.NEAR, MP.FAR, MP.S, torch.from_numpy(u[k]))
        loss = ref_cpu.train_step(batches[k % MP.NB], t, tt, sc, mn, sig, pp, opts)
        assert abs(float(loss) - float(g["loss_head"][0][k])) <= 2e-4 * float(g["loss_head"][0][k]), (k, float(loss), float(g["loss_head"][0][k]))
    ulp = g["psnr_ulp"][:, -1] - g["psnr"][:, -1]
    assert g["psnr"].shape == g["psnr_ulp"].shape == (len(g["seeds"]), 40) and len(g["seeds"]) >= 9 and 0.1 < np.abs(ulp).max() < 2.0  # the metric's own noise floor
