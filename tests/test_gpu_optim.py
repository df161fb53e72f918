"""GPU (-m gpu): hbr_amd.optim.Adam / AdamW (the fused kernel behind torch.optim's interface) against torch.optim
itself - the calls the reference makes at train_hash2.py:141-142,227-234 - on the drop-in modules: same parameters after
several steps of the reference's loop body, same state_dict layout (each loads the other's), schedulers drive it, and the
one-launch fast path is the one taken for HashEncoder / MLP_3D parameters."""
import numpy as np
import pytest
import torch

import ref_cpu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _world(seed=0):
    from hbr_amd.trainer import build_default_model
    from hbr_amd.vol_renderer import Volume_Renderer
    o, d, dn, gt = (a.to(DEV) for a in ref_cpu.synthetic_scene_rays(512, seed=21))
    mn, mx, sig = ref_cpu.bbox_mu_sigma(o.cpu(), d.cpu())
    enc, denc, mlp = build_default_model(mn, sig, DEV, T=2 ** 12, seed=seed)
    with torch.no_grad():
        for lvl in enc.Embedding_list:
            lvl.weight.mul_(2000.0)
    nerf = torch.nn.DataParallel(mlp, device_ids=[0])
    vr = Volume_Renderer(H=8, W=8, K=torch.eye(3), near=2.0, far=6.0, device=DEV, Pos_encode=enc, Dir_encode=denc, sigma_val=sig.to(DEV), mu=mn.to(DEV))
    return (o, d, dn, gt), enc, nerf, vr


def _run(opt_mod, steps=6):
    (o, d, dn, gt), enc, nerf, vr = _world()
    oe = opt_mod.Adam(enc.Embedding_list.parameters(), lr=0.05)
    om = opt_mod.AdamW(nerf.parameters(), lr=0.005)
    se = torch.optim.lr_scheduler.CosineAnnealingLR(oe, T_max=20, eta_min=1e-4)
    sm = torch.optim.lr_scheduler.CosineAnnealingLR(om, T_max=20, eta_min=1e-4)
    crit = torch.nn.MSELoss()
    t = torch.linspace(2.0, 6.0, 32, device=DEV)
    losses = []
    for k in range(steps):
        Cr, Cf, _ = vr.vol_render(nerf, d, o, num_samples=32, t=t, update_mask=False, dir_norm=dn, hierarchical=False)
        loss = crit(Cr, gt) + crit(Cf, gt)
        loss.backward()
        oe.step(); om.step(); se.step(); sm.step()
        om.zero_grad(set_to_none=True); oe.zero_grad(set_to_none=True)
        losses.append(float(loss.detach()))
    return enc, nerf, oe, om, losses


def test_fused_optimisers_follow_torch_optim():
    import hbr_amd.optim as fused
    ea, na, oea, oma, la = _run(torch.optim)
    eb, nb, oeb, omb, lb = _run(fused)
    assert np.allclose(la, lb, rtol=1e-5)
    ta, tb = ea.stacked_tables(), eb.stacked_tables()
    # Adam's update is lr * m / (sqrt(v) + eps): where |g| is of eps' order the quotient amplifies the last bit of g (the
    # same rule as the G9 test of the kernel against the reference's recorded step)
    dt = (ta - tb).abs()
    assert float((dt > 1e-5).float().mean()) < 1e-3 and float(dt.max()) < 2e-3
    for pa, pb in zip(na.parameters(), nb.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-4, atol=2e-6)
    # same state layout: keys, shapes, step counts; each loads the other's
    sa, sb = oma.state_dict(), omb.state_dict()
    assert sa["param_groups"][0].keys() >= {"lr", "betas", "eps", "weight_decay", "params"} and sb["param_groups"][0]["weight_decay"] == 0.01
    assert sa["state"].keys() == sb["state"].keys()
    for k in sa["state"]:
        assert set(sb["state"][k]) == {"step", "exp_avg", "exp_avg_sq"} and float(sa["state"][k]["step"]) == float(sb["state"][k]["step"]) == 6
        ma, mb = sa["state"][k]["exp_avg"], sb["state"][k]["exp_avg"]
        assert ma.shape == mb.shape and float((ma - mb).abs().max()) <= 1e-2 * float(ma.abs().max())  # six steps of slightly different parameters
    omb.load_state_dict(sa)
    oma.load_state_dict(omb.state_dict())
    assert abs(oeb.param_groups[0]["lr"] - oea.param_groups[0]["lr"]) < 1e-12 < oeb.param_groups[0]["lr"] < 0.05  # the scheduler moved it


def test_one_launch_per_group_and_fallbacks():
    import hbr_amd.optim as fused
    from hbr_amd import ops
    (o, d, dn, gt), enc, nerf, vr = _world(seed=3)
    calls = []
    real = ops.adam_step_multi
    ops.adam_step_multi = lambda segs: (calls.append(len(segs)), real(segs))[1]
    try:
        oe = fused.Adam(enc.Embedding_list.parameters(), lr=0.05)
        t = torch.linspace(2.0, 6.0, 32, device=DEV)
        Cr, _, _ = vr.vol_render(nerf, d, o, num_samples=32, t=t, dir_norm=dn, hierarchical=False)
        ((Cr - gt) ** 2).mean().backward()
        oe.step()
        assert calls == [1]          # 16 tables, 16 gradients, 16 + 16 moments: one segment, one launch
        # gradients in separate tensors (e.g. accumulated by hand): one segment per tensor, four per launch - same numbers
        ref = [p.detach().clone() for p in enc.Embedding_list.parameters()]
        (o2, d2, dn2, gt2), enc2, nerf2, vr2 = _world(seed=3)
        oe2 = fused.Adam(enc2.Embedding_list.parameters(), lr=0.05)
        Cr2, _, _ = vr2.vol_render(nerf2, d2, o2, num_samples=32, t=t, dir_norm=dn2, hierarchical=False)
        ((Cr2 - gt2) ** 2).mean().backward()
        for p in enc2.Embedding_list.parameters():
            p.grad = p.grad.clone()
        calls.clear()
        oe2.step()
        assert calls == [4, 4, 4, 4]
        for a, b in zip(ref, enc2.Embedding_list.parameters()):  # (16 384 points: K2's float-atomic kernel, sums differ in the last bit)
            assert float(((a - b).abs() > 1e-5).float().mean()) < 1e-3
    finally:
        ops.adam_step_multi = real
    with pytest.raises(NotImplementedError):
        fused.Adam(enc.Embedding_list.parameters(), lr=0.05, weight_decay=0.1)


def test_adjacent_but_separately_allocated_gradients_take_the_per_tensor_path():
    """gradients that sit back to back in memory WITHOUT sharing a storage (the caching allocator's arena
    hands out 512-byte-multiple blocks side by side) must not be taken for one flat buffer: the flat view would run
    past the first gradient's storage.  Built deterministically: views of one arena, each re-wrapped as its own
    storage-sized tensor via from_blob-like slicing is not possible in torch, so the separate storages are real
    allocations checked for adjacency; if the allocator did not place them adjacently the test still checks the result."""
    import hbr_amd.optim as fused
    from hbr_amd import ops
    from hbr_amd.optim import _consecutive
    n = 4096  # 16 KiB per tensor: a multiple of 512 B, so the small-block pool lays fresh allocations out back to back
    torch.cuda.empty_cache()
    buf = torch.zeros(4 * n, device=DEV)
    ps = [torch.nn.Parameter(buf[i * n:(i + 1) * n]) for i in range(4)]  # consecutive views of ONE buffer
    torch.manual_seed(0)
    grads = [torch.randn(n, device=DEV) for _ in range(4)]               # four storages of their own
    adjacent = all(grads[i + 1].data_ptr() == grads[i].data_ptr() + 4 * n for i in range(3))
    for p, g in zip(ps, grads):
        p.grad = g
    assert _consecutive([p.grad for p in ps]) is None                     # separate storages, whatever their addresses
    calls = []
    real = ops.adam_step_multi
    ops.adam_step_multi = lambda segs: (calls.append(len(segs)), real(segs))[1]
    try:
        opt = fused.Adam(ps, lr=0.01)
        opt.step()                                                        # used to raise setStorage ... out of bounds when adjacent
    finally:
        ops.adam_step_multi = real
    assert calls == [4], (calls, adjacent)
    want = [torch.nn.Parameter(torch.zeros(n, device=DEV)) for _ in range(4)]
    for p, g in zip(want, grads):
        p.grad = g.clone()
    torch.optim.Adam(want, lr=0.01).step()
    for a, b in zip(ps, want):
        assert torch.allclose(a, b, rtol=0, atol=1e-7)
    print("gradients were adjacent in the arena:", adjacent)


def test_fused_optimiser_step_bumps_versions_and_invalidates_the_weight_image():
    """the fused optimisers write p / m / v through raw pointers; the version counters must move as after a
    torch in-place op, so that ops' record of the packed weight image (keyed on _version) goes stale."""
    import hbr_amd.optim as fused
    from hbr_amd import ops
    (o, d, dn, gt), enc, nerf, vr = _world(seed=5)
    mlp = nerf.module
    flat, _ = mlp.flat_params()
    om = fused.AdamW(nerf.parameters(), lr=0.005)
    t = torch.linspace(2.0, 6.0, 32, device=DEV)
    Cr, _, _ = vr.vol_render(nerf, d, o, num_samples=32, t=t, dir_norm=dn, hierarchical=False)
    ((Cr - gt) ** 2).mean().backward()
    assert any(v[1][0] == flat.data_ptr() for v in ops._mlp_image.values())   # the render packed an image of `flat`
    v0 = [p._version for p in nerf.parameters()]
    om.step()
    assert all(p._version > a for p, a in zip(nerf.parameters(), v0))
    # (the parameters alias `flat` through `.data`, which does not share version counters: the record is dropped by address)
    assert not any(v[1][0] == flat.data_ptr() for v in ops._mlp_image.values())
    Cr2, _, _ = vr.vol_render(nerf, d, o, num_samples=32, t=t, dir_norm=dn, hierarchical=False)
    assert any(v[1][0] == flat.data_ptr() for v in ops._mlp_image.values())
    ops.free_workspaces()
    assert not ops._mlp_image
