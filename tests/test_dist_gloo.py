"""CPU, world_size 2 over gloo: the ray-sharding + single flat all-reduce used for N>1 GPUs reproduces the
single-process gradient.  The per-rank compute here is the CPU oracle (tests may use it); the sharding, flat-buffer
and collective code is the product's (hbr_amd/dist.py)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

import ref_cpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    R, S, L, T = 32, 12, 4, 2 ** 8
    o, d, dn, gt = ref_cpu.synthetic_rays(R, seed=3)
    mn, mx, sig = ref_cpu.bbox_mu_sigma(o, d)
    rng = np.random.default_rng(4)
    tables = torch.from_numpy(rng.uniform(-0.5, 0.5, (L, T, 2)).astype(np.float32))
    params = ref_cpu.mlp_init(5)
    t = ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.from_numpy(rng.uniform(0, 1, S).astype(np.float32)))
    scales = ref_cpu.level_scales(16, 256.0, L)
    return (o, d, dn, gt), t, tables, params, scales, mn, sig


def _flat_grad(batch, t, tables, params, scales, mn, sig):
    tabs = [tables[l].clone().requires_grad_(True) for l in range(tables.shape[0])]
    # the MLP oracle is fixed at 32 inputs: pad the 4-level features with zeros via 12 dummy zero tables
    zeros = [torch.zeros_like(tabs[0]) for _ in range(16 - len(tabs))]
    sc = torch.cat([scales, scales[-1:].repeat(16 - len(tabs))])
    prm = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    o, d, dn, gt = batch
    Cr, _, _ = ref_cpu.render(o, d, t, dn, tabs + zeros, sc, mn, sig, prm)
    ref_cpu.train_loss(Cr, gt).backward()
    return torch.cat([torch.stack([x.grad for x in tabs]).reshape(-1)] + [p.grad.reshape(-1) for p in prm.values()])


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from hbr_amd import dist as hd
    r, w = hd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    batch, t, tables, params, scales, mn, sig = _problem()
    # replicas start identical: rank 1 deliberately starts from garbage and is overwritten by the broadcast
    if rank == 1:
        tables = tables + 1.0
    hd.broadcast_params_([tables], src=0)
    shard = hd.shard_batch(batch, rank, world)
    assert shard[0].shape[0] == batch[0].shape[0] // world
    flat = _flat_grad(shard, t, tables, params, scales, mn, sig)
    # the staged variant the trainer uses for N > 1 (MLP block, upper levels, lower levels): the pieces partition the
    # buffer, so it must give the very same bits as the single collective
    staged = flat.clone()
    red = hd.StagedAllReduce(world)
    nt = tables.numel()
    cut = nt // 2
    for piece in (staged[nt:], staged[cut:nt], staged[:cut]):
        red.launch(piece)
    red.finish()
    staged.mul_(1.0 / world)
    # ... and in the two pieces the trainer issues since round 3: [upper levels | MLP block], then the lower levels
    staged2 = flat.clone()
    red = hd.StagedAllReduce(world)
    assert red.active
    for piece in (staged2[cut:], staged2[:cut]):
        red.launch(piece)
    red.finish()
    staged2.mul_(1.0 / world)
    # ... and the list form the --hierarchical route uses (28 tensors packed into one collective)
    pieces = [flat[:cut].clone(), flat[cut:nt].clone().view(-1, 2), flat[nt:].clone()]
    hd.allreduce_mean_grads_(pieces, world)
    hd.allreduce_mean_(flat, world)
    assert torch.equal(staged, flat) and torch.equal(staged2, flat)
    assert torch.equal(torch.cat([p.reshape(-1) for p in pieces]), flat)
    torch.save(flat, os.path.join(out_dir, f"g{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_sharded_gradient_equals_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g0 = torch.load(tmp_path / "g0.pt", weights_only=True)
    g1 = torch.load(tmp_path / "g1.pt", weights_only=True)
    assert torch.equal(g0, g1)  # every rank ends with the same reduced gradient
    batch, t, tables, params, scales, mn, sig = _problem()
    full = _flat_grad(batch, t, tables, params, scales, mn, sig)
    # mean over equal shards of per-shard mean losses == mean over all rays (up to summation order)
    assert torch.allclose(g0, full, rtol=1e-4, atol=1e-6 * float(full.abs().max()))


def test_staged_allreduce_is_inert_without_a_group():
    """world == 1: no collective is issued - not even with force=True - unless a process group exists (the one-rank RCCL
    rehearsal of tests/test_gpu_rccl_world1.py initialises one)."""
    from hbr_amd import dist as hd
    assert not torch.distributed.is_initialized()
    x = torch.arange(8.0)
    for force in (False, True):
        red = hd.StagedAllReduce(1, force=force)
        assert not red.active
        red.launch(x)
        red.finish()
    assert torch.equal(x, torch.arange(8.0))


def test_shard_bounds_cover_and_are_equal():
    from hbr_amd.dist import shard_bounds
    for n in (16000, 16001, 7):
        for world in (1, 2, 4, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert all(hi - lo == n // world for lo, hi in spans)
            assert spans[0][0] == 0 and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def test_cosine_lr_matches_torch_scheduler():
    from hbr_amd.trainer import cosine_lr
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=0.05)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=37, eta_min=1e-4)
    for k in range(37):
        assert abs(opt.param_groups[0]["lr"] - cosine_lr(0.05, 1e-4, k, 37)) < 1e-9
        opt.step(); sch.step()
