"""GPU (-m gpu): the world > 1 branch of HashNeRFTrainer.step on a device - two ranks sharing cuda:0 over gloo (the
box has one GPU; RCCL over xGMI is the driver's 8-GPU run).  Rays are sharded, each rank runs the half-level K2
launches interleaved with the staged all-reduce, and the reduced gradient / updated parameters must equal the
single-process step over the whole batch (mean of equal-shard means == mean over all rays, up to summation order)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
R, S, T = 2048, 64, 2 ** 12   # 65 536 points per rank: the LDS scatter kernel on every rank


def _setup(dev):
    from hbr_amd import synthetic
    from hbr_amd.trainer import build_default_model
    o, d, dn, gt = (a.to(dev) for a in synthetic.scene_rays(R, seed=81))
    mn, mx, sig = synthetic.ray_bbox(o, d)
    enc, _, mlp = build_default_model(mn, sig, dev, T=T, seed=5)
    with torch.no_grad():
        enc.stacked_tables().uniform_(-0.3, 0.3, generator=torch.Generator(device=dev).manual_seed(6))
    return (o, d, dn.reshape(-1), gt), enc, mlp


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from hbr_amd import dist as hd
    from hbr_amd.trainer import HashNeRFTrainer
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    hd.init_from_env(backend="gloo")
    batch, enc, mlp = _setup(dev)
    tr = HashNeRFTrainer(enc, mlp, num_samples=S, total_steps=100, seed=9, overlap_comm=True)
    assert tr.world == world and tr.split_scatter
    shard = hd.shard_batch(batch, rank, world)
    losses = [float(tr.step(*shard)) for _ in range(2)]
    torch.cuda.synchronize()
    out = {"grad": tr.grad.cpu(), "tables": tr.tables.cpu(), "flat": tr.flat.cpu(), "losses": losses}
    # autotune_comm: both ranks must reach the same decision (it is taken from one MAX all-reduce) and stay replicas
    tune = tr.autotune_comm(lambda i: shard, steps=2)
    torch.cuda.synchronize()
    out.update(chosen=tune["chosen"], tuned_ms=[tune["single_ms_per_step"], tune["staged_ms_per_step"]], tables_after_tune=tr.tables.cpu())
    torch.save(out, os.path.join(out_dir, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_two_rank_trainer_step_equals_single_process(tmp_path):
    from hbr_amd.trainer import HashNeRFTrainer
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "r1.pt", weights_only=True)
    # replicas stay identical: same reduced gradient, same parameters after two optimiser steps
    assert torch.equal(r0["grad"], r1["grad"]) and torch.equal(r0["tables"], r1["tables"]) and torch.equal(r0["flat"], r1["flat"])
    assert r0["chosen"] == r1["chosen"] and r0["tuned_ms"] == r1["tuned_ms"] and torch.equal(r0["tables_after_tune"], r1["tables_after_tune"])
    dev = torch.device("cuda", 0)
    batch, enc, mlp = _setup(dev)
    tr = HashNeRFTrainer(enc, mlp, num_samples=S, total_steps=100, seed=9)
    assert tr.world == 1
    losses = [float(tr.step(*batch)) for _ in range(2)]
    # grad holds the SUM over ranks of per-shard mean-loss gradients; 1/world is folded into the Adam kernel
    g_sum, g_full = r0["grad"].to(dev) / 2, tr.grad
    assert torch.allclose(g_sum, g_full, rtol=1e-3, atol=1e-5 * float(g_full.abs().max()))
    assert np.isclose(np.mean([r0["losses"][0], r1["losses"][0]]), losses[0], rtol=1e-5)
    assert torch.allclose(r0["tables"].to(dev), tr.tables, rtol=0, atol=2e-3)   # Adam: +-lr*sign(g) where |g| ~ 0
    assert float((r0["tables"].to(dev) - tr.tables).abs().mean()) < 1e-5
