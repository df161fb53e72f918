#!/usr/bin/env python3
"""Driver benchmark: ray-samples/s of a FULL hash-NeRF train step (fwd + bwd + optimiser) at 128 samples/ray
on the synthetic lego-shaped scene (BASELINE.json metric; configs[1]: L=16, F=2, T=2^16, 16000 rays x 128 samples,
bf16 MLP, fp32 tables).

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]
    N>1: one rank per GPU over RCCL.  Either the driver launches the ranks (torch.distributed.run sets WORLD_SIZE) or,
    run bare, this script starts them itself (a child `python -m torch.distributed.run ... bench.py ...`) before
    anything touches the GPU, and relays rank 0's JSON line and exit code.
    weak (default): 16000 rays PER RANK;  strong: the 16000-ray batch is split over the ranks.
    One all-reduce of the flat gradient buffer per step either way.

Prints ONE JSON line on rank 0.  Inputs are resident in HBM before the timed region.  `roofline` is measured live
with HIP events on the launch stream; `cpu_baseline` times the CPU oracle (oracle/ref_cpu.py, a port of the
reference's PyTorch path) on a bounded sample on rank 0 at N=1 only - the only place `oracle/` is imported.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16
# algorithmic work per ray-sample (DESIGN.md "Measurement")
def hash_fwd_bytes(feat_elem_bytes):   # x + 128 gathers of 8 B + y[32]      = 1164 B (fp32 y) / 1100 B (bf16 y)
    return 12 + 16 * 8 * 2 * 4 + 16 * 2 * feat_elem_bytes


def hash_bwd_bytes(feat_elem_bytes):   # x + dy[32] + 128 scatter-adds of 8 B = 1164 B (fp32 dy) / 1100 B (bf16 dy)
    return 12 + 16 * 2 * feat_elem_bytes + 16 * 8 * 2 * 4
MLP_FWD_FLOP = 2 * (32 * 64 + 64 * 64 + 64 * 16 + 39 * 64 + 64 * 64 + 64 * 3)  # 27904
MLP_BWD_FLOP = 3 * MLP_FWD_FLOP                     # recompute + data grad + weight grad


def cpu_baseline(S, mn, sig, total_steps, Rc=4096, budget_s=25.0):
    """the CPU restatement of the reference's path (oracle/ref_cpu.py - kind "port"; the reference's own
    files do not travel to the GPU box) timed on this box's host cores: fp32 fwd + bwd + optimiser steps of Rc = 4096
    rays x S samples of the same scene/config, median of up to 5 steps after one warm-up, on all cores of the box's CPU
    share and again on 8 threads (the survey's calibration point).  Bounded: each leg stops after `budget_s` seconds
    (>= 2 steps).  This is the ONLY use of oracle/ in this file."""
    import statistics
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_cpu
    try:
        share = len(os.sched_getaffinity(0))  # the box's CPU share, not os.cpu_count(): oversubscribing a cgroup stalls torch
    except AttributeError:
        share = os.cpu_count() or 1
    oc, dc, dnc, gtc = ref_cpu.synthetic_rays(Rc, seed=77)
    rng = np.random.default_rng(0)
    tabs = [torch.from_numpy(rng.uniform(-1e-4, 1e-4, (2 ** 16, 2)).astype(np.float32)).requires_grad_(True) for _ in range(16)]
    prm = {k: v.requires_grad_(True) for k, v in ref_cpu.mlp_init(0).items()}
    scales = ref_cpu.level_scales(16, 2048.0, 16)
    opts = ref_cpu.make_optimizers(tabs, prm.values(), total_steps)
    tc = ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.rand(S))
    mn_c, sig_c = mn.cpu(), sig.cpu()

    def leg(threads):
        torch.set_num_threads(threads)
        print(f"[bench] cpu baseline: {Rc} rays x {S} samples on {threads} threads ...", file=sys.stderr, flush=True)
        ref_cpu.train_step((oc, dc, dnc, gtc), tc, tabs, scales, mn_c, sig_c, prm, opts)  # warm-up
        times, c0 = [], time.perf_counter()
        while len(times) < 5 and (len(times) < 2 or time.perf_counter() - c0 < budget_s):
            s0 = time.perf_counter()
            ref_cpu.train_step((oc, dc, dnc, gtc), tc, tabs, scales, mn_c, sig_c, prm, opts)
            times.append(time.perf_counter() - s0)
        return Rc * S / statistics.median(times), len(times)

    # the GPU box gives a one-GPU job a 16-core share even where the affinity mask shows the whole host: more torch
    # threads than that only contend (measured: 64 threads 1.6e5, 16 threads 2.4e5 ray-samples/s)
    full = max(1, min(share, 16))
    v_full, n_full = leg(full)
    out = dict(value=v_full, unit="ray-samples/s", cores=full, kind="port",
               sample=f"median of {n_full} fp32 train steps (fwd+bwd+Adam/AdamW) of {Rc} rays x {S} samples, after 1 warm-up "
                      "(oracle/ref_cpu.py on torch CPU), same scene/config")
    if full != 8 and share >= 8:
        v8, n8 = leg(8)
        out["value_8_threads"] = v8
        out["sample"] += f"; value_8_threads: median of {n8} steps on 8 threads"
    return out


def dropin_leg(dev, batches, mn, sig, S, steps, warmup, total_steps, precision, optim=None, T=2 ** 16):
    """The boundary BASELINE.json's north_star names: the loop body of /root/reference/train_hash2.py:211-234 written
    against the drop-in classes - Volume_Renderer.vol_render under autocast, MSE(Cr)+MSE(Cf), loss.backward(),
    torch.optim.Adam(lr .05) on encoder.Embedding_list / AdamW(lr .005) on DataParallel(MLP_3D), two
    CosineAnnealingLR, zero_grad(set_to_none=True) - on the same resident 16 000-ray batches as the fused step.
    (bf16 autocast needs no GradScaler; the reference's scaler is an fp16 device.)  `optim`: the module the two
    optimisers come from - torch.optim (the reference's call; default) or hbr_amd.optim (same interface on the fused
    kernel, the optional sixth import line of INTEGRATION.md route A).  Returns (ms per step, last loss)."""
    import torch
    optim = optim or torch.optim
    from hbr_amd.trainer import build_default_model
    from hbr_amd.vol_renderer import Volume_Renderer
    enc, denc, mlp = build_default_model(mn, sig, dev, T=T, seed=0)                             # train_hash2.py:120-127
    nerf = torch.nn.DataParallel(mlp, device_ids=[dev.index])
    vr = Volume_Renderer(H=800, W=800, K=torch.eye(3), near=2.0, far=6.0, device=dev, Pos_encode=enc, Dir_encode=denc,
                         max_dim=2 ** 10, sigma_val=sig.to(dev), mu=mn.to(dev))                  # :124-126
    oe = optim.Adam(enc.Embedding_list.parameters(), lr=0.05)                                   # :141
    om = optim.AdamW(nerf.parameters(), lr=0.005)                                               # :142
    se = torch.optim.lr_scheduler.CosineAnnealingLR(oe, T_max=total_steps, eta_min=1e-4)        # :156-159
    sm = torch.optim.lr_scheduler.CosineAnnealingLR(om, T_max=total_steps, eta_min=1e-4)        # :160-162
    crit = torch.nn.MSELoss()                                                                   # :177

    def step(i):
        o, d, dn, gt = batches[i % len(batches)]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=precision == "bf16"):         # :218
            Cr, Cf, _ = vr.vol_render(nerf, d, o, num_samples=S, update_mask=False, dir_norm=dn, hierarchical=False)  # :220
            loss = crit(Cr, gt) + crit(Cf, gt)                                                  # :221
        loss.backward()                                                                         # :226
        oe.step(); om.step()                                                                    # :227-228
        se.step(); sm.step()                                                                    # :231-232
        om.zero_grad(set_to_none=True); oe.zero_grad(set_to_none=True)                          # :233-234
        return loss

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, float(loss.detach())


def _ev_ms(pairs):
    return sum(a.elapsed_time(b) for a, b in pairs) / max(1, len(pairs))


def hierarchical_leg(dev, batches, mn, sig, S, steps, warmup, total_steps, precision, T=2 ** 16):
    """The other training entry point: the reference's loop body with --hierarchical (train_hash2.py:34,220): the coarse
    pass's weights drive hbr_hierarchical_resample, and a second vol_render pass evaluates the 2S merged per-ray depths;
    loss = MSE(Cr) + MSE(Cf).  Drop-in classes + autograd + hbr_amd.optim, same resident batches.  Returns ms/step and
    the HIP-event spans of its three phases (render = both passes, backward, optimiser)."""
    import torch
    import hbr_amd.optim as fused_optim
    from hbr_amd.trainer import build_default_model
    from hbr_amd.vol_renderer import Volume_Renderer
    enc, denc, mlp = build_default_model(mn, sig, dev, T=T, seed=0)
    nerf = torch.nn.DataParallel(mlp, device_ids=[dev.index])
    vr = Volume_Renderer(H=800, W=800, K=torch.eye(3), near=2.0, far=6.0, device=dev, Pos_encode=enc, Dir_encode=denc,
                         max_dim=2 ** 10, sigma_val=sig.to(dev), mu=mn.to(dev))
    oe = fused_optim.Adam(enc.Embedding_list.parameters(), lr=0.05)
    om = fused_optim.AdamW(nerf.parameters(), lr=0.005)
    se = torch.optim.lr_scheduler.CosineAnnealingLR(oe, T_max=total_steps, eta_min=1e-4)
    sm = torch.optim.lr_scheduler.CosineAnnealingLR(om, T_max=total_steps, eta_min=1e-4)
    crit = torch.nn.MSELoss()
    spans = {"render_2_passes": [], "backward": [], "optimiser": []}

    def step(i, timed):
        o, d, dn, gt = batches[i % len(batches)]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timed else None
        if timed: ev[0].record()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=precision == "bf16"):
            Cr, Cf, _ = vr.vol_render(nerf, d, o, num_samples=S, update_mask=False, dir_norm=dn, hierarchical=True)
            loss = crit(Cr, gt) + crit(Cf, gt)
        if timed: ev[1].record()
        loss.backward()
        if timed: ev[2].record()
        oe.step(); om.step(); se.step(); sm.step()
        om.zero_grad(set_to_none=True); oe.zero_grad(set_to_none=True)
        if timed:
            ev[3].record()
            for k, name in enumerate(spans):
                spans[name].append((ev[k], ev[k + 1]))
        return loss

    for i in range(warmup):
        step(i, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(i, i % 4 == 0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return ms, {k: _ev_ms(v) for k, v in spans.items()}, float(loss.detach())


def render_leg(dev, batch, enc, mlp, precision, fdt, reps, S=256):
    """The image-write loop's body (train_hash2.py:277-292): 16 000-ray chunks at 256 samples under no_grad -
    hbr_render_fwd (direction encoding, K1, K3, K5 enqueued by one library call).  Returns ms per chunk through that call,
    and the HIP-event times of the same four kernels issued one by one."""
    import torch
    from hbr_amd import ops
    from hbr_amd._lib import PLANAR
    o, d, dn, _ = batch
    R = o.shape[0]
    geom, tables = enc.geometry(), enc.stacked_tables()
    flat, _ = mlp.flat_params()
    t = ops.strat_sample(2.0, 6.0, S, dev, seed=7, offset=0)
    with torch.no_grad():
        for _ in range(3):
            ops.render_fwd(geom, tables, flat, o, d, t, dn, precision=precision, feat_dtype=fdt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.render_fwd(geom, tables, flat, o, d, t, dn, precision=precision, feat_dtype=fdt)
        e1.record()
        torch.cuda.synchronize()
        whole = e0.elapsed_time(e1) / reps
        spans = {"dir_encode": [], "hash_fwd": [], "mlp_fwd": [], "composite_fwd": []}
        for _ in range(reps):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            ev[0].record()
            pe = ops.dir_encode(d, 4)
            ev[1].record()
            feat = ops.hash_encode_fwd(geom, tables, rays=(o, d, t), layout=PLANAR, dtype=fdt)
            ev[2].record()
            out = ops.mlp_fwd(feat, PLANAR, pe, S, flat, precision)
            ev[3].record()
            ops.composite_fwd(t, out.data_ptr(), 4, out.data_ptr() + 12, 4, dn, R, S, want_wts=False)
            ev[4].record()
            for k, name in enumerate(spans):
                spans[name].append((ev[k], ev[k + 1]))
        torch.cuda.synchronize()
    return whole, {k: _ev_ms(v) for k, v in spans.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rays", type=int, default=16000, help="rays per rank per step (train_hash2.py:27)")
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--hash-size", type=int, default=16, help="log2 of the rows per level (train_hash2.py:36 --hash_size; BASELINE config 2: 16)")
    ap.add_argument("--feat-dtype", default="auto", choices=["auto", "f32", "bf16"],
                    help="storage of the feature / feature-gradient buffers between the hash and MLP kernels (auto: the MLP precision)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong", "both"],
                    help="weak: --rays per rank; strong: --rays in total, split over the ranks; both: one JSON line each")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the reference-loop-on-drop-in-classes leg (N=1 only)")
    ap.add_argument("--only-dropin", action="store_true", help="profiling aid: run only the drop-in leg and print its ms/step")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Bare `python bench.py --gpus N`: become the launcher.  Nothing has touched the GPU yet (no torch import, no
        # HIP call), the ranks are CHILD processes, and this process only relays their output and exit code.
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    import torch
    from hbr_amd import _lib, synthetic
    from hbr_amd.trainer import HashNeRFTrainer, build_default_model

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the product path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev  # (rehearsals with more ranks than GPUs share a device; the real run has one GPU per rank)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("HBR_DIST_BACKEND", "nccl")  # nccl == RCCL on ROCm; gloo only for 1-GPU rehearsals
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
    # One-GPU RCCL rehearsal (HBR_RCCL_REHEARSAL=1, world == 1 only): a one-rank RCCL group, the half-level scatter
    # launches and the three staged all-reduces issued as in the multi-GPU step - what the step pays for the stream
    # hand-offs and the split before any byte crosses xGMI.  The line says so in `config.rccl_rehearsal`.
    rehearsal = world == 1 and os.environ.get("HBR_RCCL_REHEARSAL") == "1"
    if rehearsal:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert _lib.lib().hbr_device_ok() == 1, "not a gfx950 device"

    def run(scaling):
        S = args.samples
        R = args.rays if scaling == "weak" else args.rays // world  # rays per rank per step
        if R < 1:
            raise SystemExit("--scaling strong: fewer rays than ranks")
        # ---- synthetic lego-shaped workload, resident in HBM ---------------------------------
        # a pool of pre-shuffled ray batches per rank; the bbox comes from a fixed seed so every rank agrees on it
        o0, d0, _, _ = synthetic.hemisphere_rays(65536, seed=0)
        mn, mx, sig = synthetic.ray_bbox(o0, d0, 2.0, 6.0)
        pool = 8
        batches = []
        for b in range(pool):
            # rays toward the object from the upper hemisphere; ground truth = the analytic solid of synthetic.solid_field
            # composited on a fine quadrature (a consistent radiance field, so the loss/PSNR of the run mean something)
            o, d, dn, gt = synthetic.scene_rays(R, seed=1000 + rank * pool + b, device=dev)
            batches.append(tuple(a.contiguous() for a in (o, d, dn.reshape(-1), gt)))
        T = 2 ** args.hash_size
        enc, denc, mlp = build_default_model(mn, sig, dev, T=T, seed=0)  # same init on every rank (replicated parameters)
        prec = _lib.BF16 if args.precision == "bf16" else _lib.F32
        fdt = prec if args.feat_dtype == "auto" else (_lib.BF16 if args.feat_dtype == "bf16" else _lib.F32)
        total_steps = 4000 * 1000  # train_hash2.py:156-157: epochs * len(loader) (1000 epochs x 4000 batches of the 64M lego rays)
        tr = HashNeRFTrainer(enc, mlp, near=2.0, far=6.0, num_samples=S, total_steps=total_steps, precision=prec, feat_dtype=fdt,
                             overlap_comm=os.environ.get("HBR_OVERLAP_COMM", "1") != "0",  # A/B switch for the staged all-reduce
                             split_scatter=True if rehearsal and os.environ.get("HBR_OVERLAP_COMM", "1") != "0" else None)
        tr.always_reduce = rehearsal
        if os.environ.get("HBR_SPLIT_LEVEL"):
            tr.split_level = int(os.environ["HBR_SPLIT_LEVEL"])
        torch.manual_seed(1234)  # identical jitter t[S] on every rank

        def sync():
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()

        if args.only_dropin:
            ms, dl = dropin_leg(dev, batches, mn, sig, S, args.steps, args.warmup, total_steps, args.precision, T=T)
            print(json.dumps({"dropin_ms_per_step": ms, "dropin_value": R * S / (ms * 1e-3), "loss": dl, "steps": args.steps}), flush=True)
            return
        # world > 1 and no explicit HBR_OVERLAP_COMM: the trainer measures the one-collective and the staged step on
        # this node and keeps the faster (same bits either way); its steps are ordinary training steps, before warm-up
        tune = None
        if (world > 1 or rehearsal) and "HBR_OVERLAP_COMM" not in os.environ:
            tune = tr.autotune_comm(lambda i: batches[i % pool])
        print(f"[bench] rank {rank}/{world}: model + {pool} batches resident, warming up", file=sys.stderr, flush=True)
        for i in range(args.warmup):
            tr.step(*batches[i % pool])
        sync()
        # per-kernel HIP events on the launch stream, inside the timed region, on every 4th timed step: eight event
        # records per step cost 25-30 us of the 1.25 ms when taken on every step (measured A/B), and the step's `value`
        # is what this loop's wall clock says
        timers = None if args.no_kernel_events else {}
        t0 = time.perf_counter()
        for i in range(args.steps):
            tr.timers = timers if (timers is not None and i % 4 == 0) else None
            tr.step(*batches[i % pool])
        tr.timers = timers
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            dt = float(tmax.item())
        print(f"[bench] timed {args.steps} steps in {dt:.3f}s", file=sys.stderr, flush=True)
        loss = float(tr.last_loss.item())
        samples = R * S * world * args.steps
        value = samples / dt

        # ---- per-kernel durations from the HIP events recorded on the launch stream --------------------------
        kern = {}
        if tr.timers:
            for name, evs in tr.timers.items():
                kern[name] = sum(a.elapsed_time(b) for a, b in evs) / len(evs)  # ms
        N = R * S
        roofs = {}
        fb = 2 if fdt == _lib.BF16 else 4  # bytes per stored feature / feature-gradient element
        comm_ms = kern.pop("allreduce_exposed", None)  # world > 1: time the compute stream waits for the collective
        if kern:
            roofs["hash_fwd"] = dict(bound="hbm", achieved=hash_fwd_bytes(fb) * N / (kern["hash_fwd"] * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
            roofs["hash_bwd"] = dict(bound="hbm", achieved=hash_bwd_bytes(fb) * N / (kern["hash_bwd"] * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
            if "mlp_fwd" in kern:  # (absent when the step runs hbr_mlp_render_bwd: the forward is part of the backward launch)
                roofs["mlp_fwd"] = dict(bound="mfma", achieved=MLP_FWD_FLOP * N / (kern["mlp_fwd"] * 1e-3) / 1e12, peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s")
            roofs["mlp_bwd"] = dict(bound="mfma", achieved=MLP_BWD_FLOP * N / (kern["mlp_bwd"] * 1e-3) / 1e12, peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s")
            if "mlp_fwd" not in kern:
                # hbr_mlp_render_bwd: ONE launch does the MLP forward (once - no separate forward, no recompute), the compositing,
                # the loss and the MLP backward: the same 3 x 27 904 FLOP per sample are now ALL the MLP arithmetic of the step
                roofs["mlp_bwd"]["covers"] = ("MLP forward + alpha compositing + loss + MLP backward in one launch (hbr_mlp_render_bwd); "
                                             "the step has no separate mlp_fwd / compositing launches")
            # Stored profile numbers (rocprofv3 --pmc passes cannot run inside this process): HBM bytes per launch and the
            # issue-side counters, each labelled with the profile it came from so that a stale file cannot pass as live.
            pmc, issue = {}, {}
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    pmc = json.load(f)
                with open(os.path.join(ROOT, "profiles", "pmc_issue.json")) as f:
                    issue = json.load(f)
            except Exception:
                pass
            # the stored counters belong to one build of the kernels: say so when the running sources are not that build
            live_sha = _lib.kernel_source_sha()
            stale = {"traffic": pmc.get("kernel_source_sha") != live_sha, "issue": issue.get("kernel_source_sha") != live_sha}
            # HBM bytes a launch cannot avoid at this size: K1 tables once + the feature write; K2 the feature-gradient read,
            # the gradient-table write and the normalised coordinates (written once); K3/K4 their streams
            compulsory = {"hash_fwd": 16 * T * 8 + N * 32 * fb, "hash_bwd": N * 32 * fb + 16 * T * 8 + N * 12,
                          "mlp_fwd": N * 32 * fb + N * 16,
                          # (the one-launch render + backward reads the features and writes their gradient; no [N,4] out / d out)
                          "mlp_bwd": 2 * N * 32 * fb + (2 * N * 16 if "mlp_fwd" in kern else 0)}
            for k, r in roofs.items():
                r["frac"] = r["achieved"] / r["peak"]
                r["kernel"] = k
                r["avg_ms"] = kern[k]
                same_shape = (R, S, fb, T) == (16000, 128, 2, 2 ** 16)  # the stored profiles are of the default configuration
                r["traffic"] = pmc.get(k) if same_shape else None
                r["traffic_source"] = pmc.get("source", "profiles/pmc_traffic.json") + " (stored rocprofv3 --pmc profile of this configuration, not measured in this run)" if r["traffic"] else None
                if r["bound"] == "hbm":
                    # `frac` prices SURVEY 8d's ALGORITHMIC bytes (every gathered / scattered table row counted as memory
                    # traffic).  The tables live in the L2s, so this is not an HBM fraction and can exceed 1; the HBM
                    # fraction proper and the resource that does bound the kernel follow.
                    r["frac_basis"] = "algorithmic bytes (SURVEY 8d) / 8 TB/s - NOT HBM bytes: the hash tables are L2-resident"
                    if r["traffic"]:
                        r["hbm_frac"] = r["traffic"] / (kern[k] * 1e-3) / 1e9 / HBM_PEAK_GBS
                        r["traffic_vs_compulsory"] = r["traffic"] / compulsory[k]
                    if r["frac"] > 1:
                        r["frac_note"] = "above 1: the byte model is not a ceiling for a cache-resident table; see hbm_frac and roofline_issue"
                if same_shape and k in issue:
                    r["roofline_issue"] = issue[k]
                    r["roofline_issue_stale"] = stale["issue"]
                if r["traffic"]:
                    r["traffic_source_stale"] = stale["traffic"]
        dominant = max(kern, key=kern.get) if kern else None

        # ---- the reference's own loop on the drop-in classes (the boundary north_star names), same batches ---------
        dropin = None
        if rank == 0 and world == 1 and not args.no_dropin:
            ms, dl = dropin_leg(dev, batches, mn, sig, S, args.steps, args.warmup, total_steps, args.precision, T=T)
            import hbr_amd.optim as fused_optim
            ms2, _ = dropin_leg(dev, batches, mn, sig, S, args.steps, args.warmup, total_steps, args.precision, optim=fused_optim, T=T)
            dropin = dict(ms_per_step=ms, value=R * S / (ms * 1e-3), loss=dl, ms_fused_optim=ms2)
            print(f"[bench] drop-in loop: {ms:.3f} ms/step (torch.optim), {ms2:.3f} ms/step (hbr_amd.optim)", file=sys.stderr, flush=True)

        # ---- the other two entry points (N = 1 only): --hierarchical training and the image-write render -----------
        extras = {}
        if rank == 0 and world == 1 and not args.no_dropin:
            hs = max(8, args.steps // 2)
            hms, hspans, hloss = hierarchical_leg(dev, batches, mn, sig, S, hs, max(3, args.warmup // 2), total_steps, args.precision, T=T)
            rms, rspans = render_leg(dev, batches[0], enc, mlp, prec, fdt, reps=max(5, args.steps // 5))
            extras = {"hierarchical_ms_per_step": hms, "hierarchical_value": R * 2 * S / (hms * 1e-3), "hierarchical_phases_ms": hspans,
                      "hierarchical_note": f"train_hash2.py --hierarchical on the drop-in classes + hbr_amd.optim: coarse pass of {S} + fine pass of {2 * S} "
                                           "merged depths per ray, loss = MSE(Cr) + MSE(Cf); value counts the fine pass's 2S samples per ray",
                      "render_ms_per_16k_rays_256": rms * 16000 / R, "render_value": R * 256 / (rms * 1e-3), "render_kernels_ms": rspans,
                      "render_note": f"hbr_render_fwd on {R} rays x 256 samples under no_grad (the image-write loop body, train_hash2.py:277-292); "
                                     "render_kernels_ms: the same four kernels issued one by one"}
            print(f"[bench] hierarchical loop: {hms:.3f} ms/step; render 16k x 256: {rms:.3f} ms", file=sys.stderr, flush=True)

        # ---- CPU baseline: the oracle (a port of the reference's PyTorch-CPU path) on a bounded sample ---------
        cpu = None
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(S, mn, sig, total_steps)

        line = None
        if rank == 0:
            line = {
                "metric": "ray-samples/sec @128 samples/ray on lego (full train step: fwd+bwd+optimiser)",
                "value": value, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
                "dtype": "bf16" if prec == _lib.BF16 else "f32", "data": "synthetic",
                "config": {"workload": f"lego-shaped synthetic scene, hash encoding L=16 F=2 T=2^{args.hash_size} N_min=16 N_max=2048, "
                                       f"{R} rays/rank x {S} samples/ray ({scaling} scaling), MLP 32-64-64-16 / 39-64-64-3, fp32 tables, "
                                       f"{'bf16' if prec == _lib.BF16 else 'fp32'} MLP (MFMA), "
                                       f"{'bf16' if fdt == _lib.BF16 else 'fp32'} feature/feature-gradient buffers, Adam+AdamW+cosine",
                           "rays_per_rank": R, "samples_per_ray": S, "global_rays": R * world, "levels": 16, "table_rows": T,
                           "parallelism": f"ray-sharded dp{world}, 1 all-reduce/step" if world > 1 else "single GPU"},
                "loss": loss,
                # the step's MLP forward + compositing + loss + backward as ONE launch (hbr_mlp_render_bwd; then no mlp_fwd entry in `kernels`)
                "fused_render": bool(tr.timers is not None and "mlp_fwd" not in tr.timers) if not args.no_kernel_events else None,
                "roofline": roofs.get(dominant), "roofline_hash_lookup": roofs.get("hash_fwd"), "kernels": roofs or None,
                "cpu_baseline": cpu,
                # multi-GPU diagnostics: the one all-reduce per step of the flat 8.05 MiB gradient buffer
                "allreduce_exposed_ms": comm_ms, "overlap_comm": bool(tr.overlap_comm), "split_scatter": bool(tr.split_scatter),
                "allreduce_bytes": int(tr.grad.numel() * 4) if (world > 1 or rehearsal) else 0,
                "dist_backend": (torch.distributed.get_backend() if (world > 1 or rehearsal) else None),
                "rccl_rehearsal": rehearsal, "comm_autotune": tune,
                # train_hash2.py:211-234 as written (vol_render + autograd + torch.optim), drop-in classes, same batches
                "dropin_ms_per_step": dropin and dropin["ms_per_step"], "dropin_value": dropin and dropin["value"],
                # the same loop with `from hbr_amd.optim import Adam, AdamW` in place of torch.optim's (same interface, fused kernel)
                "dropin_fused_optim_ms_per_step": dropin and dropin["ms_fused_optim"],
                **extras,
            }
            print(json.dumps(line), flush=True)
        return line

    for sc in (("weak", "strong") if args.scaling == "both" else (args.scaling,)):
        run(sc)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
