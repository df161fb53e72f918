"""The train_hash2.py loop (reference train_hash2.py:106-306) restated on the gfx950 pipeline.

One `HashNeRFTrainer.step(batch)` = one iteration of the reference's loop body (:211-239):
    t = strat_sampler(near, far, S)                       helper.py:234-235
    Cr, Cf, _ = vol_render(..., hierarchical=False)       vol_renderer.py:141-245
    loss = MSE(Cr, gt) + MSE(Cf, gt)                      train_hash2.py:177,221   (= 2*MSE, Cf is Cr)
    loss.backward()                                       :226
    Adam(lr .05) on the tables, AdamW(lr .005) on the MLP :141-142,227-228
    cosine annealing of both learning rates               :156-162,231-232
without going through autograd: forward kernels, the loss kernel, three backward kernels, one optional
RCCL all-reduce of a single flat gradient buffer, two fused Adam launches.  bf16 needs no GradScaler.

Multi-GPU: rays are sharded over ranks (one process per GPU), parameters are replicated, and
the only collective is ONE all-reduce(sum) per step over [d tables | d MLP] (8.05 MiB fp32); the 1/world
factor is folded into the Adam kernel's grad_scale.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _lib, ops
from .dist import StagedAllReduce
from ._lib import BF16, F32, MLP_PARAM_FLOATS, PLANAR
from .encoder import PositionalEncoder
from .hash_encoding import HashEncoder
from .test_hash import MLP_3D


def cosine_lr(base_lr: float, eta_min: float, step: int, t_max: int) -> float:
    """torch.optim.lr_scheduler.CosineAnnealingLR in closed form (lr used by optimiser step `step`, 0-based)."""
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * step / t_max)) / 2.0


class HashNeRFTrainer:
    def __init__(self, encoder: HashEncoder, mlp: MLP_3D, near: float = 2.0, far: float = 6.0, num_samples: int = 128,
                 total_steps: int = 100000, lr_embed: float = 0.05, lr_mlp: float = 0.005, eta_min: float = 1e-4,
                 weight_decay_mlp: float = 0.01, precision: int = BF16, feat_dtype: Optional[int] = None, num_freq: int = 4,
                 process_group=None, scatter_algo: int = 0, overlap_comm: bool = False, split_scatter: Optional[bool] = None,
                 seed: int = 0):
        if encoder.L * encoder.F != 32 or encoder.E != 0:
            raise NotImplementedError("HashNeRFTrainer drives the train_hash2.py:107,120,127 model: L=16 levels x F=2 features into "
                                      f"MLP_3D's 32 inputs; got L={encoder.L}, F={encoder.F}, E={encoder.E}")
        self.enc, self.mlp = encoder, mlp
        self.near, self.far, self.S = float(near), float(far), int(num_samples)
        self.total_steps = int(total_steps)
        self.lr_embed, self.lr_mlp, self.eta_min, self.wd_mlp = lr_embed, lr_mlp, eta_min, weight_decay_mlp
        # Storage type of the planar feature buffer K1 -> K3/K4 and of the feature-gradient buffer K4 -> K2.  Default: the
        # MLP's precision.  In bf16 mode that is what the MLP reads anyway (its MFMA fragments are bf16, so the forward
        # is bit-identical to fp32 storage) and what the reference's autocast hands back (the gradient of the Linear
        # input is produced in half precision and cast up, train_hash2.py:218); it halves 0.8 GB of traffic per step.
        if feat_dtype is None:
            feat_dtype = precision
        self.precision, self.feat_dtype, self.num_freq = precision, feat_dtype, num_freq
        self.scatter_algo = scatter_algo
        self.seed = int(seed)
        self.overlap_comm = overlap_comm
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        # overlap_comm: the scatter kernel runs as two launches so that the first one's all-reduce overlaps the second
        # launch.  Off by default since round 3: on one MI355X with a one-rank RCCL group the staging itself costs
        # ~0.1 ms per step, more than an 8 MiB all-reduce over xGMI is expected to take - `autotune_comm`
        # measures both on the node it runs on (bench.py does that).  Tests force the split on a single GPU, where the
        # staged reduce is a no-op and the result must equal the single launch.
        self.split_scatter = (self.world > 1 and overlap_comm) if split_scatter is None else bool(split_scatter)
        self.always_reduce = False  # tests: issue the collectives with world == 1 too (RCCL stream hand-offs on one GPU)
        # first level of the FIRST scatter launch (levels [split_level, L), reduced while [0, split_level) are scattered)
        self.split_level = max(1, encoder.geometry().L // 2)
        self._bind_parameters()
        dev = self.tables.device
        self.geom = encoder.geometry()
        self.n_tab = self.tables.numel()
        n = self.n_tab + MLP_PARAM_FLOATS
        n_pad = (n + 3) // 4 * 4
        self.grad = torch.zeros(n_pad, dtype=torch.float32, device=dev)       # [d tables | d MLP | pad]
        self.m = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        self.g_tab = self.grad[:self.n_tab].view_as(self.tables)
        self.g_mlp = self.grad[self.n_tab:self.n_tab + MLP_PARAM_FLOATS]
        self._amax = torch.zeros(16, dtype=torch.float32, device=dev)
        self.step_count = 0
        self.last_loss = None
        self.timers = None  # optional dict name -> list[(start_event, end_event)], filled when set by bench.py
        self.grad_hook = None  # optional callable(flat gradient buffer) run in front of the optimiser (studies: tools/psnr_converged_study.py)
        # the step's small launches folded together: prologue (depths + direction encoding + weight image), compositing +
        # loss + compositing backward, one Adam launch.  HBR_FUSED_SMALL=0: the separate launches (A/B, tests).
        import os
        self.fused_small = os.environ.get("HBR_FUSED_SMALL", "1") != "0"
        # MLP forward + compositing + loss + backward as one launch where the shape allows (HBR_FUSED_RENDER=0: the separate launches)
        self.fused_render = os.environ.get("HBR_FUSED_RENDER", "1") != "0"

    # ---- helpers ------------------------------------------------------------------------------
    def _bind_parameters(self):
        """(Re-)fetch the stacked table buffer and the flat MLP block from the modules.  Both calls are cheap pointer
        checks that re-establish the aliasing if a caller replaced parameter storages (load_state_dict into fresh
        tensors, .to() round trips), so the trainer never trains a stale copy."""
        self.tables = self.enc.stacked_tables()
        self.flat, self.splits = self.mlp.flat_params()
        if getattr(self, "grad", None) is not None and self.grad.device != self.tables.device:
            raise RuntimeError("the modules were moved to another device after the trainer was built; build a new trainer")

    def _timed(self, name, fn):
        if self.timers is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self.timers.setdefault(name, []).append((e0, e1))
        return r

    def autotune_comm(self, batch_fn, steps: int = 20) -> dict:
        """world > 1: runs `steps` training steps with the step's all-reduce as ONE collective behind the scatter kernel
        and `steps` with it staged (two half-level scatter launches, the first piece reduced under the second launch),
        takes the slower rank's time for each (one MAX all-reduce, so every rank decides alike) and keeps the faster.
        Both ways produce the same bits (tests/test_gpu_shipped_paths.py, tests/test_gpu_rccl_world1.py), so this is a
        pure scheduling choice: the staging costs ~0.1 ms per step of its own and pays only where the
        collective is slower than that.  `batch_fn(i)` -> (rays_o, rays_d, dir_norm, gt).  Returns the measurements."""
        import time
        if not (self.world > 1 or (self.always_reduce and torch.distributed.is_initialized())) or self.geom.L < 2:
            return {}
        res, dev = {}, self.tables.device
        # The measurement runs real optimiser steps: snapshot parameters, moments and the step counter (the cosine
        # schedule's position and the depth jitter's Philox offset) and put them back, so that a run tuned with
        # `--overlap_comm auto` trains exactly the trajectory of `on` / `off` with the same seed.
        self._bind_parameters()
        snap = (self.tables.clone(), self.flat.clone(), self.m.clone(), self.v.clone(), self.step_count, self.last_loss)
        for name, split in (("single", False), ("staged", True)):
            self.split_scatter = split
            for i in range(2):  # workspaces, RCCL channels for this message size
                self.step(*batch_fn(i))
            torch.cuda.synchronize(dev)
            torch.distributed.barrier(group=self.pg)
            t0 = time.perf_counter()
            for i in range(steps):
                self.step(*batch_fn(i))
            torch.cuda.synchronize(dev)
            dt = torch.tensor([(time.perf_counter() - t0) / steps * 1e3], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(dt, op=torch.distributed.ReduceOp.MAX, group=self.pg)
            res[name + "_ms_per_step"] = float(dt.item())
        with torch.no_grad():
            self.tables.copy_(snap[0]); self.flat.copy_(snap[1]); self.m.copy_(snap[2]); self.v.copy_(snap[3])
        self.step_count, self.last_loss = snap[4], snap[5]
        self.split_scatter = res["staged_ms_per_step"] < res["single_ms_per_step"]
        self.overlap_comm = self.split_scatter
        res["chosen"] = "staged" if self.split_scatter else "single"
        return res

    def sample_t(self, device) -> torch.Tensor:
        """This step's shared depths t[S]: one launch, jitter drawn on the device from (seed, step) - every rank that
        was built with the same `seed` samples the same depths, as the sharded step requires."""
        return ops.strat_sample(self.near, self.far, self.S, device, seed=self.seed, offset=self.step_count)

    # ---- one optimisation step ------------------------------------------------------------------
    def step(self, rays_o, rays_d, dir_norm, gt, t: Optional[torch.Tensor] = None):
        """rays_o/rays_d [R,3], dir_norm [R,1] or [R], gt [R,3], all resident on the GPU."""
        self._bind_parameters()
        S, g = self.S, self.geom
        R = rays_o.shape[0]
        dn = dir_norm.reshape(-1) if torch.is_tensor(dir_norm) else None
        fused = self.fused_small and self.num_freq == 4
        if fused:
            # ONE launch: this step's depths (unless given), the direction encoding, the MLP's weight-fragment image
            strat = None if t is not None else (self.near, self.far, S, None, self.seed, self.step_count)
            t_new, pe = ops.render_prologue(rays_o.device, self.precision, params=self.flat, rays_d=rays_d, strat=strat)
            t = t if t is not None else t_new
        else:
            if t is None:
                t = self.sample_t(rays_o.device)
            pe = ops.dir_encode(rays_d, self.num_freq)
        rays = (rays_o, rays_d, t)
        # forward
        feat = self._timed("hash_fwd", lambda: ops.hash_encode_fwd(g, self.tables, rays=rays, layout=PLANAR, dtype=self.feat_dtype))
        amax = self._amax  # (L == 16: checked in __init__)
        # Round 4: MLP forward + compositing + loss + their backward in ONE launch (hbr_mlp_render_bwd) where whole rays fit a
        # workgroup round of the backward kernel - S in {32, 64, 128}, bf16 MLP, per-ray dir_norm tensor or none.
        rendered = None
        if fused and self.fused_render and self.precision == BF16:
            rendered = self._timed("mlp_bwd", lambda: ops.mlp_render_bwd(feat, pe, self.flat, self.precision, t, dn, gt, self.g_mlp, absmax_out=amax,
                                                                         image_ready=True, overwrite=True))
            if rendered is None and self.timers is not None:
                self.timers["mlp_bwd"].pop()  # the library refused the shape: the span timed nothing
        if rendered is not None:
            loss, dfeat, _ = rendered
            return self._finish_step(loss, dfeat, rays, amax, g, R, S)
        out = self._timed("mlp_fwd", lambda: ops.mlp_fwd(feat, PLANAR, pe, S, self.flat, self.precision, image_ready=fused))
        if fused:
            # ONE launch: compositing, loss = 2*MSE and its gradient, compositing backward (each ray in its own wave)
            loss, d_out, _ = ops.composite_loss_fwd_bwd(t, out, dn, R, S, gt)
        else:
            Cr, _ = ops.composite_fwd(t, out.data_ptr(), 4, out.data_ptr() + 12, 4, dn, R, S, want_wts=False)
            loss, dCr = ops.mse2_loss(Cr, gt)
            d_out = torch.empty_like(out)
            ops.composite_bwd(t, out.data_ptr(), 4, out.data_ptr() + 12, 4, dn, R, S, dCr, d_out.data_ptr(), d_out.data_ptr() + 12)
        # No memset of the 8 MiB gradient buffer: K4 and K2 WRITE their outputs (`overwrite`; where K2 runs a path that
        # can only accumulate, ops zeroes that slice itself).  The padding behind the MLP block is never written.
        # K4 also reports max |d feat| per level: K2's fixed-point scale, without K2 re-reading the buffer for it
        # (image_ready: the workspace still holds the weight fragments this step's mlp_fwd packed from self.flat)
        dfeat = self._timed("mlp_bwd", lambda: ops.mlp_bwd(feat, PLANAR, pe, S, self.flat, self.precision, d_out, self.g_mlp,
                                                           absmax_out=amax, image_ready=True, overwrite=True))
        return self._finish_step(loss, dfeat, rays, amax, g, R, S)

    def _finish_step(self, loss, dfeat, rays, amax, g, R, S):
        """K2, the step's one all-reduce, the optimiser."""
        if self.split_scatter and g.L >= 2:
            # The step's one all-reduce, issued in two pieces that partition the flat gradient buffer: the upper levels
            # together with the MLP block (final after K4) while the lower levels' scatter is still running, then the
            # lower levels - the only exposed piece (4 MiB at L = 16, split_level = 8).  What the staging itself costs
            # (two launches instead of one, the stream hand-offs): DESIGN 6; `autotune_comm` measures both ways.  Planar [L,N,2] dfeat and [L,T,F] grads make a level
            # range a contiguous slice, so the same kernel runs on each half (bit-identical to the single launch:
            # tests/test_gpu_shipped_paths.py).  With world == 1 the reduce calls are no-ops.
            red = StagedAllReduce(self.world, self.pg, force=self.always_reduce)
            nt, half = self.n_tab, min(max(1, int(self.split_level)), g.L // 2)
            cut = half * g.T * g.F

            def scatter_halves():  # timed as ONE hash_bwd span (both launches), like the single-launch path
                # the second launch re-uses the first one's coordinates (algo 2 then 3; the first is the larger, so its
                # workspace fits both) - where the LDS kernels can run at all: for a shape they refuse (T > 2^28:
                # workspace size 0) both take the caller's algo, so a model that trains on one GPU does not raise on N
                # (asking with the caller's algo: 0 = auto answers 0 bytes below the size from which it picks the LDS kernels)
                lds = (self.scatter_algo in (0, 2)
                       and _lib.lib().hbr_hash_bwd_workspace_bytes(R * S, g.L - half, g.T, g.F, self.scatter_algo) > 0)
                # pieces: [upper levels | MLP block] - contiguous in the flat buffer, final once K4 and the first launch
                # are done - then the lower levels
                for k, (lo, hi, piece) in enumerate(((half, g.L, self.grad[cut:]), (0, half, self.grad[:cut]))):
                    sub = ops.HashGeom(g.scales[lo:hi], g.mu, g.sigma, g.T, g.F)
                    ops.hash_encode_bwd(sub, dfeat[lo:hi], self.g_tab[lo:hi], rays=rays, layout=PLANAR,
                                        algo=(2 if k == 0 else 3) if lds else self.scatter_algo,
                                        dy_absmax=None if amax is None else amax[lo:hi], overwrite=True)
                    red.launch(piece)

            self._timed("hash_bwd", scatter_halves)
            self._timed("allreduce_exposed", red.finish) if red.active else red.finish()
        else:
            self._timed("hash_bwd", lambda: ops.hash_encode_bwd(g, dfeat, self.g_tab, rays=rays, layout=PLANAR, algo=self.scatter_algo,
                                                                dy_absmax=amax, overwrite=True))
            # the one collective of the step
            if self.world > 1 or (self.always_reduce and torch.distributed.is_initialized()):
                self._timed("allreduce_exposed", lambda: torch.distributed.all_reduce(self.grad, op=torch.distributed.ReduceOp.SUM, group=self.pg))
        if self.grad_hook is not None:
            self.grad_hook(self.grad)
        # optimiser (dense Adam over every table row, as the reference's torch.optim.Adam does)
        k = self.step_count
        gs = 1.0 / self.world
        nt = self.n_tab
        tab = dict(p=self.tables.view(-1), g=self.grad[:nt], m=self.m[:nt], v=self.v[:nt],
                   lr=cosine_lr(self.lr_embed, self.eta_min, k, self.total_steps), weight_decay=0.0)
        mlp = dict(p=self.flat, g=self.g_mlp, m=self.m[nt:nt + MLP_PARAM_FLOATS], v=self.v[nt:nt + MLP_PARAM_FLOATS],
                   lr=cosine_lr(self.lr_mlp, self.eta_min, k, self.total_steps), weight_decay=self.wd_mlp)
        common = dict(beta1=0.9, beta2=0.999, eps=1e-8, step=k + 1, grad_scale=gs)
        if self.fused_small:  # Adam on the tables + AdamW on the MLP in ONE launch
            ops.adam_step_multi([{**tab, **common}, {**mlp, **common}])
        else:
            ops.adam_step(**tab, **common)
            ops.adam_step(**mlp, **common)
        self.step_count += 1
        self.last_loss = loss
        return loss

    # ---- inference ------------------------------------------------------------------------------
    @torch.no_grad()
    def render(self, rays_o, rays_d, dir_norm, num_samples: Optional[int] = None, t: Optional[torch.Tensor] = None, chunk: int = 16000):
        """Image-write path (train_hash2.py:277-292): 16000-ray chunks, unmasked branch."""
        S = num_samples or self.S
        if t is None:
            t = ops.strat_sample(self.near, self.far, S, rays_o.device, seed=self.seed, offset=(1 << 40) + self.step_count)
        self._bind_parameters()
        outs = []
        for i in range(0, rays_o.shape[0], chunk):
            dn = dir_norm[i:i + chunk].reshape(-1) if torch.is_tensor(dir_norm) else None
            Cr, _, _ = ops.render_fwd(self.geom, self.tables, self.flat, rays_o[i:i + chunk], rays_d[i:i + chunk], t, dn,
                                      precision=self.precision, feat_dtype=self.feat_dtype)
            outs.append(Cr)
        return torch.cat(outs)


def build_default_model(mu: torch.Tensor, sigma: torch.Tensor, device, L: int = 16, F: int = 2, T: int = 2 ** 16,
                        N_min=16, N_max=2048.0, num_freq: int = 4, seed: Optional[int] = None):
    """The objects train_hash2.py:120-127 builds (encoder, direction encoder, MLP)."""
    if seed is not None:
        torch.manual_seed(seed)
    enc = HashEncoder(N_min=N_min, N_max=N_max, L=L, F=F, T=T, dim=3, mu=mu.to(device), sigma=sigma.to(device), device=device)
    denc = PositionalEncoder(d_model=3, num_freq=num_freq)
    mlp = MLP_3D(num_sig=2, num_col=2, L=L, F=F, d_view=3 * num_freq * 2).to(device)
    return enc, denc, mlp
