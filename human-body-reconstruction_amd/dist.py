"""Multi-GPU plumbing: one process per GPU, rays sharded, parameters replicated, ONE all-reduce per step.

The reference's only parallelism is single-process torch.nn.DataParallel around the MLP (train_hash2.py:127); it is
replaced, not reproduced.  The data path has no exchange step other than the gradient all-reduce:
each rank renders R/P rays with its own replica, computes loss_p = mean over its shard, and
grad = (1/P) * sum_p grad_p equals the single-process gradient of the mean over all R rays (equal shards).
Backend "nccl" is RCCL on ROCm (xGMI between the 8 GPUs of a node); tests use "gloo" on CPU.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, device: Optional[torch.device] = None) -> Tuple[int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run sets
    them).  Returns (rank, world).  No-op for world == 1."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n rays for `rank`; shards are equal-sized (n is truncated to a multiple of
    world so every rank's mean-loss carries the same weight)."""
    per = n // world
    return rank * per, (rank + 1) * per


def shard_batch(batch, rank: int, world: int):
    lo, hi = shard_bounds(batch[0].shape[0], rank, world)
    return tuple(b[lo:hi] for b in batch)


def allreduce_mean_(flat: torch.Tensor, world: int, group=None, scale_in_place: bool = True) -> torch.Tensor:
    """Sum `flat` over ranks (one collective over the single flat gradient buffer) and, unless the caller folds the
    1/world factor into its optimiser kernel, scale it."""
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if scale_in_place:
            flat.mul_(1.0 / world)
    return flat


def allreduce_mean_grads_(grads, world: int, group=None) -> None:
    """Average a list of gradient tensors over the ranks with ONE all-reduce: they are packed into a single flat
    buffer (8.05 MiB for the 16 tables + 12 MLP tensors), reduced, and copied back - not one collective per tensor."""
    if world <= 1 or not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / world)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


class StagedAllReduce:
    """The step's gradient all-reduce issued in pieces, each as soon as its slice of the flat buffer is final, so that
    all but the last piece overlap the kernels still producing the rest (RCCL runs them on its own stream; `launch`
    orders a piece after everything already enqueued on the caller's current stream, `finish` makes the current stream
    wait for all of them).  The pieces partition the buffer, so the result is bit-identical to one collective over
    the whole buffer.  With world == 1 every call is a no-op, unless `force` (the one-GPU RCCL rehearsal of
    tests/test_gpu_rccl_world1.py: a one-rank all-reduce leaves the data as it is but takes the same stream hand-offs)."""

    def __init__(self, world: int, group=None, force: bool = False):
        self.world, self.group, self._works = world, group, []
        self.active = world > 1 or (force and dist.is_initialized())

    def launch(self, piece: torch.Tensor) -> None:
        if self.active and piece.numel():
            self._works.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self) -> None:
        for w in self._works:
            w.wait()
        self._works = []


def broadcast_params_(tensors, src: int = 0, group=None) -> None:
    """Make replicas identical at start-up (the reference relies on DataParallel's replicate each forward)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in tensors:
            dist.broadcast(t, src=src, group=group)
