"""torch.optim-compatible Adam / AdamW on the fused gfx950 kernel.

The reference steps `torch.optim.Adam(encoder.Embedding_list.parameters(), lr=0.05)` and
`torch.optim.AdamW(nerf.parameters(), lr=0.005)` (train_hash2.py:141-142,227-228); on the GPU those are 15
`multi_tensor_apply` launches per step (0.15 ms of the drop-in loop's 1.46 ms).  These classes take the same arguments,
keep the same `state_dict()` layout (`step`, `exp_avg`, `exp_avg_sq` per parameter; `param_groups` with lr / betas / eps /
weight_decay) and work with `torch.optim.lr_scheduler.*`, but update a whole parameter group with ONE launch of
`hbr_adam_step_multi` when the group's parameters - and their gradients - are consecutive views of one buffer, which is
what HashEncoder / MLP_3D parameters and the gradients ops.RenderFn hands back are.  Anything else (gradients that
autograd accumulated into separate tensors, a parameter without a gradient) takes one segment per tensor, four segments
per launch.  Same arithmetic as torch's single-tensor Adam (tests/test_gpu_optim.py: one step bit-for-bit within 2 ulp,
many steps within 1e-6).

Swap-in for a maintainer of the reference:    from hbr_amd.optim import Adam, AdamW
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch

from . import ops


def _consecutive(tensors: List[torch.Tensor]) -> Optional[torch.Tensor]:
    """One flat fp32 view covering `tensors` if they are contiguous and laid out back to back in one storage, else None."""
    if not tensors:
        return None
    first = tensors[0]
    if first.dtype != torch.float32 or not first.is_cuda:
        return None
    end = first.data_ptr()
    total = 0
    for t in tensors:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() != end or t.untyped_storage().data_ptr() != first.untyped_storage().data_ptr():
            return None
        end += t.numel() * 4
        total += t.numel()
    if first.data_ptr() % 16:
        return None
    return torch.as_strided(first, (total,), (1,))  # same storage, from the first tensor's offset


class _FusedAdam(torch.optim.Optimizer):
    _decoupled = False

    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        if weight_decay != 0 and not self._decoupled:
            raise NotImplementedError("L2 weight decay inside Adam is not implemented by the fused kernel: use AdamW (decoupled) or weight_decay=0")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def _state_for(self, group):
        """exp_avg / exp_avg_sq of a group live in ONE flat buffer when its parameters do, so that the moments of
        consecutive parameters are consecutive too; the per-parameter state tensors are views of it."""
        ps = [p for p in group["params"]]
        missing = [p for p in ps if len(self.state[p]) == 0]
        if not missing:
            return
        flat = _consecutive(ps) if len(missing) == len(ps) else None
        if flat is not None:
            m, v = torch.zeros_like(flat), torch.zeros_like(flat)
            off = 0
            for p in ps:
                n = p.numel()
                self.state[p].update(step=torch.tensor(0.0), exp_avg=m[off:off + n].view_as(p), exp_avg_sq=v[off:off + n].view_as(p))
                off += n
        else:
            for p in missing:
                self.state[p].update(step=torch.tensor(0.0), exp_avg=torch.zeros_like(p, memory_format=torch.contiguous_format),
                                     exp_avg_sq=torch.zeros_like(p, memory_format=torch.contiguous_format))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        segs = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.is_sparse:
                    raise ops.HbrError("hbr_amd.optim needs dense float32 parameters on the MI355X")
            self._state_for(group)
            for p in ps:
                self.state[p]["step"] += 1
            k = int(self.state[ps[0]]["step"])
            b1, b2 = group["betas"]
            common = dict(lr=float(group["lr"]), beta1=b1, beta2=b2, eps=group["eps"], weight_decay=group["weight_decay"], step=k)
            same_step = all(int(self.state[p]["step"]) == k for p in ps)
            whole = len(ps) == len(group["params"]) and same_step
            flat_p = _consecutive(ps) if whole else None
            flat_g = _consecutive([p.grad for p in ps]) if flat_p is not None else None
            flat_m = _consecutive([self.state[p]["exp_avg"] for p in ps]) if flat_g is not None else None
            flat_v = _consecutive([self.state[p]["exp_avg_sq"] for p in ps]) if flat_m is not None else None
            if flat_v is not None:
                segs.append(dict(p=flat_p, g=flat_g, m=flat_m, v=flat_v, **common))
            else:
                for p in ps:
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    if not p.is_contiguous():
                        raise ops.HbrError("hbr_amd.optim needs contiguous parameters")
                    st = self.state[p]
                    one = dict(p=p.view(-1), g=g.view(-1), m=st["exp_avg"].view(-1), v=st["exp_avg_sq"].view(-1), **common)
                    one["step"] = int(st["step"])
                    if any(t.data_ptr() % 16 for t in (one["p"], one["g"], one["m"], one["v"])):  # the kernel loads 16-byte vectors
                        self._unaligned(one)
                    else:
                        segs.append(one)
        for i in range(0, len(segs), 4):
            ops.adam_step_multi(segs[i:i + 4])
        return loss

    @staticmethod
    def _unaligned(seg):
        """A tensor that does not start on a 16-byte boundary (a view at an odd offset): aligned copies in, results out."""
        p, g, m, v = (seg[k].clone() for k in ("p", "g", "m", "v"))
        ops.adam_step_multi([dict(seg, p=p, g=g, m=m, v=v)])
        seg["p"].copy_(p); seg["m"].copy_(m); seg["v"].copy_(v)


class Adam(_FusedAdam):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay=0) on the fused kernel (train_hash2.py:141)."""
    _decoupled = False


class AdamW(_FusedAdam):
    """torch.optim.AdamW(params, lr, betas, eps, weight_decay=0.01) on the fused kernel (train_hash2.py:142)."""
    _decoupled = True

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
