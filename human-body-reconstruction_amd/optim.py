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
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            self._state_for(group)
            # step counters: one tensor shared by the group's parameters while they move together (one add per step
            # instead of one per tensor; state_dict() shows it under every parameter, as torch's does with its own)
            st0 = self.state[ps[0]]["step"]
            if all(self.state[p]["step"] is st0 for p in ps[1:]):
                st0 += 1
            else:
                if len(ps) == len(group["params"]) and all(float(self.state[p]["step"]) == float(st0) for p in ps[1:]):
                    for p in ps[1:]:
                        self.state[p]["step"] = st0   # equal counters (fresh state, or a loaded state_dict): share from now on
                    st0 += 1
                else:
                    for p in ps:
                        self.state[p]["step"] += 1
            k = int(st0)
            b1, b2 = group["betas"]
            common = dict(lr=float(group["lr"]), beta1=b1, beta2=b2, eps=group["eps"], weight_decay=group["weight_decay"], step=k)
            fast = self._fast_views(gi, group, ps) if all(int(self.state[p]["step"]) == k for p in (ps[0], ps[-1])) else None
            flat_g = self._flat_grads(ps, fast) if fast is not None else None
            if flat_g is not None:
                segs.append(dict(p=fast[0], g=flat_g, m=fast[1], v=fast[2], **common))
            else:
                for p in ps:
                    if not p.is_cuda or p.dtype != torch.float32 or p.grad.is_sparse:
                        raise ops.HbrError("hbr_amd.optim needs dense float32 parameters on the MI355X")
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
        # the kernel wrote through raw pointers: move every parameter's version counter as torch's in-place update would
        # (a flat one-launch segment only carries the FIRST tensor's counter)
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    torch.autograd.graph.increment_version(p)
        return loss

    def _fast_views(self, gi, group, ps):
        """(flat parameters, flat exp_avg, flat exp_avg_sq, element offsets) of a group whose tensors and moments are
        consecutive views of one buffer each - looked up once and kept while the first and last tensors stay where they
        were (a .to() / load_state_dict that moves storage drops the entry)."""
        if len(ps) != len(group["params"]):
            return None
        cache = self.__dict__.setdefault("_fast", {})
        key = (ps[0].data_ptr(), ps[-1].data_ptr(), self.state[ps[0]]["exp_avg"].data_ptr(), self.state[ps[-1]]["exp_avg_sq"].data_ptr())
        hit = cache.get(gi)
        if hit is not None and hit[0] == key:
            return hit[1]
        flat_p = _consecutive(ps)
        flat_m = _consecutive([self.state[p]["exp_avg"] for p in ps]) if flat_p is not None else None
        flat_v = _consecutive([self.state[p]["exp_avg_sq"] for p in ps]) if flat_m is not None else None
        views = None
        if flat_v is not None:
            offs, off = [], 0
            for p in ps:
                offs.append(off)
                off += p.numel() * 4
            views = (flat_p, flat_m, flat_v, offs)
        cache[gi] = (key, views)
        return views

    @staticmethod
    def _flat_grads(ps, fast):
        """One flat view over the gradients if they are laid out exactly like the parameters IN ONE STORAGE (what
        ops.RenderFn hands back and AccumulateGrad adopts), else None.  `_consecutive` checks the storage as well as the
        addresses: separately allocated gradients (p.grad.clone(), a regulariser's gradient that reached AccumulateGrad
        first) can sit back to back in the caching allocator's arena without sharing one."""
        if any(p.grad.numel() != p.numel() for p in ps):
            return None
        flat = _consecutive([p.grad for p in ps])
        return flat if flat is not None and flat.numel() == fast[0].numel() else None

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
