"""MLP_3D: the density + colour field MLP (reference test_hash.py:20-72) on the MFMA kernels K3/K4.

Module/parameter names match the reference (`sig_model.{0,2,4}`, `col_model.{0,2,4}`), so state dicts
interchange, including the `module.` prefix when wrapped in torch.nn.DataParallel (train_hash2.py:127).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from ._lib import MLP_PARAM_FLOATS, HbrError


class MLP_3D(nn.Module):
    def __init__(self, num_sig=3, num_col=2, h_size=64, d_view=3, L=16, F=2, E=0, use_sdf=False, max_bound=1.0, min_bound=-1.0):
        super().__init__()
        self.d_view = d_view
        self.max_bound, self.min_bound = max_bound, min_bound  # only used by the SDF helpers (out of scope)
        sig = [nn.Linear(L * F + E, h_size), nn.ReLU()]
        for i in range(num_sig):
            if i == num_sig - 1:
                sig.append(nn.Linear(h_size, 1 + 15))
            else:
                sig += [nn.Linear(h_size, h_size), nn.ReLU()]
        self.sig_model = nn.Sequential(*sig)
        col = [nn.Linear(15 + d_view, h_size), nn.ReLU()]
        for i in range(num_col):
            if i == num_col - 1:
                col.append(nn.Linear(h_size, 3))
            else:
                col += [nn.Linear(h_size, h_size), nn.ReLU()]
        self.col_model = nn.Sequential(*col)
        self.use_sdf = use_sdf
        self._cfg_ok = (num_sig == 2 and num_col == 2 and h_size == 64 and d_view == 24 and L * F + E == 32 and not use_sdf)
        self._flat = None

    # parameter order of the C ABI's flat block (include/hbr_hip.h)
    def _ordered(self):
        out = []
        for seq in (self.sig_model, self.col_model):
            for idx in (0, 2, 4):
                out += [seq[idx].weight, seq[idx].bias]
        return out

    def _require_cfg(self):
        if not self._cfg_ok:
            raise NotImplementedError(
                "the gfx950 MLP kernels are built for the train_hash2.py:127 instance "
                "MLP_3D(num_sig=2, num_col=2, h_size=64, d_view=24, L=16, F=2, E=0, use_sdf=False); "
                "no eager fallback is provided")

    def flat_params(self):
        """One contiguous fp32 block aliased by the 12 parameters (so kernels read the live weights every
        call and optimizers update them in place).  Returns (flat, splits)."""
        self._require_cfg()
        ps = self._ordered()
        flat = self._flat
        off, ok = 0, flat is not None and flat.device == ps[0].device
        splits = []
        for p in ps:
            n = p.numel()
            if ok and not (p.data_ptr() == flat.data_ptr() + off * 4 and p.is_contiguous()):
                ok = False
            splits.append((off, off + n, tuple(p.shape)))
            off += n
        assert off == MLP_PARAM_FLOATS
        if not ok:
            flat = torch.cat([p.detach().float().reshape(-1) for p in ps]).contiguous()
            for p, (a, b, shape) in zip(ps, splits):
                p.data = flat[a:b].view(shape)
            self._flat = flat
        return flat, tuple(splits)

    def forward(self, x, viewdirs=None, mask=None):
        """x [N,32] hash features, viewdirs [N,24] encoded directions -> [N,4] = (rgb, sigma)
        (test_hash.py:52-72).  viewdirs=None returns the density column only (:73-77)."""
        flat, splits = self.flat_params()
        if x.dim() != 2 or x.shape[-1] != 32:
            raise HbrError("MLP_3D.forward expects x of shape [N,32]")
        if viewdirs is None:
            vd = torch.zeros((x.shape[0], 24), dtype=torch.float32, device=x.device)
        else:
            vd = viewdirs
        out = ops.MlpFn.apply(x, vd, 1, flat, ops.precision_from_autocast(), splits, *self._ordered())
        if viewdirs is None:
            out = out[:, 3:4]
            return out * mask if mask is not None else out
        if mask is not None:
            out = out * mask[..., None]
        return out
