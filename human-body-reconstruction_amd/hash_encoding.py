"""HashEncoder: multiresolution hash-grid encoder backed by the gfx950 kernels K1/K2.

Same constructor / forward signature, attribute names and state-dict keys as the reference's
hash_encoding.py:5-170 (`Embedding_list.{i}.weight`, [T,F] fp32 each).  The 16 per-level weights are
views into ONE stacked [L,T,F] buffer so a single kernel launch covers all levels.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from ._lib import HbrError


class _Level(nn.Module):
    """Stand-in for nn.Embedding(T, F): only `.weight` is used by callers
    (`encoder.Embedding_list.parameters()`, train_hash2.py:141)."""

    def __init__(self, weight: torch.Tensor):
        super().__init__()
        self.weight = nn.Parameter(weight)

    def forward(self, idx):  # gather, for API completeness
        return self.weight[idx]


class HashEncoder(nn.Module):
    def __init__(self, N_max, N_min, L, E=0, T=2 ** 14, F=2, dim=2, mu=None, sigma=None, device=None):
        super().__init__()
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        if dim != 3:
            raise NotImplementedError("hbr_amd.HashEncoder implements the 3-D grid used by vol_render (dim=3); "
                                      "the reference's dim=2 image demo is out of scope")
        if F != 2:
            raise NotImplementedError("kernels are built for F=2 features per level (train_hash2.py:107)")
        self.device = device
        # level growth factor, evaluated with the reference's ops/dtypes (hash_encoding.py:11-13):
        # python float -> fp32 tensor, python int -> int64 tensor, log(int64) -> fp32
        self.N_max = torch.tensor(N_max)
        self.N_min = torch.tensor(N_min)
        self.b = torch.exp((torch.log(self.N_max) - torch.log(self.N_min)) / (L - 1))
        self.L, self.F, self.T, self.E, self.dim = L, F, int(T), E, dim
        self.sigma = 1 if sigma is None else (sigma.to(device) if torch.is_tensor(sigma) else sigma)
        self.mu = 0 if mu is None else mu
        # hash_encoding.py:30-33: U(-1e-4, 1e-4) per level
        stacked = torch.empty((L, self.T, F), dtype=torch.float32, device=device)
        nn.init.uniform_(stacked, a=-1e-4, b=1e-4)
        self._stacked = stacked
        self.Embedding_list = nn.ModuleList([_Level(stacked[i]) for i in range(L)])
        self._geom = None
        self._geom_key = None

    # ---- geometry -----------------------------------------------------------------------------
    def level_scales(self) -> torch.Tensor:
        """N_l = N_min * b**l as fp32 (hash_encoding.py:153); not integers, e.g. N_15 = 2047.99.."""
        return torch.stack([(self.N_min * self.b ** i).to(torch.float32) for i in range(self.L)])

    def geometry(self) -> ops.HashGeom:
        mu, sigma = self.mu, self.sigma
        key = (id(mu), id(sigma), None if not torch.is_tensor(mu) else mu._version)
        if self._geom is None or key != self._geom_key:
            if torch.is_tensor(mu):
                m = [float(v) for v in mu.detach().float().cpu().reshape(-1)]
                m = m * 3 if len(m) == 1 else m
            else:
                m = [float(mu)] * 3
            s = float(sigma)  # one sync at first use; sigma is a constant of the scene (train_hash2.py:119)
            self._geom = ops.HashGeom(tuple(float(v) for v in self.level_scales()), (m[0], m[1], m[2]), s, self.T, self.F)
            self._geom_key = key
        return self._geom

    # ---- parameter storage --------------------------------------------------------------------
    def stacked_tables(self) -> torch.Tensor:
        """[L,T,F] buffer aliased by Embedding_list[i].weight.  Re-established if the module was moved
        (`.to(device)`, load_state_dict keep aliasing; `.to` onto another device does not)."""
        ws = [lvl.weight for lvl in self.Embedding_list]
        st = self._stacked
        nbytes = self.T * self.F * 4
        ok = st.device == ws[0].device and all(w.data_ptr() == st.data_ptr() + i * nbytes and w.is_contiguous()
                                               for i, w in enumerate(ws))
        if not ok:
            st = torch.stack([w.detach().float() for w in ws]).contiguous()
            for i, w in enumerate(ws):
                w.data = st[i]  # keeps the Parameter object (optimizers hold references to it)
            self._stacked = st
        return st

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._stacked = fn(self._stacked)
        if torch.is_tensor(self.sigma):
            self.sigma = fn(self.sigma)
        if torch.is_tensor(self.mu):
            self.mu = fn(self.mu)
        return out

    # ---- forward ------------------------------------------------------------------------------
    def forward(self, x, aux=None):
        assert x.shape[-1] == self.dim  # hash_encoding.py:147
        if x.dim() != 2:
            raise HbrError("HashEncoder.forward expects x of shape [N,3] (as vol_render passes it)")
        st = self.stacked_tables()
        return ops.HashEncodeFn.apply(x, st, self.geometry(), self.E, *[lvl.weight for lvl in self.Embedding_list])
