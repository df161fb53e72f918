"""Dense-grid field query for meshing (reference nerf2mesh.py:26-88).

The reference samples a res^3 grid of the bounding box, feeds 400 000-point batches through encoder -> dir encoder ->
MLP with the fixed view direction (0,0,1) (nerf2mesh.py:69-70) and stores `[res,res,res,4]` = (rgb, density)
(:85-87) for `torchmcubes.marching_cubes(density, 30.0)` (third-party, out of scope).  Here the query re-uses K1 and
K3 unchanged (planar features, one encoded direction row shared by every point of a batch).

Quirks reproduced: coordinates are built in float64, laid out by `np.meshgrid`'s default 'xy' indexing and cast to
float16 (nerf2mesh.py:30-40), so the grid points are fp16-rounded positions in (y, x, z)-major order; the view
direction is float16 too, so its sin/cos encoding is rounded to float16.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import ops
from ._lib import PLANAR
from .hash_encoding import HashEncoder
from .test_hash import MLP_3D


def _axis_linspace64(lo: float, hi: float, res: int, idx: torch.Tensor) -> torch.Tensor:
    """np.linspace(lo, hi, res)[idx] in float64, as numpy evaluates it: step = (hi - lo) / (res - 1);
    y = idx * step + lo (two roundings); the last sample is `hi` itself."""
    if res == 1:
        return torch.full(idx.shape, float(lo), dtype=torch.float64, device=idx.device)
    step = (float(hi) - float(lo)) / (res - 1)
    y = idx.to(torch.float64) * step + float(lo)
    return torch.where(idx == res - 1, torch.full_like(y, float(hi)), y)


def grid_coordinates(min_bound, max_bound, res: int, device, start: int = 0, stop: Optional[int] = None,
                     index: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Rows [start, stop) - or the rows `index` (int64 flat indices) - of the reference's [res^3, 3] lattice as fp32 values of fp16-rounded positions
    (nerf2mesh.py:30-40): float64 `np.linspace` per axis, `np.meshgrid(x, y, z)` with its default 'xy' indexing - so
    the flat index is (iy*res + ix)*res + iz - stacked, cast to float16.  Built on `device` from the flat index, so
    a 512^3 lattice (1.3e8 points, 3.2 GB as float64 on the host in the reference) never exists as a whole."""
    mn = np.asarray(torch.as_tensor(min_bound).detach().cpu(), dtype=np.float64).reshape(-1)
    mx = np.asarray(torch.as_tensor(max_bound).detach().cpu(), dtype=np.float64).reshape(-1)
    stop = res ** 3 if stop is None else min(stop, res ** 3)
    i = torch.arange(start, stop, dtype=torch.int64, device=device) if index is None else index.to(device=device, dtype=torch.int64)
    iz, ix, iy = i % res, (i // res) % res, i // (res * res)
    grid = torch.stack([_axis_linspace64(mn[0], mx[0], res, ix), _axis_linspace64(mn[1], mx[1], res, iy),
                        _axis_linspace64(mn[2], mx[2], res, iz)], dim=1)
    return grid.to(torch.float16).float()


@torch.no_grad()
def query_density_grid(encoder: HashEncoder, mlp: MLP_3D, min_bound, max_bound, res: int = 256, batch: int = 400000,
                       view_dir=(0.0, 0.0, 1.0), precision=None, num_freq: int = 4, out_device=None) -> torch.Tensor:
    """Returns density_grid_w_rgb [res,res,res,4] fp32 = (r,g,b,density) (nerf2mesh.py:85-87)."""
    mlp = mlp.module if hasattr(mlp, "module") else mlp
    tables = encoder.stacked_tables()
    dev = tables.device
    geom = encoder.geometry()
    flat, _ = mlp.flat_params()
    prec = ops.precision_from_autocast() if precision is None else precision
    # the reference feeds a float16 view_dir through PositionalEncoder (nerf2mesh.py:69-70,81): sin/cos come out as
    # float16 and are promoted back to fp32 by the concat in MLP_3D.forward
    pe = ops.dir_encode(torch.tensor([view_dir], dtype=torch.float32, device=dev), num_freq).half().float().contiguous()
    n = res ** 3
    out = torch.empty((n, 4), dtype=torch.float32, device=out_device or dev)
    # The reference walks the lattice in 400 000-point batches because its GPU had to hold the activations of each; the
    # query is pointwise, so the batch size changes no value, and 336 batches of a 512^3 lattice spend their time in launch
    # gaps and torch's index arithmetic (0.20 s where the two kernels need 0.03).  `batch` is therefore a LOWER bound:
    # chunks of 2^24 points (2.1 GB of planar fp32 features at most) unless the caller asks for larger ones.
    batch = max(int(batch), 1 << 24)
    for i in range(0, n, batch):
        x = grid_coordinates(min_bound, max_bound, res, dev, i, i + batch)
        feat = ops.hash_encode_fwd(geom, tables, x=x, layout=PLANAR)
        o = ops.mlp_fwd(feat, PLANAR, pe, x.shape[0], flat, prec)   # group = batch size: every point uses pe row 0
        out[i:i + batch] = o.to(out.device)
    return out.reshape(res, res, res, 4)
