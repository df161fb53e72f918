"""Dense-grid field query for meshing (reference nerf2mesh.py:26-88).

The reference samples a res^3 grid of the bounding box, feeds 400 000-point batches through encoder -> dir encoder ->
MLP with the fixed view direction (0,0,1) (nerf2mesh.py:69-70) and stores `[res,res,res,4]` = (rgb, density)
(:85-87) for `torchmcubes.marching_cubes(density, 30.0)` (third-party, out of scope).  Here the query re-uses K1 and
K3 unchanged (planar features, one encoded direction row shared by every point of a batch).

Quirk reproduced: coordinates are built in float16 (nerf2mesh.py:40) and cast to fp32 before encoding, so the grid
points are the fp16-rounded positions, not the exact lattice.
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import PLANAR
from .hash_encoding import HashEncoder
from .test_hash import MLP_3D


def grid_coordinates(min_bound: torch.Tensor, max_bound: torch.Tensor, res: int, device) -> torch.Tensor:
    """[res^3, 3] fp32 values of fp16-rounded lattice points, x slowest / z fastest ('ij' meshgrid, nerf2mesh.py:36-41)."""
    axes = [torch.linspace(float(min_bound[i]), float(max_bound[i]), res, device=device).to(torch.float16) for i in range(3)]
    gx, gy, gz = torch.meshgrid(*axes, indexing="ij")
    return torch.stack([gx, gy, gz], dim=-1).reshape(-1, 3).float()


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
    pts = grid_coordinates(min_bound, max_bound, res, dev)
    pe = ops.dir_encode(torch.tensor([view_dir], dtype=torch.float32, device=dev), num_freq)  # [1,24], shared by all points
    out = torch.empty((pts.shape[0], 4), dtype=torch.float32, device=out_device or dev)
    for i in range(0, pts.shape[0], batch):
        x = pts[i:i + batch].contiguous()
        feat = ops.hash_encode_fwd(geom, tables, x=x, layout=PLANAR)
        o = ops.mlp_fwd(feat, PLANAR, pe, x.shape[0], flat, prec)   # group = batch size: every point uses pe row 0
        out[i:i + batch] = o.to(out.device)
    return out.reshape(res, res, res, 4)
