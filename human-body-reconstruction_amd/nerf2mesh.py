"""`python -m hbr_amd.nerf2mesh` - the dense-grid half of the reference's nerf2mesh.py (lines 26-88) on the MI355X path.

Reads the bounds file (`stack([min_bound, max_bound])`, train_hash2.py:115) and the two reference-format checkpoints,
queries the field on a resolution^3 lattice of the bounding box (fp16-rounded coordinates, view direction (0,0,1),
400 000-point batches - nerf2mesh.py:30-40,69-84) and writes `density_grid_w_rgb.npy` = float32 [res,res,res,4] =
(r, g, b, density), the file the reference caches (:85-87) and feeds to `torchmcubes.marching_cubes(density, 30.0)`.
Marching cubes itself (torchmcubes / open3d: third-party, not vendored by the reference) is not part of this package;
any consumer of that .npy - the reference's own script included - picks up from here.

Flags are the reference's (nerf2mesh.py:15-24) plus --resolution (256 there, hard-coded :27; BASELINE config 5 asks
for 512), --out, --precision.
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import torch

_REFERENCE_FLAGS = (
    ("--use_sdf", None, None, "rejected: the SDF branch is outside the accelerated path"),
    ("--hierarchical", None, None, "accepted for flag parity; a grid query has no ray sampling"),
    ("--max_res", float, 2048, "finest grid resolution N_max the checkpoint was trained with"),
    ("--hash_size", float, 16, "log2 of the rows per level of the checkpoint"),
    ("--model_name", str, "default", "unused (flag parity)"),
    ("--bound_pth", str, "bounds.npy", "the [2,3] min/max bounds file written by the trainer"),
    ("--ckpt_name", str, "N_2048_T_16", "prefix of {..}_Nerf_hash.pth / {..}_encoder_hash.pth"),
    ("--near", float, 2.0, "unused by the query (flag parity)"),
    ("--far", float, 6.0, "unused by the query (flag parity)"),
)


def build_parser():
    p = argparse.ArgumentParser(description="dense density/colour grid of a trained hash-NeRF (MI355X)")
    for flag, typ, default, text in _REFERENCE_FLAGS:
        if typ is None:
            p.add_argument(flag, action="store_true", help=text)
        else:
            p.add_argument(flag, type=typ, default=default, help=text)
    p.add_argument("--resolution", type=int, default=256, help="lattice points per axis (reference: 256)")
    p.add_argument("--out", default="density_grid_w_rgb.npy", help="output .npy (reference: density_grid_w_rgb.npy)")
    p.add_argument("--precision", default="fp32", choices=["fp32", "bf16"], help="MLP arithmetic (the reference's script runs fp32)")
    p.add_argument("--batch", type=int, default=400000, help="points per launch (reference: 400000)")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.use_sdf:
        raise NotImplementedError("--use_sdf: the SDF branch is out of scope")
    from . import _lib, checkpoint
    from .grid_query import query_density_grid
    from .trainer import build_default_model

    if not torch.cuda.is_available():
        raise _lib.HbrError("nerf2mesh needs an MI355X; there is no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device())
    mn, mx, mu, sigma = checkpoint.load_bounds(args.bound_pth)
    print("BOUNDING BOX:", mx.tolist(), mn.tolist())
    enc, _, mlp = build_default_model(mu, sigma, dev, L=16, F=2, T=int(2 ** args.hash_size), N_max=args.max_res)
    nerf = torch.nn.DataParallel(mlp, device_ids=[dev.index])
    d = os.path.dirname(args.ckpt_name) or "."
    checkpoint.load_checkpoint(os.path.basename(args.ckpt_name), nerf, enc, directory=d)
    prec = _lib.BF16 if args.precision == "bf16" else _lib.F32
    res = int(args.resolution)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    grid = query_density_grid(enc, mlp, mn, mx, res=res, batch=args.batch, precision=prec)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    np.save(args.out, grid.cpu().numpy())
    print(f"{res}^3 = {res ** 3} points in {dt:.3f} s ({res ** 3 / dt:.3e} points/s) -> {args.out} {tuple(grid.shape)}")
    return {"resolution": res, "seconds": dt, "points_per_s": res ** 3 / dt, "out": args.out,
            "occupied_fraction": float((grid[..., 3] > 30.0).float().mean())}


if __name__ == "__main__":
    print(main())
