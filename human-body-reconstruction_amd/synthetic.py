"""Synthetic lego-shaped workload for running the trainer and the benchmark without a dataset.

No dataset ships with the reference (`data/lego` is user-supplied, train_hash2.py:51), so the CLI's `--synthetic`
switch and `bench.py` draw their rays here: cameras on the upper hemisphere of radius 4.03 looking at the object,
near 2 / far 6, and - for `scene_rays` - ground truth composited from an analytic density/colour field, so that loss
and PSNR of a run mean something.  Product code, self-contained (numpy + torch only).
"""
from __future__ import annotations

import numpy as np
import torch

LEGO_RADIUS = 4.03


def _f32(a) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def hemisphere_rays(R: int, seed: int = 0, radius: float = LEGO_RADIUS):
    """R rays from cameras on the upper hemisphere aimed at points inside |x| < 0.6 (numpy PCG64 stream, so the rays
    do not depend on torch's RNG).  Returns o[R,3], unit d[R,3], dir_norm[R,1] in [1,1.2), gt[R,3] (a smooth colour of
    the aim point) as fp32 CPU tensors."""
    rng = np.random.default_rng(seed)
    az = rng.uniform(0, 2 * np.pi, R)
    pol = rng.uniform(0.05, 0.5 * np.pi, R)
    cam = radius * np.stack([np.cos(az) * np.sin(pol), np.sin(az) * np.sin(pol), np.cos(pol)], -1)
    aim = rng.uniform(-0.6, 0.6, (R, 3))
    v = aim - cam
    v = v / np.linalg.norm(v, axis=-1, keepdims=True)
    colour = 0.5 + 0.5 * np.sin(3.0 * aim + np.array([0.0, 1.0, 2.0]))
    dn = rng.uniform(1.0, 1.2, (R, 1))
    return _f32(cam), _f32(v), _f32(dn), _f32(colour)


def ray_bbox(o: torch.Tensor, d: torch.Tensor, near: float = 2.0, far: float = 6.0):
    """AABB of the ray points at t in {near, far+1.5} (helper.py:109-141) -> (min[3], max[3], diagonal 0-d): the
    encoder's mu and sigma (train_hash2.py:117-119)."""
    ends = torch.tensor([near, far + 1.5], dtype=torch.float32, device=o.device)
    p = (o[:, None, :] + d[:, None, :] * ends[None, :, None]).reshape(-1, 3)
    lo, hi = p.min(dim=0).values, p.max(dim=0).values
    return lo, hi, ((hi - lo) ** 2).sum().sqrt()


def solid_field(x: torch.Tensor):
    """Analytic 'lego-like' solid inside |x| < 1.2: three soft boxes and a sphere.  Returns (sigma >= 0 [...],
    rgb [...,3])."""
    def vec(v):
        return torch.tensor(v, dtype=x.dtype, device=x.device)

    def box(centre, half):
        return ((x - vec(centre)).abs() - vec(half)).max(dim=-1).values  # negative inside

    dist = torch.minimum(box((0.0, 0.0, -0.3), (0.9, 0.6, 0.2)), box((-0.3, 0.0, 0.1), (0.4, 0.35, 0.25)))
    dist = torch.minimum(dist, box((0.45, 0.1, 0.15), (0.2, 0.45, 0.3)))
    dist = torch.minimum(dist, (x - vec((0.0, -0.2, 0.55))).norm(dim=-1) - 0.3)
    return 25.0 * torch.sigmoid(-dist / 0.03), 0.5 + 0.5 * torch.sin(4.0 * x + vec((0.0, 2.0, 4.0)))


def _composite_uniform(t: torch.Tensor, rgb: torch.Tensor, sigma: torch.Tensor) -> torch.Tensor:
    """The reference's compositing rule (helper.py:65-105: delta_last = 0, alpha = 1-exp(-sigma*delta), exclusive
    transmittance) in plain torch; only used to paint the synthetic ground truth."""
    delta = torch.zeros_like(t)
    delta[:-1] = t[1:] - t[:-1]
    p = sigma * delta[None, :]
    trans = torch.exp(-torch.cumsum(p, dim=-1))
    trans = torch.cat([torch.ones_like(trans[:, :1]), trans[:, :-1]], dim=-1)
    w = trans * (1 - torch.exp(-p))
    return (w[:, :, None] * rgb).sum(dim=-2)


def scene_rays(R: int, seed: int = 0, radius: float = LEGO_RADIUS, near: float = 2.0, far: float = 6.0, quad: int = 384,
               device="cpu"):
    """hemisphere_rays() whose ground truth is `solid_field` composited on a `quad`-sample uniform quadrature, i.e. a
    multi-view-consistent radiance field.  dir_norm = 1.  Returns o, d, dir_norm[R,1], gt[R,3] on `device`."""
    o, d, _, _ = hemisphere_rays(R, seed=seed, radius=radius)
    o, d = o.to(device), d.to(device)
    t = torch.linspace(near, far, quad, device=device)
    gt = []
    for i in range(0, R, 4096):
        pts = o[i:i + 4096, None, :] + d[i:i + 4096, None, :] * t[None, :, None]
        sg, rgb = solid_field(pts)
        gt.append(_composite_uniform(t, rgb, sg))
    return o, d, torch.ones((R, 1), device=device), torch.cat(gt).clamp(0, 1)
