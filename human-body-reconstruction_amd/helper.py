"""Ray geometry, sampling and compositing helpers with the reference's names/signatures (helper.py).

`calc_color` runs on the wave-scan compositing kernel K5; the geometry helpers are cheap tensor plumbing
(boundary inputs) and stay in PyTorch on whatever device their inputs live on.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops


def get_od(H, W, K, c2w: torch.Tensor, find_inv: Optional[bool] = False):
    """Pixel -> ray.  Returns (rays_o [B,HW,3], unit rays_d [B,HW,3], |d| [B,HW,1]) (helper.py:176-208):
    camera dir = ((i-cx)/fx, -(j-cy)/fy, -1), world dir = R @ dir, origin = c2w[:, :3, 3]."""
    dev = c2w.device
    H, W = int(H), int(W)
    jj, ii = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    K = K.to(dev) if torch.is_tensor(K) else torch.as_tensor(K, device=dev)
    u = ((ii - K[0, 2]) / K[0, 0]).reshape(-1)
    v = ((jj - K[1, 2]) / K[1, 1]).reshape(-1)
    cam = torch.stack((u, -v, -torch.ones_like(u)), dim=-1)
    R = c2w[..., :3, :3]
    if find_inv:
        R = torch.linalg.inv(R)
    d = (R @ cam.mT).mT
    o = c2w[..., :3, 3:4].mT.expand(-1, d.shape[1], -1)
    n = torch.norm(d, dim=-1, keepdim=True)
    return o, d / n, n


def _take_cuda_philox(device, n: int):
    """(seed, offset) of the device's torch CUDA generator, advancing it by n draws - what a torch.rand(n) on that
    device would consume (Philox offsets move in units of 4).  So `torch.manual_seed(s)` replays the same depths,
    `torch.cuda.get_rng_state` / `set_rng_state` save and restore them, and nothing process-global is kept here."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    gen = torch.cuda.default_generators[idx]
    seed, off = gen.initial_seed(), gen.get_offset()
    gen.set_offset(off + (int(n) + 3) // 4 * 4)
    return seed, off


def strat_sampler(tn, tf, num_samples: int, exp: Optional[bool] = False, device=None) -> torch.Tensor:
    """t[S] = linspace(tn,tf,S) + U[0,1)^S * (tf-tn)/S  (helper.py:210-237); ONE jitter per sample index,
    shared by every ray; may exceed tf.  `exp` samples uniformly in log-depth.
    On the GPU (non-exp) this is one kernel launch: the uniforms come from a counter-based generator keyed by the
    (seed, offset) of torch's CUDA generator for that device, which is advanced as torch.rand would advance it - the
    values differ from torch.rand's, the reproducibility rules do not.  On the CPU, and for `exp`, the reference's torch
    ops are used as they are."""
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    tn, tf = torch.as_tensor(tn, dtype=torch.float32), torch.as_tensor(tf, dtype=torch.float32)
    if exp:
        lt = torch.linspace(float(torch.log(tn)), float(torch.log(tf)), num_samples, device=device)
        lt = lt + torch.rand_like(lt) * float(torch.log(tf) - torch.log(tn)) / num_samples
        return torch.exp(lt)
    if torch.device(device).type == "cuda":
        seed, off = _take_cuda_philox(device, num_samples)
        return ops.strat_sample(float(tn), float(tf), int(num_samples), device, seed=seed, offset=off)
    t = torch.linspace(float(tn), float(tf), num_samples, device=device)
    return t + torch.rand_like(t) * float(tf - tn) / num_samples


def hierarchical_sampling(rays_o: torch.Tensor, rays_d: torch.Tensor, z_vals: torch.Tensor, weights: torch.Tensor,
                          n_samples: int, tn: float, tf: float, perturb: bool = False, device: str = "cuda",
                          u: Optional[torch.Tensor] = None, samples01: Optional[torch.Tensor] = None):
    """Inverse-CDF resampling of the reference's second pass (helper.py:23-51).  Returns (points [R, S+n, 3],
    combined depths [R, S+n]).  Negative weights count as 0; the new depths index ONE shared random vector
    `samples` of length n_samples (the reference's behaviour, helper.py:43-45), then merge-sort with z_vals.
    `u` [R,S] / `samples01` [n] optionally supply the two uniform draws (parity tests); default: drawn on the device
    from torch's CUDA generator state (manual_seed replays them).  One kernel, one wave per ray
    (hbr_hierarchical_resample) - no cumsum / searchsorted / cat / sort launches."""
    w = weights.detach().reshape(weights.shape[0], -1)
    seed = off = 0
    if u is None or samples01 is None:
        seed, off = _take_cuda_philox(w.device, w.numel() + n_samples)
    combined = ops.hierarchical_resample(w, z_vals, n_samples, float(tn), float(tf), u=u, samples01=samples01, seed=seed, offset=off)
    rays = rays_o[..., None, :] + rays_d[..., None, :] * combined[..., :, None]
    return rays, combined


def calc_color(t, rgb, sigma, dir_norm, use_sdf: bool = False, var_model=None, rays=None, model=None, encoder=None,
               device: str = "cuda"):
    """Alpha compositing (helper.py:53-107, non-SDF branch): returns (Cr [R,3], wts [R,S,1], None).
    delta_last = 0, sigma clamped at -10 (zero gradient where clamped), negative sigma allowed."""
    if use_sdf:
        raise NotImplementedError("the SDF branch (helper.py:80-89) is out of scope (flag default off, train_hash2.py:33)")
    if t.dim() not in (1, 2):
        raise ValueError("t must be [S] (shared) or [R,S] (per ray)")
    Cr, wts = ops.CompositeFn.apply(t, rgb, sigma, dir_norm)
    return Cr, wts[:, :, None], None


def find_bounding_box(data_loader, near, far, K, num_samples=64, exp=False, device=None):
    """AABB of all ray points at t in {near, far+1.5} over a loader of (image, c2w, _) batches
    (helper.py:109-141).  Returns (max_bound[3], min_bound[3])."""
    if device is None:
        device = K.device
    W, H = 2 * K[0, 2], 2 * K[1, 2]
    near, far = torch.as_tensor(near, dtype=torch.float32), torch.as_tensor(far, dtype=torch.float32)
    if exp:
        t = torch.stack([near, far * torch.exp((torch.log(far) - torch.log(near)) / num_samples)]).to(device)
    else:
        t = torch.stack([near, far + 1.5]).to(device)
    mn = torch.full((3,), 1e7, device=device)
    mx = torch.full((3,), -1e7, device=device)
    with torch.no_grad():
        for batch in data_loader:
            c2w = batch[1].to(device)
            o, d, _ = get_od(H, W, K, c2w)
            pts = (o[..., None, :] + d[..., None, :] * t[None, :, None]).reshape(-1, 3)
            mn = torch.minimum(mn, pts.min(dim=0).values)
            mx = torch.maximum(mx, pts.max(dim=0).values)
    return mx, mn


def find_bounding_box2(data_loader, near, far, K=None, num_samples=64, exp=False, device=None):
    """Same over a loader of pre-materialised (rays_o, rays_d, ...) batches (helper.py:143-174)."""
    near, far = float(near), float(far)
    mn = mx = None
    with torch.no_grad():
        for batch in data_loader:
            o, d = batch[0], batch[1]
            t = torch.tensor([near, far + 1.5], device=o.device)
            pts = (o[..., None, :] + d[..., None, :] * t[None, :, None]).reshape(-1, 3)
            lo, hi = pts.min(dim=0).values, pts.max(dim=0).values
            mn = lo if mn is None else torch.minimum(mn, lo)
            mx = hi if mx is None else torch.maximum(mx, hi)
    return mx, mn


def calc_psnr(pred, target):
    """10*log10(1/mse) (helper.py:301-304)."""
    return 10 * torch.log10(1.0 / torch.mean((pred - target) ** 2))


def cumprod_exclusive(tensor: torch.Tensor) -> torch.Tensor:
    """Exclusive cumulative product along the last dim (helper.py:268-291)."""
    cp = torch.cumprod(tensor, -1)
    return torch.cat([torch.ones_like(cp[..., :1]), cp[..., :-1]], dim=-1)
