"""CPU oracle for the hash-NeRF render/train hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU / numpy *restatement* of the reference algorithm
(RishabhSri14/Human-Body-Reconstruction).  It is the checker the HIP path is compared
against; it is never the thing shipped or measured as the product.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.

Parity status: PINNED.  Every function below is checked by `tests/test_oracle_golden.py`
against golden vectors in `tests/golden/*.npz` that were produced by importing the
reference's own modules (`oracle/make_golden.py`, run in the build container where
`/root/reference` is mounted).

Each function cites the reference file:line it follows.  Nothing here is copied from the
reference: the reference loops over levels building [N,8,3] int64 index tensors with
`torch.where`; this restatement evaluates the eight corners explicitly with uint32-equivalent
arithmetic so that it doubles as the specification of the HIP kernels' integer path.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

# The three spatial-hash multipliers (hash_encoding.py:24).  The reference stores them as
# int32 (so 2654435761 wraps to -1640531535) and multiplies in int64; only the low bits
# survive the final modulo when T is a power of two, so uint32 arithmetic is equivalent.
PRIME_Y_I64 = -1640531535
PRIME_Z_I64 = 805459861
PRIME_Y_U32 = 2654435761
PRIME_Z_U32 = 805459861


# --------------------------------------------------------------------------------------
# level scales                                               hash_encoding.py:11-13, :153
# --------------------------------------------------------------------------------------
def level_scales(N_min, N_max, L: int) -> torch.Tensor:
    """fp32 per-level grid scales N_l = N_min * b**l, evaluated with the same torch ops and
    dtypes as the reference (`torch.tensor(N_max)` keeps python-float -> fp32, python-int ->
    int64; log of an int64 tensor promotes to fp32).  The values are NOT integers
    (N_15 = 2047.99.. for 16 -> 2048.0) and kernels must take them as data."""
    n_max = torch.tensor(N_max)
    n_min = torch.tensor(N_min)
    b = torch.exp((torch.log(n_max) - torch.log(n_min)) / (L - 1))
    out = [(n_min * b ** i).to(torch.float32) for i in range(L)]
    return torch.stack(out)


# --------------------------------------------------------------------------------------
# integer path: cell + spatial hash                          hash_encoding.py:41-55,:127-136,:157
# --------------------------------------------------------------------------------------
def spatial_hash_np(ix: np.ndarray, iy: np.ndarray, iz: np.ndarray, T: int) -> np.ndarray:
    """hash_func (hash_encoding.py:49-53): (ix*1) ^ (iy*p1) ^ (iz*p2) in int64, then Python
    floor-mod T.  Works for any T >= 1 (not only powers of two) and negative coordinates."""
    ix = ix.astype(np.int64)
    iy = iy.astype(np.int64)
    iz = iz.astype(np.int64)
    with np.errstate(over="ignore"):
        v = ix ^ (iy * np.int64(PRIME_Y_I64)) ^ (iz * np.int64(PRIME_Z_I64))
    return np.mod(v, np.int64(T))  # np.mod is floor-mod: result in [0, T)


def spatial_hash_u32_np(ix, iy, iz, log2T: int) -> np.ndarray:
    """The uint32 form the HIP kernel uses for power-of-two T (SURVEY 7 hard part 2.iv)."""
    ix = ix.astype(np.int64).astype(np.uint32)
    iy = iy.astype(np.int64).astype(np.uint32)
    iz = iz.astype(np.int64).astype(np.uint32)
    with np.errstate(over="ignore"):
        v = ix ^ (iy * np.uint32(PRIME_Y_U32)) ^ (iz * np.uint32(PRIME_Z_U32))
    return (v & np.uint32((1 << log2T) - 1)).astype(np.int64)


def scaled_coords(x: torch.Tensor, mu, sigma, scale: torch.Tensor) -> torch.Tensor:
    """un_x = ((x - mu) / sigma) * N_l, three separately rounded fp32 ops (hash_encoding.py:154)."""
    return ((x - mu) / sigma) * scale


def corner_indices(x: torch.Tensor, mu, sigma, scales: torch.Tensor, T: int):
    """For every level: truncated cell x0 (hash_encoding.py:157 `.long()` truncates toward
    zero), fractional part, and the 8 hashed row ids.  Corner n takes x0+1 on axis d iff bit d
    of n is set (bin_mask, hash_encoding.py:34-37,:135).
    Returns (cells int64 [L,N,3], frac f32 [L,N,3], rows int64 [L,N,8])."""
    L = scales.shape[0]
    cells, fracs, rows = [], [], []
    for l in range(L):
        u = scaled_coords(x, mu, sigma, scales[l])
        c0 = u.long()
        fr = u - c0
        c = c0.numpy()
        per_corner = []
        for n in range(8):
            cx = c[:, 0] + ((n >> 0) & 1)
            cy = c[:, 1] + ((n >> 1) & 1)
            cz = c[:, 2] + ((n >> 2) & 1)
            per_corner.append(spatial_hash_np(cx, cy, cz, T))
        cells.append(c0)
        fracs.append(fr)
        rows.append(torch.from_numpy(np.stack(per_corner, axis=1)))
    return torch.stack(cells), torch.stack(fracs), torch.stack(rows)


# --------------------------------------------------------------------------------------
# hash encoder forward (autograd gives the scatter-add)       hash_encoding.py:146-170
# --------------------------------------------------------------------------------------
def hash_encode(x: torch.Tensor, tables: Sequence[torch.Tensor], scales: torch.Tensor,
                mu, sigma) -> torch.Tensor:
    """y[N, L*F].  tables[l] is the [T,F] fp32 weight of Embedding_list[l].
    Trilinear weight of corner n = prod_d (bit_d(n) ? frac_d : 1-frac_d), multiplied in the
    axis order x,y,z (hash_encoding.py:142-143); frac is detached (:160) so no gradient flows
    to x."""
    assert x.shape[-1] == 3
    L = len(tables)
    T, F = tables[0].shape
    with torch.no_grad():
        _, fracs, rows = corner_indices(x.detach(), mu, sigma, scales, T)
    outs = []
    for l in range(L):
        fr = fracs[l]
        one_m = 1 - fr
        acc = None
        for n in range(8):
            wx = fr[:, 0] if (n & 1) else one_m[:, 0]
            wy = fr[:, 1] if (n & 2) else one_m[:, 1]
            wz = fr[:, 2] if (n & 4) else one_m[:, 2]
            w = (wx * wy) * wz
            term = tables[l][rows[l][:, n]] * w[:, None]
            acc = term if acc is None else acc + term
        outs.append(acc)
    return torch.cat(outs, dim=-1)


def hash_encode_backward(x, dy, scales, mu, sigma, T: int, F: int = 2) -> torch.Tensor:
    """Explicit scatter-add (what autograd's embedding_dense_backward does):
    dTable[l, row, :] += w * dy[:, l*F:(l+1)*F].  Returns [L,T,F] fp32 (accumulated in fp64
    then rounded, so it is an order-independent reference for the atomic kernel)."""
    L = scales.shape[0]
    _, fracs, rows = corner_indices(x, mu, sigma, scales, T)
    out = np.zeros((L, T, F), dtype=np.float64)
    dy64 = dy.double().numpy()
    for l in range(L):
        fr = fracs[l]
        one_m = 1 - fr
        for n in range(8):
            wx = fr[:, 0] if (n & 1) else one_m[:, 0]
            wy = fr[:, 1] if (n & 2) else one_m[:, 1]
            wz = fr[:, 2] if (n & 4) else one_m[:, 2]
            w = ((wx * wy) * wz).double().numpy()
            np.add.at(out[l], rows[l][:, n].numpy(), w[:, None] * dy64[:, l * F:(l + 1) * F])
    return torch.from_numpy(out).float()


# --------------------------------------------------------------------------------------
# view-direction encoding                                     encoder.py:16-17,:25-32
# --------------------------------------------------------------------------------------
def dir_encode(d: torch.Tensor, num_freq: int) -> torch.Tensor:
    """Per coordinate c: [sin(2*c*k)]_{k<nf} then [cos(2*c*k)]_{k<nf}; coordinates
    concatenated.  Frequencies are 2*k (k=0 gives constants 0 and 1), not 2**k."""
    k = torch.arange(num_freq, dtype=torch.int8)
    ang = 2 * d.unsqueeze(-1) * k
    return torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1).flatten(-2)


# --------------------------------------------------------------------------------------
# field MLP                                                   test_hash.py:21-72
# --------------------------------------------------------------------------------------
MLP_KEYS = ("sig_model.0", "sig_model.2", "sig_model.4", "col_model.0", "col_model.2", "col_model.4")


def mlp_forward(feat: torch.Tensor, dirs_enc: torch.Tensor, params: dict) -> torch.Tensor:
    """The (num_sig=2, num_col=2, h=64) instance built at train_hash2.py:127.
    params: '<seq>.<idx>.weight' [out,in] / '.bias' [out] (nn.Linear convention).
    Returns [N,4] = (r,g,b,sigma): sigma = LeakyReLU_0.01(out[:,0]) (test_hash.py:54,62),
    rgb = ELU(col_model(cat(out[:,1:], dirs))) (:64-67), cat order rgb then density (:69)."""
    lin = torch.nn.functional.linear
    h = torch.relu(lin(feat, params["sig_model.0.weight"], params["sig_model.0.bias"]))
    h = torch.relu(lin(h, params["sig_model.2.weight"], params["sig_model.2.bias"]))
    s = lin(h, params["sig_model.4.weight"], params["sig_model.4.bias"])
    density = torch.nn.functional.leaky_relu(s[:, 0:1], 0.01)
    c = torch.cat([s[:, 1:], dirs_enc], dim=-1)
    c = torch.relu(lin(c, params["col_model.0.weight"], params["col_model.0.bias"]))
    c = torch.relu(lin(c, params["col_model.2.weight"], params["col_model.2.bias"]))
    rgb = torch.nn.functional.elu(lin(c, params["col_model.4.weight"], params["col_model.4.bias"]))
    return torch.cat([rgb, density], dim=-1)


def mlp_init(seed: int, in_feat: int = 32, d_view: int = 24, h: int = 64) -> dict:
    """nn.Linear default init (U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias) drawn
    from a numpy PCG64 stream so fixtures do not depend on torch's RNG."""
    rng = np.random.default_rng(seed)
    shapes = {"sig_model.0": (h, in_feat), "sig_model.2": (h, h), "sig_model.4": (16, h),
              "col_model.0": (h, 15 + d_view), "col_model.2": (h, h), "col_model.4": (3, h)}
    p = {}
    for k, (o, i) in shapes.items():
        bound = 1.0 / math.sqrt(i)
        p[k + ".weight"] = torch.from_numpy(rng.uniform(-bound, bound, (o, i)).astype(np.float32))
        p[k + ".bias"] = torch.from_numpy(rng.uniform(-bound, bound, (o,)).astype(np.float32))
    return p


# --------------------------------------------------------------------------------------
# alpha compositing                                           helper.py:53-107 (non-SDF branch)
# --------------------------------------------------------------------------------------
def composite(t: torch.Tensor, rgb: torch.Tensor, sigma: torch.Tensor, dir_norm) -> Tuple[torch.Tensor, torch.Tensor]:
    """t [S] shared by all rays; rgb [R,S,3]; sigma [R,S]; dir_norm [R,1] or scalar.
    delta_i = t_{i+1}-t_i, delta_{S-1} = 0 (helper.py:65-67), times dir_norm (:71);
    sigma clamped below at -10 with zero gradient where clamped (:76, in-place masked store);
    alpha = 1-exp(-sigma*delta) (:91); T = exclusive exp(-cumsum) (:93-95); w = T*alpha (:102);
    Cr = sum_s w*rgb (:105).  sigma may be negative: alpha<0 and T>1 are reproduced."""
    delta = torch.zeros_like(t)
    delta[:-1] = t[1:] - t[:-1]
    delta = delta[None, :] * dir_norm
    keep = sigma >= -10
    sig = torch.where(keep, sigma, torch.full_like(sigma, -10.0))
    p = sig * delta
    alpha = 1 - torch.exp(-p)
    Tr = torch.exp(-torch.cumsum(p, dim=-1))
    Tr = torch.cat([torch.ones_like(Tr[:, :1]), Tr[:, :-1]], dim=-1)
    w = Tr * alpha
    Cr = (w[:, :, None] * rgb).sum(dim=-2)
    return Cr, w[:, :, None]


# --------------------------------------------------------------------------------------
# ray geometry                                                helper.py:176-208, :210-237
# --------------------------------------------------------------------------------------
def get_od(H: int, W: int, K: torch.Tensor, c2w: torch.Tensor):
    """Pixel (i,j) -> camera dir ((i-cx)/fx, -(j-cy)/fy, -1), rotated by c2w[:, :3,:3];
    origin = c2w[:, :3, 3].  Returns (o[B,HW,3], unit d[B,HW,3], |d|[B,HW,1])."""
    jj, ii = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    u = ((ii - K[0, 2]) / K[0, 0]).reshape(-1)
    v = ((jj - K[1, 2]) / K[1, 1]).reshape(-1)
    cam = torch.stack((u, -v, -torch.ones_like(u)), dim=-1)
    d = (c2w[..., :3, :3] @ cam.mT).mT
    o = c2w[..., :3, 3:4].mT.expand(-1, d.shape[1], -1)
    n = torch.norm(d, dim=-1, keepdim=True)
    return o, d / n, n


def strat_jitter_to_t(tn: float, tf: float, S: int, u01: torch.Tensor) -> torch.Tensor:
    """strat_sampler (helper.py:234-235) with the uniform draw made explicit:
    t = linspace(tn,tf,S) + u*(tf-tn)/S.  One jitter per sample index, shared by all rays."""
    return torch.linspace(tn, tf, S) + u01 * (tf - tn) / S


def sample_points(o: torch.Tensor, d: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """vol_renderer.py:165: p = o + d*t -> [R,S,3]."""
    return o[..., None, :] + d[..., None, :] * t[None, :, None]


# --------------------------------------------------------------------------------------
# one render pass and one training step          vol_renderer.py:141-245, train_hash2.py:218-239
# --------------------------------------------------------------------------------------
def render(o, d, t, dir_norm, tables, scales, mu, sigma, mlp_params, num_freq: int = 4):
    """vol_render with hierarchical=False and an all-true occupancy mask:
    returns (Cr [R,3], sigma [R,S], rgb [R,S,3])."""
    R, S = o.shape[0], t.shape[0]
    pts = sample_points(o, d, t).reshape(-1, 3)
    feat = hash_encode(pts, tables, scales, mu, sigma)
    dirs = dir_encode(d[:, None, :].expand(R, S, 3).reshape(-1, 3), num_freq)
    out = mlp_forward(feat, dirs, mlp_params)
    sig = out[:, 3].reshape(R, S)
    rgb = out[:, 0:3].reshape(R, S, 3)
    Cr, _ = composite(t, rgb, sig, dir_norm)
    return Cr, sig, rgb


def occupancy_mask(pts: torch.Tensor, grid: torch.Tensor, mu, sigma_val) -> torch.Tensor:
    """Volume_Renderer.get_mask (vol_renderer.py:133-140): cell = trunc(((p - mu) / sigma_val) * grid_size) per axis
    (three separately rounded fp32 ops, `.long()` truncates toward zero), then bool_grid[cx, cy, cz]."""
    G = grid.shape[0]
    c = (((pts - mu) / sigma_val) * G).long()
    return grid[c[..., 0], c[..., 1], c[..., 2]]


def update_grid(pts: torch.Tensor, alpha: torch.Tensor, grid: torch.Tensor, tmp: torch.Tensor, mu, sigma_val) -> None:
    """Volume_Renderer.update_grid (vol_renderer.py:116-131), in place on `grid` (bool [G,G,G]) and `tmp` (int8 [G,G,G]),
    with the effect of its tensor ops spelled out: alpha <= 0 counts as 0 (:121); `tmp[cell] += ceil(alpha).int()` is an
    index_put WITHOUT accumulation through an int8 array, so of the points sharing a cell the LAST one (in point order)
    decides, every one of them having read the cell's OLD value, and the int32 sum wraps into int8 (:123); cells whose
    count is > 0 become True, or the whole grid if there is none (:125-128); positive counts are reset to 0, a count
    that wrapped negative is not (:131)."""
    G = grid.shape[0]
    c = (((pts - mu) / sigma_val) * G).long().numpy()
    a = alpha.numpy()
    v = np.where(a <= 0, 0.0, np.ceil(a)).astype(np.int64)
    t = tmp.numpy()
    old = t.copy()
    for n in range(c.shape[0]):  # sequential: a later point overwrites an earlier one's cell
        i, j, k = c[n]
        t[i, j, k] = np.int64(old[i, j, k] + v[n]).astype(np.int8)
    pos = t > 0
    if pos.sum() == 0:
        grid[...] = True
    else:
        grid[torch.from_numpy(pos)] = True
    t[pos] = 0


def block_pattern_grid(G: int = 256) -> torch.Tensor:
    """The mixed occupancy grid of golden G13: blocks of 8^3 cells, False where (bx + 2 by + 3 bz) % 3 == 0."""
    b = torch.arange(G) // 8
    return ((b[:, None, None] + 2 * b[None, :, None] + 3 * b[None, None, :]) % 3) != 0


def render_masked(o, d, t, dir_norm, tables, scales, mu, sigma, mlp_params, grid, grid_mu, grid_sigma, num_freq: int = 4):
    """vol_render's masked branch (vol_renderer.py:209-221): the MLP runs on the kept samples only; sigma and rgb of
    the others are the zeros they were initialised with (so nothing flows back through them).
    Returns (Cr, sigma [R,S], rgb [R,S,3], mask [N])."""
    R, S = o.shape[0], t.shape[0]
    pts = sample_points(o, d, t).reshape(-1, 3)
    mask = occupancy_mask(pts, grid, grid_mu, grid_sigma)
    feat = hash_encode(pts, tables, scales, mu, sigma)
    dirs = dir_encode(d[:, None, :].expand(R, S, 3).reshape(-1, 3), num_freq)
    out = mlp_forward(feat[mask], dirs[mask], mlp_params)
    sig = torch.zeros((R * S, 1), dtype=out.dtype)
    rgb = torch.zeros((R * S, 3), dtype=out.dtype)
    sig[mask] = out[:, 3:4]
    rgb[mask] = out[:, 0:3]
    sig, rgb = sig.reshape(R, S), rgb.reshape(R, S, 3)
    Cr, _ = composite(t, rgb, sig, dir_norm)
    return Cr, sig, rgb, mask


def train_loss(Cr: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """train_hash2.py:177,221: MSE(Cr,gt)+MSE(Cf,gt) with Cf is Cr => 2*MSE."""
    return 2 * torch.mean((Cr - gt) ** 2)


def psnr(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """helper.py:301-304."""
    return 10 * torch.log10(1.0 / torch.mean((pred - target) ** 2))


def make_optimizers(table_params, mlp_params, total_steps: int):
    """train_hash2.py:141-142,156-162: Adam(lr .05) on tables, AdamW(lr .005) on the MLP,
    cosine annealing to 1e-4 over total_steps."""
    oe = torch.optim.Adam(list(table_params), lr=0.05)
    om = torch.optim.AdamW(list(mlp_params), lr=0.005)
    se = torch.optim.lr_scheduler.CosineAnnealingLR(oe, T_max=total_steps, eta_min=1e-4)
    sm = torch.optim.lr_scheduler.CosineAnnealingLR(om, T_max=total_steps, eta_min=1e-4)
    return oe, om, se, sm


def train_step(batch, t, tables, scales, mu, sigma, mlp_params, opts, num_freq: int = 4):
    """One iteration of train_hash2.py:211-239 in fp32 (no GradScaler: it is a no-op for
    fp32/bf16).  tables / mlp_params hold leaf tensors with requires_grad."""
    o, d, dir_norm, gt = batch
    oe, om, se, sm = opts
    Cr, _, _ = render(o, d, t, dir_norm, tables, scales, mu, sigma, mlp_params, num_freq)
    loss = train_loss(Cr, gt)
    loss.backward()
    oe.step(); om.step(); se.step(); sm.step()
    om.zero_grad(set_to_none=True); oe.zero_grad(set_to_none=True)
    return loss.detach()


# --------------------------------------------------------------------------------------
# synthetic lego-shaped workload.  Shared by bench.py's cpu_baseline leg and tests.
# --------------------------------------------------------------------------------------
def synthetic_rays(R: int, seed: int = 0, radius: float = 4.03):
    """R rays from cameras on the upper hemisphere looking at the origin, numpy PCG64 stream.
    Returns o[R,3], d_unit[R,3], dir_norm[R,1], gt[R,3] (a smooth analytic colour) as fp32."""
    rng = np.random.default_rng(seed)
    th = rng.uniform(0, 2 * np.pi, R)
    ph = rng.uniform(0.05, 0.5 * np.pi, R)
    o = radius * np.stack([np.cos(th) * np.sin(ph), np.sin(th) * np.sin(ph), np.cos(ph)], -1)
    target = rng.uniform(-0.6, 0.6, (R, 3))
    d = target - o
    nrm = np.linalg.norm(d, axis=-1, keepdims=True)
    d = d / nrm
    gt = 0.5 + 0.5 * np.sin(3.0 * target + np.array([0.0, 1.0, 2.0]))
    dn = rng.uniform(1.0, 1.2, (R, 1))
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return f(o), f(d), f(dn), f(gt)


def bbox_mu_sigma(o: torch.Tensor, d: torch.Tensor, near: float = 2.0, far: float = 6.0):
    """find_bounding_box semantics (helper.py:109-141): AABB of the ray points at
    t in {near, far+1.5}; mu = min corner, sigma = diagonal length (train_hash2.py:117-119).
    Returns (min_bound[3], max_bound[3], sigma 0-d)."""
    tt = torch.tensor([near, far + 1.5], dtype=torch.float32)
    pts = sample_points(o, d, tt).reshape(-1, 3)
    mn = pts.min(dim=0).values
    mx = pts.max(dim=0).values
    return mn, mx, ((mx - mn) ** 2).sum().sqrt()


# --------------------------------------------------------------------------------------
# hierarchical second pass                        helper.py:23-51, vol_renderer.py:225-242
# --------------------------------------------------------------------------------------
def hierarchical_sample(o, d, t, weights, n_samples: int, tn: float, tf: float, u01: torch.Tensor, samples01: torch.Tensor):
    """The reference's resampling with its two uniform draws made explicit (u01 [R,S], samples01 [n_samples]):
    negative weights -> 0 (helper.py:36); pdf = (w+1e-5)/sum (:38); cdf = cumsum (:39); inds = searchsorted(cdf, u,
    right=True) clamped to [0, n-1] (:41,44); the new depths are NOT drawn inside the selected bins: they index a
    single shared vector samples = samples01*(tf-tn)+tn (:43,45); merged with t and sorted (:46-47).
    Returns (points [R, S+n, 3], t_fine [R, S+n])."""
    w = weights.reshape(weights.shape[0], -1).clone()
    w[w < 0] = 0
    pdf = (w + 1e-5) / torch.sum(w + 1e-5, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    inds = torch.searchsorted(cdf, u01.contiguous(), right=True).clamp(0, n_samples - 1)
    smp = (samples01 * (tf - tn) + tn)[inds]
    tt, _ = torch.sort(torch.cat([t.expand(inds.shape[0], t.shape[-1]), smp], dim=-1), dim=-1)
    return o[..., None, :] + d[..., None, :] * tt[..., :, None], tt


def composite_per_ray(t2: torch.Tensor, rgb: torch.Tensor, sigma: torch.Tensor, dir_norm):
    """calc_color with a per-ray t [R,S] (helper.py:65-71 handles both ranks)."""
    delta = torch.zeros_like(t2)
    delta[:, :-1] = t2[:, 1:] - t2[:, :-1]
    delta = delta * dir_norm
    sig = torch.where(sigma >= -10, sigma, torch.full_like(sigma, -10.0))
    p = sig * delta
    alpha = 1 - torch.exp(-p)
    Tr = torch.exp(-torch.cumsum(p, dim=-1))
    Tr = torch.cat([torch.ones_like(Tr[:, :1]), Tr[:, :-1]], dim=-1)
    w = Tr * alpha
    return (w[:, :, None] * rgb).sum(dim=-2), w[:, :, None]


def render_hierarchical(o, d, t, dir_norm, tables, scales, mu, sigma, mlp_params, u01, samples01, tn=2.0, tf=6.0, num_freq=4):
    """vol_render(hierarchical=True): coarse pass, resample, fine pass on S+S sorted depths.  Returns (Cr, Cf)."""
    R, S = o.shape[0], t.shape[0]
    pts = sample_points(o, d, t).reshape(-1, 3)
    feat = hash_encode(pts, tables, scales, mu, sigma)
    pe = dir_encode(d[:, None, :].expand(R, S, 3).reshape(-1, 3), num_freq)
    out = mlp_forward(feat, pe, mlp_params)
    Cr, wts = composite(t, out[:, 0:3].reshape(R, S, 3), out[:, 3].reshape(R, S), dir_norm)
    pts_f, t_f = hierarchical_sample(o, d, t, wts.detach(), S, tn, tf, u01, samples01)
    S2 = t_f.shape[1]
    feat_f = hash_encode(pts_f.reshape(-1, 3), tables, scales, mu, sigma)
    pe_f = dir_encode(d[:, None, :].expand(R, S2, 3).reshape(-1, 3), num_freq)
    out_f = mlp_forward(feat_f, pe_f, mlp_params)
    Cf, _ = composite_per_ray(t_f, out_f[:, 0:3].reshape(R, S2, 3), out_f[:, 3].reshape(R, S2), dir_norm)
    return Cr, Cf


# --------------------------------------------------------------------------------------
# multi-view-consistent synthetic scene (analytic density/colour field inside |x| < 1.2)
# --------------------------------------------------------------------------------------
def analytic_field(x: torch.Tensor):
    """A smooth 'lego-like' solid: union of three soft boxes and a sphere.  Returns (sigma >= 0 [..], rgb [..,3])."""
    def box(c, h):
        q = (x - torch.tensor(c, dtype=x.dtype, device=x.device)).abs() - torch.tensor(h, dtype=x.dtype, device=x.device)
        return q.max(dim=-1).values  # signed distance-like (negative inside)
    d = torch.minimum(torch.minimum(box((0.0, 0.0, -0.3), (0.9, 0.6, 0.2)), box((-0.3, 0.0, 0.1), (0.4, 0.35, 0.25))),
                      box((0.45, 0.1, 0.15), (0.2, 0.45, 0.3)))
    d = torch.minimum(d, (x - torch.tensor((0.0, -0.2, 0.55), dtype=x.dtype, device=x.device)).norm(dim=-1) - 0.3)
    sigma = 25.0 * torch.sigmoid(-d / 0.03)
    rgb = 0.5 + 0.5 * torch.sin(4.0 * x + torch.tensor((0.0, 2.0, 4.0), dtype=x.dtype, device=x.device))
    return sigma, rgb


def synthetic_scene_rays(R: int, seed: int = 0, radius: float = 4.03, near: float = 2.0, far: float = 6.0, quad: int = 384,
                         device="cpu"):
    """Rays as synthetic_rays(), but with ground truth rendered from analytic_field by the reference's own compositing
    rule on a fine uniform quadrature (so the target is a consistent radiance field and PSNR means something).
    dir_norm = 1.  Returns o, d, dir_norm[R,1], gt[R,3] on `device`."""
    o, d, _, _ = synthetic_rays(R, seed=seed, radius=radius)
    o, d = o.to(device), d.to(device)
    t = torch.linspace(near, far, quad, device=device)
    gts = []
    for i in range(0, R, 4096):
        pts = sample_points(o[i:i + 4096], d[i:i + 4096], t)
        sg, rgb = analytic_field(pts)
        gts.append(composite(t, rgb, sg, 1)[0])
    return o, d, torch.ones((R, 1), device=device), torch.cat(gts).clamp(0, 1)


# --------------------------------------------------------------------------------------
# BASELINE config 1 (CPU plumbing): the vanilla positional-encoding NeRF     vol_renderer.py:12-86, train.py:16-19
# --------------------------------------------------------------------------------------
# Outside the accelerated path ("OUT OF SCOPE for kernels") - restated here only so that the reference's own
# CPU-runnable configuration has a pinned check: PositionalEncoder(3, 10) on points and directions, the 8-layer MLP
# with a skip connection, then the same masked assign + compositing as the hash path.
def vanilla_nerf_init(seed: int, d_input: int = 60, d_viewdirs: int = 60, n_layers: int = 8, d_filter: int = 256, skip=(4,)) -> dict:
    """nn.Linear default init from a numpy stream; keys follow the reference's state dict (`layers.i`, `alpha_out`,
    `rgb_filters`, `branch`, `output`)."""
    rng = np.random.default_rng(seed)
    shapes = {"layers.0": (d_filter, d_input)}
    for i in range(n_layers - 1):  # layer i+1 follows the concat when i is a skip index (vol_renderer.py:33-35)
        shapes[f"layers.{i + 1}"] = (d_filter, d_filter + d_input if i in skip else d_filter)
    shapes.update({"alpha_out": (1, d_filter), "rgb_filters": (d_filter, d_filter), "branch": (d_filter // 2, d_filter + d_viewdirs),
                   "output": (3, d_filter // 2)})
    p = {}
    for k, (o, i) in shapes.items():
        b = 1.0 / math.sqrt(i)
        p[k + ".weight"] = torch.from_numpy(rng.uniform(-b, b, (o, i)).astype(np.float32))
        p[k + ".bias"] = torch.from_numpy(rng.uniform(-b, b, (o,)).astype(np.float32))
    return p


def vanilla_nerf_forward(x: torch.Tensor, viewdirs: torch.Tensor, p: dict, n_layers: int = 8, skip=(4,)) -> torch.Tensor:
    """NeRF.forward with view directions (vol_renderer.py:50-86): ReLU after every trunk layer, the input re-attached
    after layer `skip`; alpha = sigmoid(alpha_out(h)); rgb = relu(output(relu(branch(cat(rgb_filters(h), dirs)))));
    returns [N,4] = (rgb, alpha)."""
    lin = torch.nn.functional.linear
    h = x
    for i in range(n_layers):
        h = torch.relu(lin(h, p[f"layers.{i}.weight"], p[f"layers.{i}.bias"]))
        if i in skip:
            h = torch.cat([h, x], dim=-1)
    alpha = torch.sigmoid(lin(h, p["alpha_out.weight"], p["alpha_out.bias"]))
    c = lin(h, p["rgb_filters.weight"], p["rgb_filters.bias"])
    c = torch.relu(lin(torch.cat([c, viewdirs], dim=-1), p["branch.weight"], p["branch.bias"]))
    c = torch.relu(lin(c, p["output.weight"], p["output.bias"]))
    return torch.cat([c, alpha], dim=-1)


def render_vanilla(o, d, t, dir_norm, p, num_freq: int = 10, n_layers: int = 8, skip=(4,)):
    """vol_render with Pos_encode = Dir_encode = PositionalEncoder(3, num_freq), all-true occupancy grid:
    returns (Cr [R,3], sigma [R,S], rgb [R,S,3])."""
    R, S = o.shape[0], t.shape[0]
    pts = sample_points(o, d, t).reshape(-1, 3)
    dirs = d[:, None, :].expand(R, S, 3).reshape(-1, 3)
    out = vanilla_nerf_forward(dir_encode(pts, num_freq), dir_encode(dirs, num_freq), p, n_layers, skip)
    sig, rgb = out[:, 3].reshape(R, S), out[:, 0:3].reshape(R, S, 3)
    Cr, _ = composite(t, rgb, sig, dir_norm)
    return Cr, sig, rgb
