"""Second (hierarchical) render pass of vol_render (reference vol_renderer.py:225-242, helper.py:23-51).
Flag-gated in the reference (train_hash2.py:34, default off).  The resampling is index plumbing on the GPU
(cumsum / searchsorted / sort in torch); the field evaluation and compositing of the 2S merged depths run on the same
kernels as the first pass (ops.RenderFn with per-ray t)."""
from __future__ import annotations

from . import ops
from .helper import hierarchical_sampling


def render_fine(renderer, mlp, rays_d, rays_o, t, wts, num_samples, dir_norm):
    rng = renderer.fine_rng() if callable(getattr(renderer, "fine_rng", None)) else (None, None)
    _, t_fine = hierarchical_sampling(rays_o, rays_d, z_vals=t, weights=wts, n_samples=num_samples, tn=float(renderer.near),
                                      tf=float(renderer.far), u=rng[0], samples01=rng[1])
    enc = renderer.Pos_encode
    stacked = enc.stacked_tables()
    flat, splits = mlp.flat_params()
    tabs = [lvl.weight for lvl in enc.Embedding_list]
    prec = ops.precision_from_autocast()
    Cf, _, _, _ = ops.RenderFn.apply(rays_o, rays_d, t_fine, dir_norm, enc.geometry(), stacked, flat, prec,
                                  renderer.Dir_encode.max_seq_len, splits, prec if renderer.feat_dtype is None else renderer.feat_dtype,
                                  len(tabs), None, *tabs, *mlp._ordered())  # fine pass: no mask (vol_renderer.py:236)
    renderer.last_t_fine = t_fine
    return Cf
