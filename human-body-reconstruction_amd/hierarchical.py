"""Second (hierarchical) render pass of vol_render (reference vol_renderer.py:225-242, helper.py:23-51).
Flag-gated in the reference (train_hash2.py:34, default off); SURVEY 8(f4) schedules it after the main path."""


def render_fine(renderer, mlp, rays_d, rays_o, t, wts, num_samples, dir_norm):
    raise NotImplementedError("hierarchical=True (inverse-CDF resampling + second pass) is not built yet; "
                              "pass hierarchical=False as train_hash2.py does by default")
