"""MI355X-native hash-NeRF render/train hot path (drop-in for the reference's hash_encoding.py +
vol_renderer.py + MLP_3D path as driven by train_hash2.py).

Import as `hbr_amd` (alias package at the repo root).  Module names mirror the reference's so that
`from hash_encoding import *` becomes `from hbr_amd.hash_encoding import *`:

    hbr_amd.hash_encoding.HashEncoder        hbr_amd.encoder.PositionalEncoder
    hbr_amd.test_hash.MLP_3D                 hbr_amd.vol_renderer.Volume_Renderer
    hbr_amd.helper.{get_od, strat_sampler, calc_color, find_bounding_box, calc_psnr, ...}

All compute goes through libhbr_hip.so (hand-written HIP for gfx950, C ABI in include/hbr_hip.h).
There is no CPU or eager-PyTorch fallback: calling an op without the library or without an MI355X
raises.
"""
__version__ = "0.1.0"
