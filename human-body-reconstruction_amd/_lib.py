"""ctypes binding of libhbr_hip.so (C ABI: include/hbr_hip.h).  Fails loudly when the library is missing."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# HBR_LIB overrides the library path (A/B timing of ablation builds only)
LIB_PATH = os.environ.get("HBR_LIB") or os.path.join(_HERE, "libhbr_hip.so")

ROWS, PLANAR = 0, 1
F32, BF16 = 0, 1
IMAGE_READY = 0x100  # hbr_hip.h: OR-ed into hbr_mlp_bwd's precision
OVERWRITE = 0x200    # hbr_hip.h: OR-ed into hbr_hash_encode_bwd's algo / hbr_mlp_bwd's precision
EUNSUPPORTED = -2
MLP_PARAM_FLOATS = 14227
VERSION = 300           # HBR_VERSION of include/hbr_hip.h this binding was written against

_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); must list every symbol declared in include/hbr_hip.h
SIGNATURES = {
    "hbr_version": (_i, []),
    "hbr_strerror": (C.c_char_p, [_i]),
    "hbr_device_ok": (_i, []),
    "hbr_hash_encode_fwd": (_i, [_p, _p, _p, _p, _l, _l, _p, _p, _p, _f, _i, _l, _i, _p, _i, _l, _i, _p]),
    "hbr_hash_encode_bwd": (_i, [_p, _p, _p, _p, _l, _l, _p, _i, _l, _i, _p, _p, _p, _f, _i, _l, _i, _p, _i, _p, _l, _p]),
    "hbr_hash_bwd_workspace_bytes": (_l, [_l, _i, _l, _i, _i]),
    "hbr_hash_bwd_workspace_bytes_min": (_l, [_l, _i, _l, _i, _i]),
    "hbr_composite_fwd": (_i, [_p, _l, _p, _l, _p, _l, _p, _l, _l, _p, _p, _p]),
    "hbr_composite_bwd": (_i, [_p, _l, _p, _l, _p, _l, _p, _l, _l, _p, _p, _p, _p, _p]),
    "hbr_strat_sample": (_i, [_f, _f, _l, _p, C.c_uint64, C.c_uint64, _p, _p]),
    "hbr_occupancy_mask": (_i, [_p, _p, _p, _p, _l, _l, _p, _i, _p, _f, _p, _p]),
    "hbr_dir_encode": (_i, [_p, _l, _i, _i, _p, _p]),
    "hbr_mlp_workspace_bytes": (_l, [_i]),
    "hbr_mlp_fwd": (_i, [_p, _i, _l, _i, _p, _l, _l, _p, _i, _p, _p, _p, _l, _p]),
    "hbr_mlp_bwd": (_i, [_p, _i, _l, _i, _p, _l, _l, _p, _i, _p, _p, _p, _p, _p, _l, _p]),
    "hbr_mlp_render_bwd": (_i, [_p, _i, _l, _i, _p, _l, _l, _p, _i, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _l, _p]),
    "hbr_render_fwd_workspace_bytes": (_l, [_l, _l, _i, _i, _i, _i]),
    "hbr_render_fwd": (_i, [_p, _p, _p, _p, _l, _l, _p, _p, _p, _f, _i, _l, _i, _p, _i, _i, _p, _p, _p, _p, _p, _l, _p]),
    "hbr_mse2_workspace_bytes": (_l, []),
    "hbr_mse2_loss_fwd_bwd": (_i, [_p, _p, _l, _f, _p, _p, _p, _p]),
    "hbr_adam_step": (_i, [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _l, _f, _p]),
    "hbr_adam_step_multi": (_i, [_i, _p, _p]),
    "hbr_render_prologue": (_i, [_f, _f, _l, _p, C.c_uint64, C.c_uint64, _p, _p, _l, _p, _p, _i, _p, _l, _p]),
    "hbr_hierarchical_resample": (_i, [_p, _p, _l, _p, _p, C.c_uint64, C.c_uint64, _f, _f, _l, _l, _l, _p, _p]),
    "hbr_occupancy_update_workspace_bytes": (_l, [_i]),
    "hbr_occupancy_update": (_i, [_p, _p, _p, _p, _l, _l, _p, _p, _p, _i, _p, _f, _p, _l, _p]),
    "hbr_composite_loss_workspace_bytes": (_l, [_l]),
    "hbr_composite_loss_fwd_bwd": (_i, [_p, _l, _p, _l, _p, _l, _p, _l, _l, _p, _f, _p, _p, _p, _p, _p, _p, _p]),
}


class AdamSegment(C.Structure):
    """HbrAdamSegment of include/hbr_hip.h"""
    _fields_ = [("p", _p), ("g", _p), ("m", _p), ("v", _p), ("n", _l), ("lr", _f), ("beta1", _f), ("beta2", _f), ("eps", _f),
                ("weight_decay", _f), ("step", _l), ("grad_scale", _f)]

_lib = None


class HbrError(RuntimeError):
    pass


def kernel_source_sha() -> str:
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.h, csrc/build.sh, include/hbr_hip.h), sorted by name: what a
    stored counter profile (profiles/pmc_*.json) was taken on, independent of link-time details of the .so."""
    import glob
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(_HERE)
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) +
                   [os.path.join(_HERE, "csrc", "build.sh"), os.path.join(root, "include", "hbr_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libhbr_hip.so (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["bash", os.path.join(_HERE, "csrc", "build.sh")], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode:
        raise HbrError("building libhbr_hip.so failed")
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HbrError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU/eager fallback for this path)")
        # torch first: it ships its own HIP runtime (torch/lib/libamdhip64.so).  Loaded before the library, that runtime
        # also satisfies the library's libamdhip64.so.7 dependency - ONE runtime in the process.  The other way round
        # (library first: /opt/rocm's runtime, then torch's next to it) the process holds two, and whichever initialises
        # second does not see the device (seen as hbr_device_ok() == 0 when build() and smoke() shared a process).
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the .so does not export it
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise HbrError(f"{what}: {lib().hbr_strerror(rc).decode()} (code {rc})")


def require_gpu(t) -> None:
    if not t.is_cuda:
        raise HbrError("hbr_amd ops need tensors on an MI355X (cuda) device; there is no CPU fallback")
