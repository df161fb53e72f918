"""CPU: the C-ABI library builds, loads, and exports every symbol include/hbr_hip.h declares; host-side
logic (level scales, aliasing of stacked parameters, state-dict keys, argument validation)."""
import os
import re

import numpy as np
import pytest
import torch

import ref_cpu
from conftest import ROOT, load_golden


@pytest.fixture(scope="module")
def L():
    import hbr_amd._lib as L
    if not os.path.exists(L.LIB_PATH):
        L.build()
    return L


def test_header_symbols_all_exported(L):
    hdr = open(os.path.join(ROOT, "include", "hbr_hip.h")).read()
    declared = set(re.findall(r"\b(hbr_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = L.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in hbr_hip.h but not exported by libhbr_hip.so"
    assert declared == set(L.SIGNATURES), (declared ^ set(L.SIGNATURES))
    # ... and nothing else: the library is linked with -fvisibility=hidden + a version script (no kernel stubs, no
    # C++ helpers, no __hip_cuid_* in the dynamic symbol table)
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if ln.strip()}
    assert exported == declared, sorted(exported ^ declared)
    assert lib.hbr_version() == L.VERSION
    assert lib.hbr_strerror(-2).decode().startswith("configuration not supported")
    # MFMA-fragment image of the weights (34 KiB forward + 28 KiB backward + 6 x 64 biases), then one weight-gradient
    # slab per backward workgroup: 256 workgroups x 4 waves x (6 tiles x 16 + 10 bias) registers x 64 lanes x 4 B
    img = (34 + 10 + 28) * 1024 + 6 * 64 * 4  # 10: one bias k-step per forward output tile
    assert img % 256 == 0
    # ... the per-wave feature-gradient maxima of the backward (16 levels x 1024 waves x 4 B) and one slab of totals
    # ... and (hbr_mlp_render_bwd) one squared-error partial per wave of the backward grid
    slabs = 256 * 4 * (6 * 16 + 10) * 64 * 4 + 16 * 1024 * 4 + 4 * (6 * 16 + 10) * 64 * 4 + 256 * 4 * 4
    assert lib.hbr_mlp_workspace_bytes(L.BF16) == img + slabs
    img32 = (264 + 216) * 256 + 6 * 64 * 4
    assert lib.hbr_mlp_workspace_bytes(L.F32) == (img32 + 255) // 256 * 256 + slabs


def test_argument_validation_without_gpu(L):
    """Validation happens before any launch, so these return codes are testable on CPU."""
    import ctypes as C
    lib = L.lib()
    sc = (C.c_float * 16)(*[16.0] * 16)
    mu = (C.c_float * 3)(0, 0, 0)
    # null tables
    assert lib.hbr_hash_encode_fwd(None, None, None, None, 4, 1, None, sc, mu, 1.0, 16, 1024, 2, None, 0, 32, 0, None) == -1
    # F != 2 unsupported
    assert lib.hbr_hash_encode_fwd(8, None, None, None, 4, 1, 8, sc, mu, 1.0, 16, 1024, 4, 8, 0, 64, 0, None) == -2
    # rows stride too small
    assert lib.hbr_hash_encode_fwd(8, None, None, None, 4, 1, 8, sc, mu, 1.0, 16, 1024, 2, 8, 0, 16, 0, None) == -1
    # too many levels
    assert lib.hbr_hash_encode_fwd(8, None, None, None, 4, 1, 8, sc, mu, 1.0, 33, 1024, 2, 8, 1, 0, 0, None) == -1
    # S > 4096 unsupported by the one-wave-per-ray compositor
    assert lib.hbr_composite_fwd(8, 0, 8, 3, 8, 1, None, 4, 5000, 8, None, None) == -2
    # per-ray t stride shorter than S
    assert lib.hbr_composite_fwd(8, 3, 8, 3, 8, 1, None, 4, 16, 8, None, None) == -1
    # workspace too small
    assert lib.hbr_mlp_fwd(16, 0, 32, 0, 16, 4, 1, 16, 1, 16, None, 16, 10, None) == -4
    # K0: null outputs / bad grid size; K2 (validation precedes any launch): algo 2 without its workspace, T too large
    assert lib.hbr_strat_sample(2.0, 6.0, 16, None, 0, 0, None, None) == -1
    assert lib.hbr_occupancy_mask(8, None, None, None, 4, 1, 8, 0, mu, 1.0, 8, None) == -1
    assert lib.hbr_hash_encode_bwd(8, None, None, None, 70000, 1, 8, 1, 0, 0, None, sc, mu, 1.0, 16, 1024, 2, 8, 2, None, 0, None) == -4
    assert lib.hbr_hash_encode_bwd(8, None, None, None, 70000, 1, 8, 1, 0, 0, None, sc, mu, 1.0, 16, 2 ** 29, 2, 8, 2, 8, 64, None) == -2
    assert lib.hbr_hash_bwd_workspace_bytes(2048000, 16, 65536, 2, 0) > lib.hbr_hash_bwd_workspace_bytes_min(2048000, 16, 65536, 2, 0) > 0
    assert lib.hbr_hash_bwd_workspace_bytes(1000, 16, 65536, 2, 0) == 0  # auto: the global-atomics kernel at this size
    # adam: misaligned pointer
    assert lib.hbr_adam_step(4, 16, 16, 16, 8, 0.1, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, None) == -1


def test_hash_encoder_host_logic():
    from hbr_amd.hash_encoding import HashEncoder
    g = load_golden("g2_level_scales.npz")
    enc = HashEncoder(N_max=2048.0, N_min=16, L=16, T=2 ** 10, F=2, dim=3, mu=torch.tensor([-1.0, 0.5, 2.0]),
                      sigma=torch.tensor(3.0), device="cpu")
    assert np.array_equal(enc.level_scales().numpy().view(np.uint32), g["f2048_16"].view(np.uint32))
    enc_i = HashEncoder(N_max=2048, N_min=16, L=16, T=2 ** 10, F=2, dim=3, device="cpu")
    assert np.array_equal(enc_i.level_scales().numpy().view(np.uint32), g["i2048_16"].view(np.uint32))
    geom = enc.geometry()
    assert geom.mu == (-1.0, 0.5, 2.0) and geom.sigma == 3.0 and geom.T == 1024 and geom.L == 16
    assert list(enc.state_dict().keys()) == [f"Embedding_list.{i}.weight" for i in range(16)]
    w = enc.Embedding_list[3].weight
    assert w.shape == (1024, 2) and float(w.abs().max()) <= 1e-4  # U(-1e-4, 1e-4) init
    # the 16 parameters alias one stacked buffer, and keep doing so after load_state_dict / optimizer steps
    st = enc.stacked_tables()
    assert st.shape == (16, 1024, 2) and w.data_ptr() == st[3].data_ptr()
    sd = {k: torch.randn_like(v) for k, v in enc.state_dict().items()}
    enc.load_state_dict(sd)
    assert torch.equal(enc.stacked_tables()[5], sd["Embedding_list.5.weight"])
    opt = torch.optim.Adam(list(enc.Embedding_list.parameters()), lr=0.05)
    for p in enc.Embedding_list.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    assert torch.allclose(enc.stacked_tables()[5], sd["Embedding_list.5.weight"] - 0.05, atol=1e-6)
    # parameters replaced behind our back (e.g. module.to(other device)) get re-stacked, keeping identity
    p7 = enc.Embedding_list[7].weight
    p7.data = p7.data.clone()
    st2 = enc.stacked_tables()
    assert enc.Embedding_list[7].weight is p7 and p7.data_ptr() == st2[7].data_ptr()
    with pytest.raises(NotImplementedError):
        HashEncoder(N_max=64, N_min=16, L=4, dim=2, device="cpu")


def test_mlp_host_logic():
    from hbr_amd.test_hash import MLP_3D
    g = load_golden("g8_render_step.npz")
    m = MLP_3D(num_sig=2, num_col=2, L=16, F=2, d_view=24, max_bound=torch.ones(3), min_bound=-torch.ones(3))
    keys = ["module." + k for k in m.state_dict().keys()]
    assert keys == list(g["state_keys_mlp"])  # same names as the reference under DataParallel
    flat, splits = m.flat_params()
    assert flat.numel() == 14227 and splits[6] == (7312, 7312 + 64 * 39, (64, 39))
    assert m.col_model[0].weight.data_ptr() == flat.data_ptr() + 7312 * 4
    assert sum(p.numel() for p in m.parameters()) == 14227
    with pytest.raises(NotImplementedError):
        MLP_3D(num_sig=3, num_col=2, d_view=24).flat_params()


def test_ops_refuse_cpu_tensors():
    """No CPU / eager fallback: the product path must fail loudly off the GPU."""
    from hbr_amd._lib import HbrError
    from hbr_amd.encoder import PositionalEncoder
    from hbr_amd.hash_encoding import HashEncoder
    enc = HashEncoder(N_max=2048.0, N_min=16, L=16, T=2 ** 10, F=2, dim=3, device="cpu")
    with pytest.raises(HbrError):
        enc(torch.zeros(4, 3))
    with pytest.raises(HbrError):
        PositionalEncoder(3, 4)(torch.zeros(4, 3))


def test_helper_geometry_matches_golden():
    from hbr_amd import helper
    g = load_golden("g7_rays.npz")
    o, d, n = helper.get_od(int(g["H"]), int(g["W"]), torch.from_numpy(g["K"]), torch.from_numpy(g["c2w"]))
    assert np.allclose(o.numpy(), g["o"], atol=1e-6) and np.allclose(d.numpy(), g["d"], atol=1e-6)
    assert np.allclose(n.numpy(), g["n"], rtol=1e-6)
    torch.manual_seed(7)
    t = helper.strat_sampler(torch.tensor(2.0), torch.tensor(6.0), 16, device="cpu")
    assert np.allclose(t.numpy(), g["strat_t"], atol=1e-6)
    p = load_golden("g10_psnr.npz")
    assert np.allclose(helper.calc_psnr(torch.from_numpy(p["a"]), torch.from_numpy(p["b"])).numpy(), p["psnr"], rtol=1e-6)
    # bounding box semantics: AABB at t in {near, far+1.5}
    o, d, dn, _ = ref_cpu.synthetic_rays(64, seed=5)
    mn, mx, _ = ref_cpu.bbox_mu_sigma(o, d, 2.0, 6.0)
    mx2, mn2 = helper.find_bounding_box2([(o, d, dn, None)], 2.0, 6.0)
    assert torch.allclose(mx2, mx) and torch.allclose(mn2, mn)


def test_no_valu_write_within_two_slots_of_an_mfma_operand_read(L):
    """Round 4 (DESIGN 3, K4 "root cause"): gfx950 needs 2 wait states between a VALU write of a VGPR and an MFMA reading
    it as A / B / C; hipcc pads the pair only when it knows the writer is a VALU instruction - not when it sits inside
    an `asm` statement.  The round-3 builds carried 23 such pairs (inline-asm v_pk_max_i16 one slot ahead of the
    consuming v_mfma), masked by LDS waits in the shipped kernels and fatal in the two-tile experiment.  The built
    library's disassembly must hold none: every MFMA of every kernel is checked."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mfma_scan", os.path.join(ROOT, "tools", "dev", "mfma_operand_hazard_scan.py"))
    scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scan)
    total, found = scan.scan_library(L.LIB_PATH)
    assert total > 5000, f"only {total} MFMAs found: the disassembly did not cover mlp.hip's kernels"
    assert not found, "\n".join(found[:20])
