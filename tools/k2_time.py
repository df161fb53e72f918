"""Time only the K2 scatter kernel at the BASELINE size, fp32 and bf16 dy (HBR_LIB selects the build under test;
K2_ALGO: 2 = the LDS kernels (default), 1 = global float atomics, 0 = the library's own choice)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import torch, ref_cpu
from hbr_amd import ops
from hbr_amd._lib import PLANAR
dev = "cuda:0"
R, S, L, T = int(os.environ.get('K2_RAYS', 16000)), 128, 16, 2 ** int(os.environ.get('K2_LOG2T', 16))
o, d, dn, gt = ref_cpu.synthetic_rays(R, seed=0)
mn, mx, sig = ref_cpu.bbox_mu_sigma(o, d)
sc = ref_cpu.level_scales(16, 2048.0, L)
geom = ops.HashGeom(tuple(float(v) for v in sc), tuple(float(v) for v in mn), float(sig), T, 2)
t = ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.rand(S)).to(dev)
rays = (o.to(dev), d.to(dev), t)
dt = torch.zeros((L, T, 2), device=dev)
ALGO = int(os.environ.get('K2_ALGO', 2))
for dtype in (torch.float32, torch.bfloat16):
    dy = torch.rand((L, R * S, 2), device=dev).to(dtype)
    for _ in range(3):
        ops.hash_encode_bwd(geom, dy, dt, rays=rays, layout=PLANAR, algo=ALGO)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.hash_encode_bwd(geom, dy, dt, rays=rays, layout=PLANAR, algo=ALGO)
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("HBR_LIB", "default"), f"T=2^{os.environ.get('K2_LOG2T', 16)} R={R}", dtype, f"algo {ALGO} hash_bwd {e0.elapsed_time(e1) / 10:.4f} ms", flush=True)
