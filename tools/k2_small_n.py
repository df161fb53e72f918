"""K2 at small batches: the global-atomics kernel (algo 1) against the LDS kernels (algo 2) - where should `auto` switch?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import torch, ref_cpu
from hbr_amd import ops
from hbr_amd._lib import PLANAR
dev = "cuda:0"
L, T, S = 16, 2 ** 16, 32
sc = ref_cpu.level_scales(16, 2048.0, L)
for R in (16, 64, 256, 1024, 2048, 4096):
    o, d, dn, gt = ref_cpu.synthetic_rays(R, seed=0)
    mn, mx, sig = ref_cpu.bbox_mu_sigma(o, d)
    geom = ops.HashGeom(tuple(float(v) for v in sc), tuple(float(v) for v in mn), float(sig), T, 2)
    t = ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.rand(S)).to(dev)
    rays = (o.to(dev), d.to(dev), t)
    dy = torch.rand((L, R * S, 2), device=dev).to(torch.bfloat16)
    res = []
    for algo in (1, 2):
        dt = torch.zeros((L, T, 2), device=dev)
        for _ in range(3):
            ops.hash_encode_bwd(geom, dy, dt, rays=rays, layout=PLANAR, algo=algo, overwrite=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.hash_encode_bwd(geom, dy, dt, rays=rays, layout=PLANAR, algo=algo, overwrite=True)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20)
    print(f"N = {R * S:7d} points: algo 1 (global float atomics, incl. the zero-fill) {res[0]:.4f} ms   algo 2 (LDS fixed point, overwrite) {res[1]:.4f} ms", flush=True)
