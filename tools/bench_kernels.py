"""Scratch timing of the individual kernels at the BASELINE size (not the driver's bench)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import ref_cpu
from hbr_amd import ops
from hbr_amd._lib import PLANAR, ROWS, F32, BF16
dev = "cuda:0"
R, S, L, T = 16000, 128, 16, 2 ** 16
N = R * S
o, d, dn, gt = ref_cpu.synthetic_rays(R, seed=0)
mn, mx, sig = ref_cpu.bbox_mu_sigma(o, d)
sc = ref_cpu.level_scales(16, 2048.0, L)
geom = ops.HashGeom(tuple(float(v) for v in sc), tuple(float(v) for v in mn), float(sig), T, 2)
t = ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.rand(S)).to(dev)
o, d, dn, gt = o.to(dev), d.to(dev), dn.to(dev).reshape(-1), gt.to(dev)
tab = (torch.rand((L, T, 2), device=dev) - 0.5) * 2e-4
P = torch.cat([v.reshape(-1) for v in ref_cpu.mlp_init(0).values()]).to(dev)

def timeit(name, fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:40s} {ms:8.3f} ms  {N/ms/1e6:8.2f} Gsamples/s", flush=True)
    return ms

pe = ops.dir_encode(d, 4)
feat = ops.hash_encode_fwd(geom, tab, rays=(o, d, t), layout=PLANAR)
featb = ops.hash_encode_fwd(geom, tab, rays=(o, d, t), layout=PLANAR, dtype=BF16)
timeit("hash_fwd planar f32 (rays)", lambda: ops.hash_encode_fwd(geom, tab, rays=(o, d, t), layout=PLANAR, out=feat))
timeit("hash_fwd planar bf16 (rays)", lambda: ops.hash_encode_fwd(geom, tab, rays=(o, d, t), layout=PLANAR, out=featb, dtype=BF16))
dy = torch.rand_like(feat)
dt = torch.zeros((L, T, 2), device=dev)
timeit("hash_bwd algo1 atomics", lambda: ops.hash_encode_bwd(geom, dy, dt, rays=(o, d, t), layout=PLANAR, algo=1), n=3)
timeit("hash_bwd algo2 lds", lambda: ops.hash_encode_bwd(geom, dy, dt, rays=(o, d, t), layout=PLANAR, algo=2))
for prec, nm in ((BF16, "bf16"), (F32, "f32")):
    out = ops.mlp_fwd(feat, PLANAR, pe, S, P, prec)
    timeit(f"mlp_fwd {nm}", lambda: ops.mlp_fwd(feat, PLANAR, pe, S, P, prec))
    dout = torch.rand_like(out); dP = torch.zeros_like(P)
    timeit(f"mlp_bwd {nm}", lambda: ops.mlp_bwd(feat, PLANAR, pe, S, P, prec, dout, dP), n=3)
out = ops.mlp_fwd(feat, PLANAR, pe, S, P, BF16)
timeit("composite_fwd", lambda: ops.composite_fwd(t, out.data_ptr(), 4, out.data_ptr() + 12, 4, dn, R, S))
dC = torch.rand((R, 3), device=dev); dO = torch.empty_like(out)
timeit("composite_bwd", lambda: ops.composite_bwd(t, out.data_ptr(), 4, out.data_ptr() + 12, 4, dn, R, S, dC, dO.data_ptr(), dO.data_ptr() + 12))
m, v = torch.zeros_like(tab), torch.zeros_like(tab)
timeit("adam tables", lambda: ops.adam_step(tab.view(-1), dt.view(-1), m.view(-1), v.view(-1), 0.05, 0.9, 0.999, 1e-8, 0.0, 1))
