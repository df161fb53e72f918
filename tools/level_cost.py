"""Per-level cost of the hash kernels: time K1/K2 with all 16 levels set to the same scale."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import torch, ref_cpu
from hbr_amd import ops
from hbr_amd._lib import PLANAR
dev = "cuda:0"
R, S, L, T = 16000, 128, 16, 2 ** 16
N = R * S
o, d, dn, gt = ref_cpu.synthetic_rays(R, seed=0)
mn, mx, sig = ref_cpu.bbox_mu_sigma(o, d)
t = ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.rand(S)).to(dev)
o, d = o.to(dev), d.to(dev)
tab = (torch.rand((L, T, 2), device=dev) - 0.5)
sc = ref_cpu.level_scales(16, 2048.0, L)
dy = torch.rand((L, N, 2), device=dev)
dt = torch.zeros((L, T, 2), device=dev)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("sigma", float(sig), "bbox", (mx - mn).tolist())
for l in range(L):
    geom = ops.HashGeom(tuple([float(sc[l])] * L), tuple(float(v) for v in mn), float(sig), T, 2)
    f = timeit(lambda: ops.hash_encode_fwd(geom, tab, rays=(o, d, t), layout=PLANAR))
    b = timeit(lambda: ops.hash_encode_bwd(geom, dy, dt, rays=(o, d, t), layout=PLANAR, algo=2))
    print(f"level {l:2d} scale {float(sc[l]):8.2f}: fwd {f/16*1e3:7.1f} us/level   bwd {b/16*1e3:7.1f} us/level", flush=True)
