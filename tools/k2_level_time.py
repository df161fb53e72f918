"""Time K2 (LDS scatter, algo 2) on ONE level at a time at the BASELINE size: the duration of a launch over a single level is
the duration of its slowest workgroup (64 hashed workgroups, or the dense ones) - what a level's workgroups cost."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import torch, ref_cpu
from hbr_amd import ops
from hbr_amd._lib import PLANAR
dev = "cuda:0"
R, S, L, T = 16000, 128, 16, 2 ** 16
o, d, dn, gt = ref_cpu.synthetic_rays(R, seed=0)
mn, mx, sig = ref_cpu.bbox_mu_sigma(o, d)
sc = [float(v) for v in ref_cpu.level_scales(16, 2048.0, L)]
t = ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.rand(S)).to(dev)
rays = (o.to(dev), d.to(dev), t)
torch.manual_seed(0)
dy = torch.rand((L, R * S, 2), device=dev).to(torch.bfloat16)
dt = torch.zeros((L, T, 2), device=dev)
def timed(lo, hi):
    geom = ops.HashGeom(tuple(sc[lo:hi]), tuple(float(v) for v in mn), float(sig), T, 2)
    for _ in range(2):
        ops.hash_encode_bwd(geom, dy[lo:hi], dt[lo:hi], rays=rays, layout=PLANAR, algo=2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.hash_encode_bwd(geom, dy[lo:hi], dt[lo:hi], rays=rays, layout=PLANAR, algo=2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 * 1e3
print("all 16 levels: %.1f us" % timed(0, 16)); print("all 16 levels: %.1f us" % timed(0, 16))
if os.environ.get("ONLY_ALL"): sys.exit(0)
for l in range(16):
    print("level %2d alone: %.1f us" % (l, timed(l, l + 1)), flush=True)
for lo, hi in ((0, 4), (4, 8), (8, 12), (12, 16), (4, 16)):
    print("levels %d..%d: %.1f us" % (lo, hi - 1, timed(lo, hi)), flush=True)
