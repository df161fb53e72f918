"""Run only the K2 scatter kernel a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import torch, ref_cpu
from hbr_amd import ops
from hbr_amd._lib import PLANAR
dev = "cuda:0"
R, S, L, T = 16000, 128, 16, 2 ** 16
o, d, dn, gt = ref_cpu.synthetic_rays(R, seed=0)
mn, mx, sig = ref_cpu.bbox_mu_sigma(o, d)
sc = ref_cpu.level_scales(16, 2048.0, L)
geom = ops.HashGeom(tuple(float(v) for v in sc), tuple(float(v) for v in mn), float(sig), T, 2)
t = ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.rand(S)).to(dev)
o, d = o.to(dev), d.to(dev)
dy = torch.rand((L, R * S, 2), device=dev)
if os.environ.get("K2_BF16", "1") == "1":
    dy = dy.bfloat16()
dt = torch.zeros((L, T, 2), device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else "bwd"
for _ in range(3):
    if which == "bwd":
        ops.hash_encode_bwd(geom, dy, dt, rays=(o, d, t), layout=PLANAR, algo=2)
    else:
        ops.hash_encode_fwd(geom, dt, rays=(o, d, t), layout=PLANAR)
torch.cuda.synchronize()
