"""K4 (bf16 MLP backward) with optional outputs switched off at run time: what the d feat part (one dense + 8 stores + the
running maxima) costs in situ."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import torch, ref_cpu
from hbr_amd import ops
from hbr_amd._lib import PLANAR, BF16
dev = "cuda:0"
torch.manual_seed(0)
R, S = 16000, 128
N = R * S
o, d, dn, gt = ref_cpu.synthetic_rays(R, seed=0)
pe = ops.dir_encode(d.to(dev), 4)
feat = (torch.randn((16, N, 2), device=dev) * 0.3).to(torch.bfloat16)
amax = torch.zeros(16, device=dev)
P = torch.cat([v.reshape(-1) for v in ref_cpu.mlp_init(0).values()]).to(dev)
dout = torch.randn((N, 4), device=dev)
dP = torch.zeros_like(P)
def run(**kw):
    for _ in range(3): ops.mlp_bwd(feat, PLANAR, pe, S, P, BF16, dout, dP, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.mlp_bwd(feat, PLANAR, pe, S, P, BF16, dout, dP, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20
for rep in range(2):
    print("full (d feat + maxima) %.4f ms | no maxima %.4f | no d feat %.4f" % (run(absmax_out=amax), run(), run(need_dfeat=False)), flush=True)
