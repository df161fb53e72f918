"""Time only the bf16 MLP backward kernel at the BASELINE size (HBR_LIB selects the build under test).
K4_RENDER=1: the one-launch MLP forward + compositing + loss + backward (hbr_mlp_render_bwd) instead."""
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
feat = (torch.randn((16, N, 2), device=dev) * 0.3).to(torch.bfloat16 if os.environ.get("FEAT", "bf16") == "bf16" else torch.float32)
amax = torch.zeros(16, device=dev)
P = torch.cat([v.reshape(-1) for v in ref_cpu.mlp_init(0).values()]).to(dev)
dout = torch.randn((N, 4), device=dev)
dP = torch.zeros_like(P)
if os.environ.get("K4_RENDER") == "1":  # same harness, the render variant: d out is formed in the kernel
    t = ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.rand(S)).to(dev)
    dn_d, gt_d = dn.reshape(-1).to(dev), gt.to(dev)
    _bwd = ops.mlp_bwd
    ops.mlp_bwd = lambda feat, lay, pe, S_, P, prec, dout, dP, absmax_out=None: ops.mlp_render_bwd(feat, pe, P, prec, t, dn_d, gt_d, dP, absmax_out=absmax_out)[1]
for _ in range(3):
    ops.mlp_bwd(feat, PLANAR, pe, S, P, BF16, dout, dP, absmax_out=amax)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.mlp_bwd(feat, PLANAR, pe, S, P, BF16, dout, dP, absmax_out=amax)
e1.record(); torch.cuda.synchronize()
dP.zero_()
df = ops.mlp_bwd(feat, PLANAR, pe, S, P, BF16, dout, dP, absmax_out=amax)
torch.cuda.synchronize()
import hashlib
# bit-for-bit fingerprints: d feat and absmax do not depend on summation order; dP does only through the order in which
# a wave adds its tiles (same grid => same order)
print("fingerprint dfeat", hashlib.sha1(df.view(torch.int16 if df.dtype == torch.bfloat16 else torch.int32).cpu().numpy().tobytes()).hexdigest()[:16],
      "absmax", hashlib.sha1(amax.cpu().numpy().tobytes()).hexdigest()[:16], "dP sum", float(dP.double().abs().sum()), flush=True)
print(os.environ.get("HBR_LIB", "default"), f"mlp_bwd bf16 {e0.elapsed_time(e1) / 20:.4f} ms", flush=True)
