"""Time only the bf16 MLP forward kernel at the BASELINE size (HBR_LIB selects the build under test)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import torch, ref_cpu
from hbr_amd import ops
from hbr_amd._lib import PLANAR, BF16
dev = "cuda:0"
R, S = 16000, 128
N = R * S
o, d, dn, gt = ref_cpu.synthetic_rays(R, seed=0)
pe = ops.dir_encode(d.to(dev), 4)
feat = (torch.randn((16, N, 2), device=dev) * 0.3).to(torch.bfloat16)
P = torch.cat([v.reshape(-1) for v in ref_cpu.mlp_init(0).values()]).to(dev)
for _ in range(3):
    ops.mlp_fwd(feat, PLANAR, pe, S, P, BF16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    ops.mlp_fwd(feat, PLANAR, pe, S, P, BF16)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("HBR_LIB", "default"), f"mlp_fwd bf16 (pack + kernel) {e0.elapsed_time(e1) / 50:.4f} ms", flush=True)
