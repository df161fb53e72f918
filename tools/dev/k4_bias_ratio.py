"""ON THE GPU BOX: per-parameter ratio of the bf16 MLP backward's gradients, variant library vs default build."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch, ref_cpu
    from hbr_amd import ops
    from hbr_amd._lib import PLANAR, BF16
    N, out = int(sys.argv[2]), sys.argv[3]
    g = torch.Generator().manual_seed(5)
    feat = (torch.randn((16, N, 2), generator=g) * 0.3).to("cuda:0")
    pe = ops.dir_encode(torch.nn.functional.normalize(torch.randn((N, 3), generator=g), dim=1).to("cuda:0"), 4)
    dout = torch.randn((N, 4), generator=g).to("cuda:0")
    P = torch.cat([v.reshape(-1) for v in ref_cpu.mlp_init(2).values()]).to("cuda:0")
    dP = torch.zeros_like(P)
    ops.mlp_bwd(feat, PLANAR, pe, 1, P, BF16, dout, dP)
    torch.cuda.synchronize()
    np.save(out, dP.cpu().numpy())
    sys.exit(0)
import ref_cpu
var, N = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4096
res = []
for name, lib in (("default", None), ("variant", os.path.abspath(var))):
    env = dict(os.environ); env.pop("HBR_LIB", None)
    if lib: env["HBR_LIB"] = lib
    out = f"/tmp/k4ratio_{name}.npy"
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(N), out], env=env, check=True)
    res.append(np.load(out))
d, v = res
off = 0
for k, t in ref_cpu.mlp_init(2).items():
    n = t.numel()
    a, b = d[off:off + n], v[off:off + n]
    num = float(np.dot(a, b)) / max(float(np.dot(a, a)), 1e-30)
    print(f"{k:24s} n={n:5d}  |default| {np.abs(a).max():.3e}  variant/default (least squares) {num:.4f}  max|diff| {np.abs(a - b).max():.3e}")
    off += n
