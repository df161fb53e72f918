"""Where one wave of the bf16 MLP backward spends its cycles (development tool).  Needs a library built with
-DHBR_K4_PROF=1 (tools/variant.sh prof mlp -DHBR_K4_PROF=1) selected through HBR_LIB: wave 0 of workgroup 0 timestamps
its phase boundaries with the shader clock; this prints the cycles per 32-point tile of every phase."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import torch, ref_cpu
from hbr_amd import ops
from hbr_amd._lib import PLANAR, BF16, LIB_PATH
dev = "cuda:0"
R, S = 16000, 128
N = R * S
o, d, dn, gt = ref_cpu.synthetic_rays(R, seed=0)
pe = ops.dir_encode(d.to(dev), 4)
feat = (torch.randn((16, N, 2), device=dev) * 0.3).to(torch.bfloat16)
amax = torch.zeros(16, device=dev)
P = torch.cat([v.reshape(-1) for v in ref_cpu.mlp_init(0).values()]).to(dev)
dout = torch.randn((N, 4), device=dev)
dP = torch.zeros_like(P)
dbg = C.CDLL(LIB_PATH)
buf = (C.c_ulonglong * 64)()
for _ in range(3):
    ops.mlp_bwd(feat, PLANAR, pe, S, P, BF16, dout, dP, absmax_out=amax)
torch.cuda.synchronize()
assert dbg.hbr_debug_k4_prof(buf, 1) == 0
CALLS = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(CALLS):
    ops.mlp_bwd(feat, PLANAR, pe, S, P, BF16, dout, dP, absmax_out=amax)
e1.record(); torch.cuda.synchronize()
assert dbg.hbr_debug_k4_prof(buf, 0) == 0
tiles = CALLS * ((N // 32 + 1023) // 1024)
names = ["issue next tile's loads", "dZ out (expf)"]
for l in ("C3", "C2", "C1", "L3", "L2", "L1+dfeat"):
    names += [f"{l}: put+dense", f"{l}: barrier wait", f"{l}: owner reads+MFMA+epilogue"]
names += ["fwd: inputs arrive + unpack"]
for l in ("L1", "L2", "L3", "C1", "C2"):
    names += [f"fwd {l}: dense", f"fwd {l}: epilogue"]
names += ["fwd C3: dense"]
if buf[36] or buf[32]:  # -DHBR_K4_PROF=3: inside the owner phases (layer order L1 L2 L3 C1 C2 C3 = 0..5)
    for li, l in enumerate(("L1", "L2", "L3", "C1", "C2", "C3")):
        names += [f"  own {l}: source {w}" for w in range(4)] + [f"  own {l}: dense + first fragments arrive"]
tot = sum(buf[i] for i in range(len(names)))
print(f"{os.environ.get('HBR_LIB', 'default')}: {e0.elapsed_time(e1) / CALLS:.4f} ms per call (instrumented); {tot / tiles:.0f} cycles per tile in marked phases")
if tot == 0:
    print(f"  clock held during the sweep: {buf[62] / max(buf[63], 1) * 0.1:.3f} GHz ({buf[62] / tiles:.0f} shader cycles per tile)"); sys.exit(0)
print(f"  clock held during the sweep: {buf[62] / max(buf[63], 1) * 0.1:.3f} GHz ({buf[62] / CALLS:.0f} shader cycles per call)")
for i, n in enumerate(names):
    print(f"  {n:36s} {buf[i] / tiles:8.0f} cycles  {100.0 * buf[i] / tot:5.1f} %")
