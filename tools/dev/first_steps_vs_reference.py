"""First 16 training losses of the fused HIP trainer (PREC=fp32 | bf16) against the reference's recorded ones (g15 `loss_head`), seed by seed:
relative deviations.  Rounding-level agreement on step 1 says the step is the reference's; growth afterwards is the chaos of training."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import make_psnr_golden as MP, ref_cpu
from hbr_amd._lib import F32, BF16
from hbr_amd.trainer import HashNeRFTrainer, build_default_model
DEV = "cuda:0"
PREC = os.environ.get("PREC", "fp32")  # bf16: against the reference's own modules under bf16 autocast (g15 loss_head_bf16)
g = np.load(os.path.join(ROOT, "tests", "golden", "g15_converged_psnr.npz"))
steps = int(g["steps"])
mn, sig, batches, test = MP.scene()
batches = [tuple(a.to(DEV) for a in b) for b in batches]
for i, seed in enumerate(int(s) for s in g["seeds"]):
    tables0, u, params0 = MP.seeded_inputs(seed, steps)
    enc, denc, mlp = build_default_model(mn, sig, DEV, L=MP.L, T=MP.T, seed=0)
    with torch.no_grad():
        for l in range(MP.L): enc.Embedding_list[l].weight.copy_(torch.from_numpy(tables0[l]))
        for k, v in params0.items():
            seq, idx, kind = k.split("."); getattr(getattr(mlp, seq)[int(idx)], kind).copy_(v)
    tr = HashNeRFTrainer(enc, mlp, near=MP.NEAR, far=MP.FAR, num_samples=MP.S, total_steps=steps, precision=F32 if PREC == "fp32" else BF16)
    ls = []
    for k in range(16):
        t = ref_cpu.strat_jitter_to_t(MP.NEAR, MP.FAR, MP.S, torch.from_numpy(u[k])).to(DEV)
        ls.append(float(tr.step(*batches[k % MP.NB], t=t)))
    rel = np.array(ls) / (g["loss_head"] if PREC == "fp32" else g["loss_head_bf16"])[i] - 1
    print(f"{PREC} seed {seed}: step 1 {rel[0]:+.2e}  step 2 {rel[1]:+.2e}  step 4 {rel[3]:+.2e}  step 8 {rel[7]:+.2e}  step 16 {rel[15]:+.2e}   mean over 16 {rel.mean():+.2e}", flush=True)
