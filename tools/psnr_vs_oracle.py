"""PSNR after a training run from identical parameters, rays and jitter: CPU oracle (fp32) vs the HIP trainer
(fp32 / bf16 with fp32 or bf16 feature buffers), at a size that runs the SHIPPED scatter kernel (65 536 points per step).
env: STEPS="200,600"  SEEDS="7,8"  -> one table row per (seed, steps, configuration); HIP runs twice (reproducibility)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import numpy as np, torch, ref_cpu
from hbr_amd._lib import BF16, F32
from hbr_amd.helper import calc_psnr
from hbr_amd.trainer import HashNeRFTrainer, build_default_model
dev = "cuda:0"
torch.set_num_threads(16)
R, S, L, T = 1024, 64, 16, 2 ** 12
step_list = [int(v) for v in os.environ.get("STEPS", "200").split(",")]
seeds = [int(v) for v in os.environ.get("SEEDS", "7").split(",")]
o0, d0, _, _ = ref_cpu.synthetic_rays(8192, seed=0)
mn, mx, sig = ref_cpu.bbox_mu_sigma(o0, d0)
batches = [ref_cpu.synthetic_scene_rays(R, seed=50 + i) for i in range(16)]
test = ref_cpu.synthetic_scene_rays(2048, seed=999)
t_eval = torch.linspace(2.0, 6.0, S)
sc = ref_cpu.level_scales(16, 2048.0, L)
for seed in seeds:
    rng = np.random.default_rng(seed)
    tables0 = torch.from_numpy(rng.uniform(-1e-4, 1e-4, (L, T, 2)).astype(np.float32))
    params0 = ref_cpu.mlp_init(seed + 1)
    nmax = max(step_list)
    ts = [ref_cpu.strat_jitter_to_t(2.0, 6.0, S, torch.from_numpy(rng.uniform(0, 1, S).astype(np.float32))) for _ in range(nmax)]
    oracle = {}
    if not os.environ.get("SKIP_ORACLE"):
        tabs = [tables0[l].clone().requires_grad_(True) for l in range(L)]
        prm = {k: v.clone().requires_grad_(True) for k, v in params0.items()}
        opts = ref_cpu.make_optimizers(tabs, prm.values(), nmax)
        t0 = time.time()
        for k in range(nmax):
            ref_cpu.train_step(batches[k % 16], ts[k], tabs, sc, mn, sig, prm, opts)
            if k + 1 in step_list:
                with torch.no_grad():
                    C, _, _ = ref_cpu.render(test[0], test[1], t_eval, test[2], tabs, sc, mn, sig, prm)
                oracle[k + 1] = float(ref_cpu.psnr(C, test[3]))
                print(f"seed {seed} steps {k+1:4d}  oracle fp32                       PSNR {oracle[k+1]:.3f} dB  ({time.time()-t0:.0f}s)", flush=True)
    for name, prec, fdt in (("HIP fp32", F32, F32), ("HIP bf16 / fp32 feature buffers", BF16, F32), ("HIP bf16 / bf16 feature buffers", BF16, BF16)):
        for run in (1, 2):
            enc, denc, mlp = build_default_model(mn, sig, dev, L=L, T=T, seed=0)
            with torch.no_grad():
                for l in range(L): enc.Embedding_list[l].weight.copy_(tables0[l])
                for k, v in params0.items():
                    seq, idx, kind = k.split("."); getattr(getattr(mlp, seq)[int(idx)], kind).copy_(v)
            tr = HashNeRFTrainer(enc, mlp, num_samples=S, total_steps=nmax, precision=prec, feat_dtype=fdt)
            for k in range(nmax):
                tr.step(*(a.to(dev) for a in batches[k % 16]), t=ts[k].to(dev))
                if k + 1 in step_list:
                    C = tr.render(test[0].to(dev), test[1].to(dev), test[2].to(dev), t=t_eval.to(dev))
                    p = float(calc_psnr(C.cpu(), test[3]))
                    d = f"  delta vs oracle {p - oracle[k+1]:+.3f} dB" if k + 1 in oracle else ""
                    print(f"seed {seed} steps {k+1:4d}  {name:33s} run {run} PSNR {p:.3f} dB{d}", flush=True)
