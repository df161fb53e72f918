"""Held-out PSNR at the horizon of tests/golden/g15_converged_psnr.npz (the reference's own modules, trained to a plateau):
every HIP route / precision from the same seeded inputs.  Prints one row per (seed, configuration)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import make_psnr_golden as MP, ref_cpu
from hbr_amd._lib import BF16, F32
from hbr_amd.helper import calc_psnr
from hbr_amd.trainer import HashNeRFTrainer, build_default_model
from hbr_amd.vol_renderer import Volume_Renderer
DEV = "cuda:0"
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g15_converged_psnr.npz")))
steps, ev = int(g["steps"]), [int(v) for v in g["eval_steps"]]
mn, sig, batches, test = MP.scene()
batches = [tuple(a.to(DEV) for a in b) for b in batches]
test = tuple(a.to(DEV) for a in test)
t_eval = torch.linspace(MP.NEAR, MP.FAR, MP.S, device=DEV)
seeds = [int(s) for s in os.environ.get("SEEDS", ",".join(str(int(s)) for s in g["seeds"])).split(",")]
configs = os.environ.get("CONFIGS", "fused-bf16,fused-bf16-f32feat,fused-fp32,dropin-bf16,dropin-fp32").split(",")
# PERTURBS: the initial tables moved by that many fp32 ulps (the reference's own self-noise runs use the same moves);
# a config ending in "-algo1" scatters with global float atomics (no fixed-point quantum) instead of the LDS kernels
perturbs = [int(v) for v in os.environ.get("PERTURBS", "0").split(",")]
OUT_JSON = os.environ.get("OUT_JSON", "")


def model(tables0, params0):
    enc, denc, mlp = build_default_model(mn, sig, DEV, L=MP.L, T=MP.T, seed=0)
    with torch.no_grad():
        for l in range(MP.L): enc.Embedding_list[l].weight.copy_(torch.from_numpy(tables0[l]))
        for k, v in params0.items():
            seq, idx, kind = k.split("."); getattr(getattr(mlp, seq)[int(idx)], kind).copy_(v)
    return enc, denc, mlp


def run(seed, cfg, ulps=0):
    tables0, u, params0 = MP.seeded_inputs(seed, steps)
    for _ in range(abs(ulps)):
        tables0 = np.nextafter(tables0, np.float32(np.inf if ulps > 0 else -np.inf))
    ts = torch.stack([ref_cpu.strat_jitter_to_t(MP.NEAR, MP.FAR, MP.S, torch.from_numpy(u[k])) for k in range(steps)]).to(DEV)
    enc, denc, mlp = model(tables0, params0)
    curve = []
    if cfg.startswith("fused"):
        prec = F32 if "fp32" in cfg else BF16
        tr = HashNeRFTrainer(enc, mlp, near=MP.NEAR, far=MP.FAR, num_samples=MP.S, total_steps=steps, precision=prec,
                             feat_dtype=F32 if "f32feat" in cfg else None, scatter_algo=1 if cfg.endswith("-algo1") else 0)
        # "-rnoiseK": every gradient entry multiplied by (1 + K * 2^-24 * U(-1, 1)) in front of the optimiser - what K fp32
        # roundings per accumulated sum would do (the HIP sums are exact integers / fixed-order fp32; the reference's CPU
        # index_add and BLAS sums round at every step); "-anoiseK": K * 1e-10 * max|g| of additive Gaussian noise
        import re as _re
        m = _re.search(r"-(r|a)noise(\d+)", cfg)
        if m:
            kind, K = m.group(1), float(m.group(2))
            gen = torch.Generator(device=DEV).manual_seed(1234 + seed)
            def hook(g, kind=kind, K=K, gen=gen):
                if kind == "r":
                    g.mul_(1.0 + (torch.rand(g.shape, device=g.device, generator=gen) * 2 - 1) * (K * 2.0 ** -24))
                else:
                    g.add_(torch.randn(g.shape, device=g.device, generator=gen) * (K * 1e-10 * float(g.abs().max())))
            tr.grad_hook = hook
        for k in range(steps):
            tr.step(*batches[k % MP.NB], t=ts[k])
            if k + 1 in ev:
                curve.append(float(calc_psnr(tr.render(test[0], test[1], test[2], t=t_eval), test[3])))
    else:
        ac = "bf16" in cfg
        nerf = torch.nn.DataParallel(mlp, device_ids=[0])
        vr = Volume_Renderer(H=8, W=8, K=torch.eye(3), near=MP.NEAR, far=MP.FAR, device=DEV, Pos_encode=enc, Dir_encode=denc,
                             max_dim=2 ** 10, sigma_val=sig.to(DEV), mu=mn.to(DEV))
        oe = torch.optim.Adam(enc.Embedding_list.parameters(), lr=0.05); om = torch.optim.AdamW(nerf.parameters(), lr=0.005)
        se = torch.optim.lr_scheduler.CosineAnnealingLR(oe, T_max=steps, eta_min=1e-4); sm = torch.optim.lr_scheduler.CosineAnnealingLR(om, T_max=steps, eta_min=1e-4)
        crit = torch.nn.MSELoss()
        for k in range(steps):
            o, d, dn, gt = batches[k % MP.NB]
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac):
                Cr, Cf, _ = vr.vol_render(nerf, d, o, num_samples=MP.S, t=ts[k], update_mask=False, dir_norm=dn, hierarchical=False)
                loss = crit(Cr, gt) + crit(Cf, gt)
            loss.backward(); oe.step(); om.step(); se.step(); sm.step(); om.zero_grad(set_to_none=True); oe.zero_grad(set_to_none=True)
            if k + 1 in ev:
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac):
                    C = vr.vol_render(nerf, test[1], test[0], num_samples=MP.S, t=t_eval, update_mask=False, dir_norm=test[2], hierarchical=False)[0]
                curve.append(float(calc_psnr(C, test[3])))
    return np.array(curve)


import json
results = {}
for cfg in configs:
    for ulps in perturbs:
        ds = []
        gseeds = [int(s) for s in g["seeds"]]
        for seed in seeds:
            c = run(seed, cfg, ulps); ref = g["psnr"][gseeds.index(seed)] if seed in gseeds else np.full(len(ev), np.nan)
            q = len(ev) // 4
            ds.append(c[-1] - ref[-1])
            results.setdefault(cfg, {}).setdefault(str(seed), {})[str(ulps)] = [float(v) for v in c]
            print(f"{cfg:22s} ulps {ulps:+d} seed {seed}: ref {ref[-1]:.3f}  hip {c[-1]:.3f}  delta {ds[-1]:+.3f} | step {ev[q]}: {c[q]-ref[q]:+.3f}  step {ev[2*q]}: {c[2*q]-ref[2*q]:+.3f}  step {ev[3*q]}: {c[3*q]-ref[3*q]:+.3f} | tail span {c[int(len(c)*.8):].max()-c[int(len(c)*.8):].min():.3f}", flush=True)
        ds = np.array(ds)
        print(f"{cfg:22s} ulps {ulps:+d} mean delta {ds.mean():+.3f}  std {ds.std():.3f}  max|d| {np.abs(ds).max():.3f}", flush=True)
        if OUT_JSON:
            with open(OUT_JSON, "w") as f:
                json.dump({"eval_steps": ev, "final_psnr_curves": results}, f)
