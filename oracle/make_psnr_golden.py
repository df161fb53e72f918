"""Generate tests/golden/g15_converged_psnr.npz: the REFERENCE's own modules trained to a plateau, per seed.

TEST INFRASTRUCTURE ONLY; runs in the build container (needs /root/reference, which never travels to the GPU box).
The committed .npz is data: seeds, checksums of the seeded inputs, and the reference's PSNR-vs-step curves.

What is run (the loop of /root/reference/train_hash2.py:211-234 on the reference's library modules, fp32 on CPU):
    Cr, Cf, _ = Volume_Renderer.vol_render(DataParallel(MLP_3D), ray_d, ray_o, num_samples=S, t=t,
                                           update_mask=False, dir_norm=dn, hierarchical=False)   # :220
    loss = MSE(Cr, gt) + MSE(Cf, gt)                                                             # :177,221
    loss.backward(); Adam(lr .05).step(); AdamW(lr .005).step()                                  # :141-142,226-228
    CosineAnnealingLR(T_max=total steps, eta_min=1e-4).step() x2; zero_grad(set_to_none=True)    # :156-162,231-234
(GradScaler/autocast are no-ops for an fp32 CPU run.)  The jitter `t` is passed explicitly so that the GPU test can
replay it; everything else is the reference's code.  Held-out PSNR = helper.calc_psnr (helper.py:301-304) of
vol_render on 2048 other rays at the un-jittered depths linspace(near, far, S).

Workload (small enough for the CPU, large enough that the GPU test runs the SHIPPED scatter kernel: 65 536 points/step):
16 fixed batches of 1024 rays x 64 samples of the analytic scene (ref_cpu.synthetic_scene_rays, seeds 50..65), cycled;
L=16, F=2, T=2^12, N_min=16, N_max=2048.0; tables U(-1e-4,1e-4) and nn.Linear-style MLP weights from numpy PCG64(seed).
The cosine schedule ends at the horizon, so the run anneals into a plateau.

Usage (one process per seed, ~50 min each on one thread, then merge):
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_psnr_golden.py --seed 1 --steps 2000 --out /tmp/psnr/s1.npz
    ... --seed 1 --perturb-ulps 1 --save-final /tmp/psnr_p/final_s1.npz --out /tmp/psnr_p/s1.npz      (sensitivity re-run)
    ... --seed 1 --perturb-ulps 2 / --perturb-ulps -1 ...                                          (round 4: two more per seed)
    python oracle/make_psnr_golden.py --merge /tmp/psnr/s*.npz --merge-ulp /tmp/psnr_p/s?.npz /tmp/psnr4/s?_u2.npz /tmp/psnr4/s?_u-1.npz \
                                      --merge-degenerate /tmp/psnr4/s5_u0.npz                      -> tests/golden/g15_converged_psnr.npz
    ... --seed 1 --autocast-bf16 [--perturb-ulps 1] ...; --merge ... --merge-autocast /tmp/psnr6/s*_ac*.npz   (round 4, late: the reference under bf16)
    cp /tmp/psnr_p/final_s1.npz tests/golden/g15b_trained_weights.npz    (the reference's trained weights + its held-out render)
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "g15_converged_psnr.npz")

R, S, L, T, NB = 1024, 64, 16, 2 ** 12, 16
NEAR, FAR = 2.0, 6.0
EVAL_RAYS, EVAL_SEED, BATCH_SEED0, BBOX_SEED = 2048, 999, 50, 0


def seeded_inputs(seed: int, steps: int):
    """Everything the run draws from numpy PCG64(seed), in this order: tables, then the per-step jitter; the MLP
    weights come from ref_cpu.mlp_init(seed + 1).  The GPU test calls this same function."""
    import ref_cpu
    rng = np.random.default_rng(seed)
    tables0 = rng.uniform(-1e-4, 1e-4, (L, T, 2)).astype(np.float32)
    u = rng.uniform(0, 1, (steps, S)).astype(np.float32)
    params0 = ref_cpu.mlp_init(seed + 1)
    return tables0, u, params0


def checksum(*arrays) -> float:
    return float(sum(np.abs(np.asarray(a, dtype=np.float64)).sum() for a in arrays))


def scene():
    import ref_cpu
    o0, d0, _, _ = ref_cpu.synthetic_rays(8192, seed=BBOX_SEED)
    mn, mx, sig = ref_cpu.bbox_mu_sigma(o0, d0)
    batches = [ref_cpu.synthetic_scene_rays(R, seed=BATCH_SEED0 + i) for i in range(NB)]
    test = ref_cpu.synthetic_scene_rays(EVAL_RAYS, seed=EVAL_SEED)
    return mn, sig, batches, test


def run_seed(ref_dir: str, seed: int, steps: int, eval_every: int, out: str, threads: int, perturb_ulps: int = 0, save_final: str = "",
             autocast_bf16: bool = False):
    import make_golden as MG
    import ref_cpu
    torch.set_num_threads(threads)
    ref = MG.import_reference(ref_dir)
    mn, sig, batches, test = scene()
    tables0, u, params0 = seeded_inputs(seed, steps)
    for _ in range(abs(perturb_ulps)):  # sensitivity study: every initial table entry moved to the next fp32 value (1 ulp ~ 1e-11); a negative count moves them down
        tables0 = np.nextafter(tables0, np.float32(np.inf if perturb_ulps > 0 else -np.inf))
    enc = MG.build_encoder(ref, tables0, 2048.0, 16, mn.numpy(), float(sig))
    denc = ref.encoder.PositionalEncoder(3, 4)
    nerf = torch.nn.DataParallel(MG.build_mlp(ref, params0))          # train_hash2.py:127 (no GPU: calls the module)
    vr = MG.quiet(ref.vol_renderer.Volume_Renderer, H=8, W=8, K=torch.eye(3), near=NEAR, far=FAR, device="cpu",
                  Pos_encode=enc, Dir_encode=denc, max_dim=2 ** 10, sigma_val=sig, mu=mn)
    opt_e = torch.optim.Adam(enc.Embedding_list.parameters(), lr=0.05)                     # train_hash2.py:141
    opt_m = torch.optim.AdamW(nerf.parameters(), lr=0.005)                                 # :142
    sch_e = torch.optim.lr_scheduler.CosineAnnealingLR(opt_e, T_max=steps, eta_min=1e-4)   # :156-159
    sch_m = torch.optim.lr_scheduler.CosineAnnealingLR(opt_m, T_max=steps, eta_min=1e-4)   # :160-162
    crit = torch.nn.MSELoss()                                                              # :177
    t_eval = torch.linspace(NEAR, FAR, S)
    ev_steps, ev_psnr, losses = [], [], np.zeros(steps, dtype=np.float32)
    t0 = time.time()
    for k in range(steps):
        o, d, dn, gt = batches[k % NB]
        t = ref_cpu.strat_jitter_to_t(NEAR, FAR, S, torch.from_numpy(u[k]))
        # --autocast-bf16: the reference's loop runs its forward + loss under autocast (train_hash2.py:218-221; fp16 on its CUDA
        # device); the CPU equivalent is bf16 - the reference's OWN modules at reduced precision, for the bf16 rows of the study
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast_bf16):
            Cr, Cf, _ = MG.quiet(vr.vol_render, nerf, d, o, num_samples=S, t=t, update_mask=False, dir_norm=dn, hierarchical=False)
            loss = crit(Cr, gt) + crit(Cf, gt)
        loss.backward()
        opt_e.step(); opt_m.step()
        sch_e.step(); sch_m.step()
        opt_m.zero_grad(set_to_none=True); opt_e.zero_grad(set_to_none=True)
        losses[k] = float(loss)
        if (k + 1) % eval_every == 0 or k + 1 == steps:
            with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast_bf16):
                C, _, _ = MG.quiet(vr.vol_render, nerf, test[1], test[0], num_samples=S, t=t_eval, update_mask=False,
                                   dir_norm=test[2], hierarchical=False)
                p = float(ref.helper.calc_psnr(C.float(), test[3]))
            ev_steps.append(k + 1); ev_psnr.append(p)
            print(f"seed {seed} step {k + 1:5d} loss {losses[k]:.5f} held-out PSNR {p:.3f} dB ({time.time() - t0:.0f}s)", flush=True)
            np.savez(out, seed=seed, steps=steps, eval_steps=np.array(ev_steps), psnr=np.array(ev_psnr, dtype=np.float64),
                     loss=losses[:k + 1], input_checksum=checksum(tables0, u, *[v.numpy() for v in params0.values()]),
                     scene_checksum=checksum(*[a.numpy() for b in batches[:2] for a in b], *[a.numpy() for a in test]),
                     perturb_ulps=perturb_ulps, autocast_bf16=int(autocast_bf16))
    if save_final:  # the trained parameters + the reference's own render of the held-out rays with them
        with torch.no_grad():
            C, _, _ = MG.quiet(vr.vol_render, nerf, test[1], test[0], num_samples=S, t=t_eval, update_mask=False, dir_norm=test[2], hierarchical=False)
        fin = {"tables": np.stack([enc.Embedding_list[l].weight.detach().numpy() for l in range(L)]), "Cr_eval": C.numpy(),
               "psnr": float(ref.helper.calc_psnr(C, test[3])), "seed": seed, "steps": steps, "perturb_ulps": perturb_ulps}
        for name, p_ in nerf.module.named_parameters():
            fin["p." + name] = p_.detach().numpy()
        np.savez_compressed(save_final, **fin)


def merge(files, ulp_files=(), degenerate_files=(), autocast_files=()):
    """files: the unperturbed runs (one per seed); ulp_files: re-runs of the same seeds with every initial table entry moved
    by +1 / +2 / -1 ... fp32 ulps (--perturb-ulps: the reference's OWN trajectory noise - same inputs up to 1e-11, same
    code); stored as psnr_self [seed, perturbation, eval] with self_ulps naming the perturbations (psnr_ulp = the +1 row, as
    round 3 stored it).  degenerate_files: runs of seeds from which the reference itself does not train (kept as evidence,
    excluded from the statistics: degenerate_seeds / psnr_degenerate).  autocast_files: runs of the same seeds with the reference's
    forward + loss under torch.autocast("cpu", bfloat16) (--autocast-bf16; train_hash2.py:218 runs under autocast on its GPU) -
    the reference's OWN modules at the precision of the shipped bf16 MLP: psnr_bf16 [seed, run, eval], bf16_ulps = their
    --perturb-ulps."""
    runs = sorted((np.load(f) for f in files), key=lambda z: int(z["seed"]))
    seeds = [int(z["seed"]) for z in runs]
    extra = {}
    by = {}
    for f in ulp_files:
        z = np.load(f)
        assert int(z["eval_steps"][-1]) == int(z["steps"]), f"{f}: a perturbed run did not finish"
        assert int(z["perturb_ulps"]) != 0 and int(z["seed"]) in seeds, f
        by.setdefault(int(z["perturb_ulps"]), {})[int(z["seed"])] = z["psnr"]
    if by:
        order = sorted(by, key=lambda u: (abs(u), -u))  # +1, -1, +2, -2, ...
        for u in order:
            assert sorted(by[u]) == seeds, f"perturbation {u:+d} ulps: runs for seeds {sorted(by[u])}, expected {seeds}"
        extra["self_ulps"] = np.array(order)
        extra["psnr_self"] = np.stack([np.stack([by[u][s] for u in order]) for s in seeds])
        if 1 in by:
            extra["psnr_ulp"] = np.stack([by[1][s] for s in seeds])
    if degenerate_files:
        dz = sorted((np.load(f) for f in degenerate_files), key=lambda z: int(z["seed"]))
        extra["degenerate_seeds"] = np.array([int(z["seed"]) for z in dz])
        extra["psnr_degenerate"] = np.stack([z["psnr"] for z in dz])
        extra["degenerate_input_checksum"] = np.array([float(z["input_checksum"]) for z in dz])
    if autocast_files:
        bya, bya_loss = {}, {}
        for f in autocast_files:
            z = np.load(f)
            assert int(z["autocast_bf16"]) == 1 and int(z["eval_steps"][-1]) == int(z["steps"]) and int(z["seed"]) in seeds, f
            bya.setdefault(int(z["perturb_ulps"]), {})[int(z["seed"])] = z["psnr"]
            bya_loss.setdefault(int(z["perturb_ulps"]), {})[int(z["seed"])] = z["loss"]
        order = sorted(bya, key=lambda u: (abs(u), -u))  # 0, +1, ...
        for u in order:
            assert sorted(bya[u]) == seeds, f"autocast runs at {u:+d} ulps: seeds {sorted(bya[u])}, expected {seeds}"
        extra["bf16_ulps"] = np.array(order)
        extra["psnr_bf16"] = np.stack([np.stack([bya[u][s] for u in order]) for s in seeds])
        if 0 in bya_loss:
            extra["loss_head_bf16"] = np.stack([bya_loss[0][s][:16] for s in seeds])  # first 16 losses of the unperturbed bf16 runs
    steps = {int(z["steps"]) for z in runs}
    assert len(steps) == 1, "all seeds must share the horizon"
    ev = runs[0]["eval_steps"]
    for z in runs:
        assert int(z["eval_steps"][-1]) == int(z["steps"]), f"seed {int(z['seed'])} did not finish"
    np.savez(OUT, seeds=np.array(seeds), steps=steps.pop(), eval_steps=ev,
             psnr=np.stack([z["psnr"] for z in runs]), loss_head=np.stack([z["loss"][:16] for z in runs]),
             loss_tail=np.stack([z["loss"][-64:] for z in runs]),
             input_checksum=np.array([float(z["input_checksum"]) for z in runs]),
             scene_checksum=float(runs[0]["scene_checksum"]),
             config=np.array([R, S, L, T, NB, EVAL_RAYS, EVAL_SEED, BATCH_SEED0, BBOX_SEED]), **extra)
    z = np.load(OUT)
    for i, (s, p) in enumerate(zip(z["seeds"], z["psnr"])):
        n = len(p)
        tail = p[int(n * 0.8):]
        more = ""
        if "psnr_self" in z.files:
            if "psnr_bf16" in z.files:
                more = "; under bf16 autocast " + " ".join(f"{v:.3f}" for v in z["psnr_bf16"][i][:, -1])
            more += "; re-runs with the tables moved by " + ", ".join(f"{int(u):+d} ulp: {z['psnr_self'][i][k][-1]:.3f} ({z['psnr_self'][i][k][-1] - p[-1]:+.3f})"
                                                                      for k, u in enumerate(z["self_ulps"]))
        print(f"seed {s}: final {p[-1]:.3f} dB; last 20% of the horizon spans {tail.max() - tail.min():.3f} dB{more}")
    if "degenerate_seeds" in z.files:
        for s, p in zip(z["degenerate_seeds"], z["psnr_degenerate"]):
            print(f"seed {s} (degenerate - excluded): the reference's held-out PSNR stays at {p[-1]:.3f} dB (max over the run {p.max():.3f})")
    print("wrote", OUT)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--eval-every", type=int, default=50)
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--out", default="/tmp/psnr_seed.npz")
    ap.add_argument("--perturb-ulps", type=int, default=0)
    ap.add_argument("--save-final", default="")
    ap.add_argument("--autocast-bf16", action="store_true", help="run the reference's forward + loss under torch.autocast(cpu, bfloat16)")
    ap.add_argument("--merge", nargs="+")
    ap.add_argument("--merge-ulp", nargs="*", default=[], help="the perturbed re-runs (any --perturb-ulps), all seeds of --merge")
    ap.add_argument("--merge-degenerate", nargs="*", default=[], help="runs of seeds the reference itself does not train from")
    ap.add_argument("--merge-autocast", nargs="*", default=[], help="--autocast-bf16 runs of the --merge seeds (unperturbed and perturbed)")
    a = ap.parse_args()
    if a.merge:
        merge(a.merge, a.merge_ulp, a.merge_degenerate, a.merge_autocast)
    else:
        run_seed(a.ref, a.seed, a.steps, a.eval_every, a.out, a.threads, a.perturb_ulps, a.save_final, a.autocast_bf16)
