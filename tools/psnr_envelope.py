"""The reference's own run-to-run envelope (tests/golden/g15_converged_psnr.npz: per seed the unperturbed run + re-runs with
the initial tables moved by +1 / -1 / +2 ulps) against HIP runs from the same inputs (JSON files written by
tools/psnr_converged_study.py with PERTURBS=0,1,-1,2).  Prints the table DESIGN 4 / BASELINE.md quote.
usage: python tools/psnr_envelope.py study1.json [study2.json ...]"""
import json, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
g = np.load(os.path.join(ROOT, "tests", "golden", "g15_converged_psnr.npz"))
seeds = [int(s) for s in g["seeds"]]
ulps = [0] + [int(u) for u in g["self_ulps"]]
ref = np.concatenate([g["psnr"][:, None, -1], g["psnr_self"][:, :, -1]], axis=1)  # [seed, perturbation]


def stats(x):  # x [seed, run]
    within = np.sqrt(np.mean(np.var(x, axis=1, ddof=1)))
    return x.mean(axis=1), within


rm, rw = stats(ref)
print("reference (its own modules, fp32 CPU), final held-out PSNR per seed over the perturbations", ulps)
for i, s in enumerate(seeds):
    print(f"  seed {s}: " + " ".join(f"{v:6.2f}" for v in ref[i]) + f"   mean {rm[i]:.2f}  range {ref[i].max() - ref[i].min():.2f}")
self_delta = ref[:, 1:] - ref[:, :1]
print(f"  pooled within-seed sd {rw:.3f} dB; self-deltas (perturbed - unperturbed, n = {self_delta.size}): mean {self_delta.mean():+.3f}  sd {self_delta.std(ddof=1):.3f}"
      f"  min {self_delta.min():+.2f}  max {self_delta.max():+.2f}  SE {self_delta.std(ddof=1) / np.sqrt(self_delta.size):.3f}")
if "degenerate_seeds" in g.files:
    for s, p in zip(g["degenerate_seeds"], g["psnr_degenerate"]):
        print(f"  seed {int(s)}: DEGENERATE - the reference itself stays at {p[-1]:.2f} dB for all 2000 steps (excluded)")
cfgs = {}
for f in sys.argv[1:]:
    for cfg, v in json.load(open(f))["final_psnr_curves"].items():
        cfgs.setdefault(cfg, {}).update(v)
print("\nHIP - reference, per-seed means over the same perturbations (D = mean over seeds; SE_seeds = sd of the per-seed deltas / sqrt(n);")
print("z_noise = D / sqrt((sd_ref^2 + sd_hip^2) / runs): the same difference against within-seed noise alone)")
print(f"{'configuration':24s} " + " ".join(f"seed {s:<2d}" for s in seeds) + "      D     SE_seeds  z_noise  within-seed sd (ratio to ref)  vs self-delta interval")
for cfg, v in cfgs.items():
    try:
        hip = np.array([[v[str(s)][str(u)][-1] for u in ulps] for s in seeds])
    except KeyError:
        continue
    hm, hw = stats(hip)
    d = hm - rm
    D, se = d.mean(), d.std(ddof=1) / np.sqrt(len(d))
    zn = D / np.sqrt((rw ** 2 + hw ** 2) / hip.size)
    hd = hip - ref[:, :1]
    se_j = np.sqrt(self_delta.var(ddof=1) / self_delta.size + hd.var(ddof=1) / hd.size)
    print(f"{cfg:24s} " + " ".join(f"{x:+7.2f}" for x in d) + f"  {D:+6.2f}   {se:5.2f}    {zn:+5.1f}    {hw:.3f} ({hw / rw:.2f}x)"
          f"                 mean delta vs unperturbed {hd.mean():+.2f}; self {self_delta.mean():+.2f} +- {2 * se_j:.2f}")
    for ds in ([int(x) for x in g["degenerate_seeds"]] if "degenerate_seeds" in g.files else []):
        if str(ds) in v:
            print(f"{'':24s} seed {ds} (degenerate): HIP " + " ".join(f"{c[-1]:.2f}" for c in v[str(ds)].values()) + " dB")
    print(f"{'':24s} per-seed deltas: sd {d.std(ddof=1):.2f} dB; expected from within-seed noise alone {np.sqrt((rw ** 2 + hw ** 2) / hip.shape[1]):.2f}")

if "psnr_bf16" in g.files:
    refb = g["psnr_bf16"][:, :, -1]                     # [seed, run]: the reference's own modules under torch.autocast(cpu, bfloat16)
    rb = refb.mean(axis=1)
    print("\nbf16 against bf16: the reference's OWN modules with forward + loss under torch.autocast(cpu, bfloat16) (oracle/make_psnr_golden.py")
    print("--autocast-bf16; its loop runs under autocast, train_hash2.py:218), runs at " + str([int(u) for u in g["bf16_ulps"]]) + " ulps of the initial tables")
    for i, s_ in enumerate(seeds):
        print(f"  seed {s_}: " + " ".join(f"{v:6.2f}" for v in refb[i]) + f"   mean {rb[i]:.2f}   shift against the fp32 seed mean {rb[i] - rm[i]:+.2f}")
    sh = rb - rm
    wb = np.sqrt(np.mean(np.var(refb, axis=1, ddof=1))) if refb.shape[1] > 1 else float("nan")
    print(f"  reference bf16 - reference fp32 (seed means): mean {sh.mean():+.3f}  sd over the seeds {sh.std(ddof=1):.3f}  SE {sh.std(ddof=1) / np.sqrt(len(sh)):.3f};"
          f" within-seed sd of the bf16 reference runs {wb:.3f}")
    print(f"{'configuration':24s} " + " ".join(f"seed {s_:<2d}" for s_ in seeds) + "      D     SE_seeds   sd    corr(HIP shift, reference-bf16 shift)   [HIP seed mean - reference bf16 seed mean]")
    for cfg, v in cfgs.items():
        try:
            hip = np.array([[v[str(s_)][str(u)][-1] for u in ulps] for s_ in seeds])
        except KeyError:
            continue
        hm = hip.mean(axis=1)
        d = hm - rb
        print(f"{cfg:24s} " + " ".join(f"{x:+7.2f}" for x in d) + f"  {d.mean():+6.2f}   {d.std(ddof=1) / np.sqrt(len(d)):5.2f}    {d.std(ddof=1):.2f}    {np.corrcoef(hm - rm, sh)[0, 1]:+.2f}")
