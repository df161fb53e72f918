"""Turn gpurun_out/<tag>/ (written by tools/profile_round.sh) into the files committed under profiles/:
<tag>_kernel_stats.csv (rocprofv3 --stats summary, our kernels + the largest others), <tag>_pmc_hbm.csv,
<tag>_bench.json, and profiles/pmc_traffic.json (read by bench.py for roofline.traffic).
HBM bytes per launch = (FETCH_SIZE*2 + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, gfx950 correction)."""
import csv, glob, json, os, sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
# bench.py's four timed spans -> the kernels each one launches (K2 and K4 are short sequences of kernels; a span's HBM
# traffic is the sum over its kernels, one launch of each per step)
KERNELS = {"hash_fwd": ["hash_fwd_kernel"],
           "hash_bwd": ["normalise_kernel", "absmax_", "meta_reduce_kernel", "hash_scatter_kernel", "dense_scatter_kernel",
                        "slab_reduce_kernel"],
           "mlp_fwd": ["mlp_fwd_kernel"], "mlp_bwd": ["mlp_bwd_fused_kernel", "mlp_dw_reduce_kernel"]}

stats = glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.reader(open(stats)))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(rows[0])
    for r in rows[1:]:
        if len(r[0]) < 400:  # drop torch's page-long template names (random fill etc.), all < 0.3 % of the time
            w.writerow(r)

def counter(name):
    """span -> kernel pattern -> list of per-launch counter values"""
    f = glob.glob(os.path.join(src, name, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        for k, pats in KERNELS.items():
            for pat in pats:
                if pat in r["Kernel_Name"]:
                    acc[k][pat].append(float(r["Counter_Value"]))
    return acc

fetch, write = counter("fetch"), counter("write")
traffic = {}
with open(os.path.join(dst, f"{tag}_pmc_hbm.csv"), "w") as f:
    f.write("span,kernel,launches,FETCH_SIZE_KiB_avg,WRITE_SIZE_KiB_avg,hbm_bytes_per_launch=(FETCHx2+WRITE)*1024\n")
    for k, pats in KERNELS.items():
        tot = 0.0
        for pat in pats:
            if not fetch[k][pat]:
                continue
            fa, wa = sum(fetch[k][pat]) / len(fetch[k][pat]), sum(write[k][pat]) / len(write[k][pat])
            tot += (2 * fa + wa) * 1024
            f.write(f"{k},{pat},{len(fetch[k][pat])},{fa:.1f},{wa:.1f},{(2 * fa + wa) * 1024:.0f}\n")
        traffic[k] = tot
        f.write(f"{k},TOTAL,,,,{tot:.0f}\n")
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
line = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1]
# the bench line was printed before this round's PMC passes existed: its `traffic` fields (read from the previous
# pmc_traffic.json) are refreshed with the numbers just collected, everything else is kept as printed
rec = json.loads(line)
for key in ("roofline", "roofline_hash_lookup"):
    if rec.get(key) and rec[key].get("kernel") in traffic:
        rec[key]["traffic"] = traffic[rec[key]["kernel"]]
for k, r in (rec.get("kernels") or {}).items():
    if k in traffic:
        r["traffic"] = traffic[k]
open(os.path.join(dst, f"{tag}_bench.json"), "w").write(json.dumps(rec) + "\n")
print(open(os.path.join(dst, f"{tag}_pmc_hbm.csv")).read())
