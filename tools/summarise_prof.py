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
KERNELS = {"hash_fwd": "hash_fwd_kernel", "hash_bwd": "hash_bwd_lds_kernel", "mlp_fwd": "mlp_fwd_kernel", "mlp_bwd": "mlp_bwd_fused_kernel"}

stats = glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.reader(open(stats)))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(rows[0])
    for r in rows[1:]:
        if len(r[0]) < 400:  # drop torch's page-long template names (random fill etc.), all < 0.3 % of the time
            w.writerow(r)

def counter(name):
    f = glob.glob(os.path.join(src, name, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        for k, pat in KERNELS.items():
            if pat in r["Kernel_Name"]:
                acc[k].append(float(r["Counter_Value"]))
    return acc

fetch, write = counter("fetch"), counter("write")
traffic = {}
with open(os.path.join(dst, f"{tag}_pmc_hbm.csv"), "w") as f:
    f.write("kernel,launches,FETCH_SIZE_KiB_avg,WRITE_SIZE_KiB_avg,hbm_bytes_per_launch=(FETCHx2+WRITE)*1024\n")
    for k in KERNELS:
        fa, wa = sum(fetch[k]) / len(fetch[k]), sum(write[k]) / len(write[k])
        traffic[k] = (2 * fa + wa) * 1024
        f.write(f"{k},{len(fetch[k])},{fa:.1f},{wa:.1f},{traffic[k]:.0f}\n")
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
