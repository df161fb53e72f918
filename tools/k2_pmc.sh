#!/usr/bin/env bash
# SQ / LDS counters of the K2 kernels ON THE GPU BOX (separate rocprofv3 --pmc passes, kernel-trace only).
# usage: tools/k2_pmc.sh <tag>
set -uo pipefail
TAG="${1:?tag}"; ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -o run -- python3 "$ROOT/tools/k2_only.py" bwd > "$OUT/p$i.log" 2>&1 || echo "set $i failed: $set"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "hbr::" not in k: continue
        name = k.split("hbr::")[1].split("(")[0][:40]
        acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(sys.argv[1] + "/counters.csv", "w") as out:
    out.write("kernel,counter,avg_per_launch,launches\n")
    for (k, c), v in sorted(acc.items()):
        out.write(f"{k},{c},{sum(v)/len(v):.4g},{len(v)}\n")
print(open(sys.argv[1] + "/counters.csv").read())
PY
