#!/usr/bin/env bash
# SQ / LDS / L1 counters of every kernel of the SHIPPED train step (bench.py's configuration: bf16 MLP, bf16 feature
# buffers) ON THE GPU BOX: separate rocprofv3 --pmc passes (kernel-trace only).  usage: tools/sq_counters.sh <tag>
set -uo pipefail
TAG="${1:?tag}"; ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -o run -- python3 "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline > "$OUT/p$i.log" 2>&1 || echo "set $i failed: $set"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "hbr::" not in k: continue
        name = k.split("hbr::")[1].split("(")[0][:48]
        acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(sys.argv[1] + "/counters.csv", "w") as out:
    out.write("kernel,counter,avg_per_launch,launches\n")
    for (k, c), v in sorted(acc.items()):
        out.write(f"{k},{c},{sum(v)/len(v):.5g},{len(v)}\n")
print(sum(1 for _ in open(sys.argv[1] + "/counters.csv")), "rows")
PY
