#!/usr/bin/env bash
# ON THE GPU BOX: kernel traces of the one-GPU RCCL rehearsal (bench.py HBR_RCCL_REHEARSAL=1) - staged (split scatter +
# three collectives) and unstaged (one collective) - next to the plain single-GPU step.  -> gpurun_out/<tag>/{plain,staged,single}
set -uo pipefail
TAG="${1:?tag}"; ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
Q="--no-cpu-baseline --no-dropin --steps 40 --warmup 10"
unset HBR_RCCL_REHEARSAL HBR_OVERLAP_COMM
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/plain" -o run -- python3 "$ROOT/bench.py" $Q > "$OUT/plain.log" 2>&1
export HBR_RCCL_REHEARSAL=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/staged" -o run -- python3 "$ROOT/bench.py" $Q > "$OUT/staged.log" 2>&1
export HBR_OVERLAP_COMM=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/single" -o run -- python3 "$ROOT/bench.py" $Q > "$OUT/single.log" 2>&1
for m in plain staged single; do
  echo "== $m"; grep -h '"ms_per_step"' "$OUT/$m.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], 'exposed', d['allreduce_exposed_ms'])"
  python3 - "$OUT/$m/run_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))[1:]
for r in rows[:14]:
    print(f"   {r[0][:80]:80s} calls {r[1]:>5s} avg_us {float(r[3])/1e3:9.1f} total_ms {float(r[2])/1e6:8.2f}")
PY
done
