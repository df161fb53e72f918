#!/usr/bin/env bash
# K4 development loop ON THE GPU BOX: the MLP-related GPU tests on the default build, then tools/k4_time.py under a
# rocprofv3 kernel trace for the default build and for every variant library given.
# usage: tools/k4_dev.sh <tag> [variant.so ...]      (variants: tools/variant.sh <name> mlp -D... on the CPU side, e.g. the
# phase-timer builds -DHBR_K4_PROF=1 / =2 read by tools/k4_phases.py)
set -uo pipefail
TAG="${1:?tag}"; shift; ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python3 -m pytest "$ROOT/tests" -m gpu -x -q -k "mlp or k4 or trainer or training or smoke or golden or parity" > "$OUT/pytest_k4.log" 2>&1; tail -3 "$OUT/pytest_k4.log"
cd /tmp && export TMPDIR=/tmp
i=0
for lib in default "$@"; do
  [ "$lib" = default ] && unset HBR_LIB || export HBR_LIB="$ROOT/$lib"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt$i" -o run -- python3 "$ROOT/tools/k4_time.py" > "$OUT/k4time$i.log" 2>&1
  grep mlp_bwd "$OUT/k4time$i.log"
  python3 - "$OUT/kt$i/run_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:8]:
    if "mlp" in r[0]: print(f"    {r[0][:70]:70s} calls {r[1]:>4s} avg_us {float(r[3])/1e3:9.1f}")
PY
  i=$((i+1))
done
