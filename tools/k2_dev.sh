#!/usr/bin/env bash
# K2 development loop ON THE GPU BOX: golden checks, the K2-related GPU tests, and a rocprofv3 kernel-trace of
# tools/k2_time.py (per-kernel averages).  usage: tools/k2_dev.sh <tag>
set -uo pipefail
TAG="${1:?tag}"; ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
for g in g3_encoder_T16.npz g3_encoder_T10.npz g3_encoder_T1000.npz; do python3 "$ROOT/tools/k2_check.py" $g 2>&1 | grep -E "^algo" | sed "s/^/$g /"; done | tee "$OUT/k2check.log"
python3 -m pytest "$ROOT/tests/test_gpu_shipped_paths.py" "$ROOT/tests/test_gpu_parity.py" -x -q -k "k2 or hash or trainer or training or smoke or render" > "$OUT/pytest_k2.log" 2>&1; tail -4 "$OUT/pytest_k2.log"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o run -- python3 "$ROOT/tools/k2_time.py" > "$OUT/k2time.log" 2>&1
grep hash_bwd "$OUT/k2time.log"
python3 - "$OUT/kt/run_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:16]:
    if "hbr" in r[0] or "rocclr" in r[0]: print(f"{r[0][:64]:64s} calls {r[1]:>4s} avg_us {float(r[3])/1e3:9.1f}")
PY
