#!/usr/bin/env bash
# Collect the rocprofv3 evidence for one round ON THE GPU BOX (run through gpurun):
#   tools/profile_round.sh <tag>      e.g. r01_g   -> gpurun_out/<tag>/{kt,fetch,write}/...
# Three separate runs of the same bench command: kernel trace + stats, then one PMC pass per HBM counter
# (counters are never combined with any trace domain other than --kernel-trace).
set -euo pipefail
TAG="${1:?tag}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 python3 "$ROOT/bench.py" --steps 50 --warmup 10 > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o run -- python3 "$ROOT/bench.py" --steps 25 --warmup 5 > "$OUT/kt.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o run -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 > "$OUT/fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o run -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 > "$OUT/write.log" 2>&1
echo "profile_round: done -> $OUT"
