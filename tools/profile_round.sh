#!/usr/bin/env bash
# Collect the rocprofv3 evidence for one round ON THE GPU BOX (run through gpurun):
#   tools/profile_round.sh <tag>      e.g. r03_a   -> gpurun_out/<tag>/{kt,kt_dropin,fetch,write,valu,clk,tcp}/...
# Separate runs of the same bench command: kernel trace + stats (fused step; then the drop-in loop), one PMC pass per
# HBM counter, and the issue-side counters (VALU instructions, GRBM clock, L1 line look-ups).  Counters are never
# combined with any trace domain other than --kernel-trace.
set -uo pipefail
TAG="${1:?tag}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# what the counters below are taken on: the kernel sources' fingerprint (bench.py flags stored counters whose fingerprint
# is not the running library's: traffic_source_stale) and the library file's own hash
python3 -c "import sys; sys.path.insert(0, '$ROOT'); from hbr_amd import _lib; import hashlib; print(_lib.kernel_source_sha()); print(hashlib.sha256(open(_lib.LIB_PATH,'rb').read()).hexdigest()[:16])" > "$OUT/kernel_source_sha.txt"
timeout -k 10 300 python3 "$ROOT/bench.py" --steps 50 --warmup 10 > "$OUT/bench.json" 2> "$OUT/bench.err"
Q="--no-cpu-baseline --no-dropin"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o run -- python3 "$ROOT/bench.py" --steps 25 --warmup 5 $Q > "$OUT/kt.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_dropin" -o run -- python3 "$ROOT/bench.py" --only-dropin --steps 25 --warmup 5 > "$OUT/kt_dropin.log" 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc$i" -o run -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 $Q > "$OUT/pmc$i.log" 2>&1 || echo "pmc set $i failed: $set"
done
echo "profile_round: done -> $OUT"
