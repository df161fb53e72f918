#!/usr/bin/env bash
# ON THE GPU BOX: one-GPU RCCL rehearsal (bench.py HBR_RCCL_REHEARSAL=1) for several split levels of the staged all-reduce.
set -uo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; OUT="$ROOT/gpurun_out/${1:?tag}"; mkdir -p "$OUT"
Q="--no-cpu-baseline --no-dropin --steps 200 --warmup 20"
summ() { python3 -c "import sys,json; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], 'ms/step %.4f' % d['ms_per_step'], 'exposed', d['allreduce_exposed_ms'], 'hash_bwd ms %.4f' % (1164*0+d['kernels']['hash_bwd']['achieved'] and (1100*2048000/ d['kernels']['hash_bwd']['achieved']/1e6)))" "$1" "$2"; }
timeout -k 10 300 python3 "$ROOT/bench.py" $Q > "$OUT/plain.json" 2> "$OUT/err.log"; summ "$OUT/plain.json" plain
export HBR_RCCL_REHEARSAL=1
HBR_OVERLAP_COMM=0 timeout -k 10 300 python3 "$ROOT/bench.py" $Q > "$OUT/single.json" 2>> "$OUT/err.log"; summ "$OUT/single.json" single
for lv in 2 4 6 8; do
  HBR_OVERLAP_COMM=1 HBR_SPLIT_LEVEL=$lv timeout -k 10 300 python3 "$ROOT/bench.py" $Q > "$OUT/split$lv.json" 2>> "$OUT/err.log"; summ "$OUT/split$lv.json" "split_level=$lv"
done
timeout -k 10 300 python3 "$ROOT/bench.py" $Q > "$OUT/auto.json" 2>> "$OUT/err.log"; summ "$OUT/auto.json" auto
python3 -c "import sys,json; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['comm_autotune'], d['split_scatter'])" "$OUT/auto.json"
