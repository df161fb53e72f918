#!/usr/bin/env bash
# ON THE GPU BOX: alternate tools/k4_time.py between the default build and the variants, <reps> times, and print the
# per-library mean / min.   usage: tools/k4_ab.sh <tag> <reps> [variant.so ...]
set -uo pipefail
TAG="${1:?tag}"; REPS="${2:?reps}"; shift 2; ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"; : > "$OUT/ab.log"
for rep in $(seq "$REPS"); do
  for lib in default "$@"; do
    [ "$lib" = default ] && unset HBR_LIB || export HBR_LIB="$ROOT/$lib"
    timeout -k 10 120 python3 "$ROOT/tools/k4_time.py" 2>&1 | grep -E "fingerprint|mlp_bwd" >> "$OUT/ab.log"
  done
done
python3 - "$OUT/ab.log" <<'PY'
import re, sys, collections
t, fp = collections.defaultdict(list), collections.defaultdict(set)
last = None
for line in open(sys.argv[1]):
    if line.startswith("fingerprint"): last = line.strip()
    m = re.match(r"(\S+) mlp_bwd bf16 ([0-9.]+) ms", line)
    if m:
        t[m.group(1).split("/")[-1]].append(float(m.group(2))); fp[m.group(1).split("/")[-1]].add(last)
ref = None
for k, v in t.items():
    print(f"{k:28s} mean {sum(v)/len(v):.4f} min {min(v):.4f} ms  runs {['%.4f' % x for x in v]}  fingerprints {len(fp[k])}")
print("all fingerprints equal:", len(set().union(*fp.values())) == 1)
PY
