#!/usr/bin/env bash
# ON THE GPU BOX: tools/k4_time.py (time + bit fingerprints) for the default build and each variant library given.
# usage: tools/k4_quick.sh <tag> [variant.so ...]
set -uo pipefail
TAG="${1:?tag}"; shift; ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
for lib in default "$@"; do
  [ "$lib" = default ] && unset HBR_LIB || export HBR_LIB="$ROOT/$lib"
  timeout -k 10 120 python3 "$ROOT/tools/k4_time.py" 2>&1 | grep -E "fingerprint|mlp_bwd" | tee -a "$OUT/k4quick.log"
done
