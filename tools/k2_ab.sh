#!/usr/bin/env bash
# ON THE GPU BOX: alternate tools/k2_time.py between the default build and the variants; bf16-dy line only.
# usage: tools/k2_ab.sh <reps> [variant.so ...]
set -uo pipefail
REPS="${1:?reps}"; shift; ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
for rep in $(seq "$REPS"); do
  for lib in default "$@"; do
    [ "$lib" = default ] && unset HBR_LIB || export HBR_LIB="$ROOT/$lib"
    timeout -k 10 120 python3 "$ROOT/tools/k2_time.py" 2>&1 | grep "bfloat16"
  done
done
