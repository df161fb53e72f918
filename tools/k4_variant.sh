#!/usr/bin/env bash
# Build a variant of the library whose mlp.hip is compiled with extra -D flags (CPU side, before gpurun).
# usage: tools/k4_variant.sh <name> [-DFOO=1 ...]   ->  human-body-reconstruction_amd/csrc/build/var_<name>.so
set -euo pipefail
NAME="${1:?name}"; shift
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; HERE="$ROOT/human-body-reconstruction_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-value -I"$ROOT/include" -I"$HERE" \
  -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c "$HERE/mlp.hip" -o "$HERE/build/mlp_$NAME.o"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$HERE/build/var_$NAME.so" "$HERE"/build/{c_api,sample,render,hash_encode,hash_scatter,composite,optim}.o "$HERE/build/mlp_$NAME.o"
echo "human-body-reconstruction_amd/csrc/build/var_$NAME.so"
