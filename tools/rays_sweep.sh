#!/usr/bin/env bash
# ON THE GPU BOX: the fused step at several batch sizes (what a rank sees under --scaling strong on 8 / 4 / 2 / 1 GPUs, and larger)
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
for r in "$@"; do
  timeout -k 10 200 python3 "$ROOT/bench.py" --rays "$r" --steps 100 --warmup 10 --no-dropin --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); N = d['config']['rays_per_rank'] * 128
ms = {k: (1100 if 'hash' in k else 0) for k in d['kernels']}
def t(k):
    r = d['kernels'][k]
    per = {'hash_fwd': 1100, 'hash_bwd': 1100, 'mlp_fwd': 27904, 'mlp_bwd': 83712}[k]
    return per * N / (r['achieved'] * (1e9 if r['unit'] == 'GB/s' else 1e12)) * 1e3
print(d['config']['rays_per_rank'], 'rays: ms/step %.4f' % d['ms_per_step'], 'value %.3e' % d['value'], ' '.join('%s %.3f' % (k, t(k)) for k in d['kernels']))"
done
