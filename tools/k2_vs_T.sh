#!/usr/bin/env bash
# ON THE GPU BOX: the fused step at several table sizes (train_hash2.py:36 --hash_size): ms/step and the K2 span
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
for h in "$@"; do
  timeout -k 10 300 python3 "$ROOT/bench.py" --hash-size "$h" --steps 40 --warmup 6 --no-dropin --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('hash_size %2d: ms/step %.4f  value %.3e  ' % ($h, d['ms_per_step'], d['value']) + '  '.join('%s %.4f ms' % (n, k[n]['avg_ms']) for n in k), flush=True)" || exit 1
done
