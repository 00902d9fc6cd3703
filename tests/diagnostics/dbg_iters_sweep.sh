#!/bin/bash
# tick time vs PGS sweep count (MRE_DEBUG_ITERS): separates the per-sweep cost from the rest of a tick
for it in 0 1 25 50 100; do
  MRE_DEBUG_ITERS=$it python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null > /tmp/b_$it.json
  python -c "import json; d=json.load(open('/tmp/b_$it.json')); print('iters', $it, 'ms/tick', round(d['ms_per_step'],3), 'nan', d['health']['nan_envs'])"
done
