#!/bin/bash
for n in 2000000 4000000; do for b in 640 768 896 1024; do
  VOXHIP_TRACE_BLOCKS=$b timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --big-rays $n 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rays=$n blocks=$b', d['trace_large_batch']['ms'])"
done; done
