#!/bin/bash
for b in 512 640 768 896 1024; do
  VOXHIP_TRACE_BLOCKS=$b timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --big-rays 8000000 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('blocks=$b', 'k_trace', d['roofline']['avg_launch_ms'], 'step', d['ms_per_step'], '8M', d['trace_large_batch']['ms'])"
done
