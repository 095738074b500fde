#!/bin/bash
# sweeps the donation thresholds of k_trace (temporary env knobs) with the bench workload
for cfg in "48 6" "63 6" "32 6" "48 3" "48 12" "63 3" "63 2" "56 4" "63 12"; do
  set -- $cfg
  VOXHIP_DON_BELOW=$1 VOXHIP_DON_BRICKS=$2 timeout -k 5 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --big-rays 8000000 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('below=$1 bricks=$2', d['roofline']['avg_launch_ms'], d['ms_per_step'], d['stages_ms'], d['trace_large_batch']['ms'])"
done
