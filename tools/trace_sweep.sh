#!/bin/bash
# sweeps the trace kernel's launch parameters (env-tunable) with the bench workload
for cfg in "1024 8 44" "1024 1 44" "1024 1 56" "1024 2 48" "512 1 48" "768 1 48" "1280 1 48" "1024 4 32" "2048 1 48" "1024 1 63"; do
  set -- $cfg
  out=$(VOXHIP_TRACE_BLOCKS=$1 VOXHIP_TRACE_STEPS=$2 VOXHIP_TRACE_REFILL=$3 timeout -k 5 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernels']['k_trace']['avg_ms'])")
  echo "blocks=$1 steps=$2 refill=$3 -> k_trace ms $out"
done
