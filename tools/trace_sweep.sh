#!/bin/bash
# sweeps the trace kernel's launch parameters (env-tunable) with the bench workload
for cfg in "8 4 44" "4 1 44" "2 1 44" "2 2 44" "3 1 56" "4 1 56" "2 1 60" "4 2 56" "8 1 44" "1 1 44" "4 1 32"; do
  set -- $cfg
  out=$(VOXHIP_TRACE_STEPS=$1 VOXHIP_TRACE_ITERS=$2 VOXHIP_TRACE_REFILL=$3 timeout -k 5 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernels']['k_trace']['avg_ms'])")
  echo "steps=$1 iters=$2 refill=$3 -> k_trace ms $out"
done
