#!/bin/bash
# builds the library with extra -D flags ON THE GPU BOX and runs the bench: tools/variant.sh "<flags>" [...]
for fl in "$@"; do
  VOXHIP_EXTRA_FLAGS="$fl" python raytracing-voxilizer-vulkan-intresection_amd/build.py > gpurun_out/variant_build.log 2>&1 || { tail -5 gpurun_out/variant_build.log; continue; }
  timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --big-rays 8000000 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$fl]', 'k_trace', d['roofline']['avg_launch_ms'], 'step', d['ms_per_step'], '8M', d['trace_large_batch']['ms'])"
done
