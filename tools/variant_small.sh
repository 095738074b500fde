#!/bin/bash
for fl in "$@"; do
  VOXHIP_EXTRA_FLAGS="$fl" python raytracing-voxilizer-vulkan-intresection_amd/build.py > gpurun_out/variant_build.log 2>&1 || { tail -5 gpurun_out/variant_build.log; continue; }
  timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --big-rays 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_survey_pass']; print('[$fl]', 'scan', k['k_scan_onepass']['avg_ms'], 'bbox', k['k_bbox']['avg_ms'], 'stage', d['stages_ms']['voxelize'], 'step', d['ms_per_step'])"
done
