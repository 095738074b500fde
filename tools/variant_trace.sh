#!/bin/bash
# builds the library with extra -D flags ON THE GPU BOX and runs the ray-stage micro-benchmark: tools/variant_trace.sh "<flags>" [...]
# ("" = the committed defaults).  Output: one line per variant in gpurun_out/variants.log
mkdir -p gpurun_out
for fl in "$@"; do
  VOXHIP_VARIANT_ONLY=${VOXHIP_VARIANT_ONLY:-vx_walk.hip} VOXHIP_EXTRA_FLAGS="$fl" python raytracing-voxilizer-vulkan-intresection_amd/build.py > gpurun_out/variant_build.log 2>&1 || { echo "[$fl] build failed"; tail -5 gpurun_out/variant_build.log; continue; }
  echo "[$fl] $(TB_NOCHECK=${TB_NOCHECK:-} timeout -k 10 300 python tools/trace_bench.py 2>&1 | tail -1)" | tee -a gpurun_out/variants.log
done
VOXHIP_VARIANT_ONLY=${VOXHIP_VARIANT_ONLY:-vx_walk.hip} python raytracing-voxilizer-vulkan-intresection_amd/build.py > /dev/null 2>&1
