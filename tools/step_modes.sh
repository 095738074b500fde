#!/bin/bash
# per-ray histograms of the walk in three diagnostic builds: all steps (1), brick tests (2), steps in cells outside the grid (3)
for m in 1 2 3; do
  VOXHIP_EXTRA_FLAGS="-DVX_TRACE_DEBUG_STEPS=$m" python raytracing-voxilizer-vulkan-intresection_amd/build.py > gpurun_out/variant_build.log 2>&1 || { tail -3 gpurun_out/variant_build.log; continue; }
  echo "mode $m"; VOXHIP_TRACE_DONATE=0 timeout -k 10 200 python tools/step_hist.py 2>&1 | tail -3
done
