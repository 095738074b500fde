#!/bin/bash
# builds the library with extra -D flags ON THE GPU BOX and runs the rebuild micro-benchmark: tools/variant_vox.sh "<flags>" [...]
# ("" = the committed defaults).  Output: the Vec line of tools/vox_time.py per variant in gpurun_out/variants_vox.log
mkdir -p gpurun_out
for fl in "$@"; do
  VOXHIP_VARIANT_ONLY=${VOXHIP_VARIANT_ONLY:-vx_kernels.hip} VOXHIP_EXTRA_FLAGS="$fl" python raytracing-voxilizer-vulkan-intresection_amd/build.py > gpurun_out/variant_build.log 2>&1 || { echo "[$fl] build failed"; tail -5 gpurun_out/variant_build.log; continue; }
  echo "[$fl] $(timeout -k 10 300 python tools/vox_time.py 2>&1 | tail -1)" | tee -a gpurun_out/variants_vox.log
done
VOXHIP_VARIANT_ONLY=${VOXHIP_VARIANT_ONLY:-vx_kernels.hip} python raytracing-voxilizer-vulkan-intresection_amd/build.py > /dev/null 2>&1
