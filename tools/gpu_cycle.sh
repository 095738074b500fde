#!/bin/bash
# one GPU round trip: parity tests, bench (1M + 8M rays), then the utilisation diagnostic on a -DVX_TRACE_DEBUG_UTIL build
set -o pipefail
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log
timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --big-rays 8000000 > gpurun_out/b.log 2>&1
python - <<'PY'
import json
l=[x for x in open("gpurun_out/b.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1]); print("value", j["value"], "ms/step", j["ms_per_step"], "k_dom", j["roofline"]["kernel"], j["roofline"]["avg_launch_ms"], j["stages_ms"], "8M:", j["trace_large_batch"])
else:
    print(open("gpurun_out/b.log").read()[-2000:])
PY
if [ "$1" = "util" ]; then
  VOXHIP_EXTRA_FLAGS="-DVX_TRACE_DEBUG_UTIL $VOXHIP_EXTRA_FLAGS" python raytracing-voxilizer-vulkan-intresection_amd/build.py > gpurun_out/util_build.log 2>&1
  timeout -k 10 200 python tools/util_hist.py 1000000 2>&1 | tail -11
  timeout -k 10 200 python tools/util_hist.py 8000000 2>&1 | tail -11
fi
