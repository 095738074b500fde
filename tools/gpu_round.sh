#!/bin/bash
# One GPU-box pass: the -m gpu suite, then the default bench, then a 2-rank rehearsal of the multi-rank path (gloo, one device).
# Usage (through gpurun): bash tools/gpu_round.sh <tag>
TAG=${1:-run}
O=gpurun_out/$TAG
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=20 --timeout 900 > $O/pytest.log 2>&1
rc=$?
tail -5 $O/pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest killed at its limit: stopping"; exit 1; fi
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print({k: d[k] for k in ("value", "ms_per_step", "verified", "stages_ms", "trace_large_batch")})
print({k: v["avg_ms"] for k, v in d["kernels_survey_pass"].items()})
PY
timeout -k 10 400 python bench.py --gpus 2 --single-device --dist-backend gloo --steps 5 --no-cpu-baseline --big-rays 0 > $O/bench_2rank.json 2> $O/bench_2rank.err || { echo "2-rank rehearsal failed"; tail -5 $O/bench_2rank.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench_2rank.json')); print('2-rank', d['n_gpus'], d['value'], d['verified'], d.get('exchange'), d.get('c4_1024'))"
