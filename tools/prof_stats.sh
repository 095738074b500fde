#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench (no PMC, no other trace domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -o kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --big-rays 0 > $R/gpurun_out/prof_kt.log 2>&1
cd $R && python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_kt/**/*kernel_stats.csv", recursive=True)
for row in csv.DictReader(open(f[0])):
    print("%-60s calls %6s avg %10.1f ns  total %12s  %5s %%" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]), row["TotalDurationNs"], row["Percentage"]))
PY
