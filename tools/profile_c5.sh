#!/bin/bash
# rocprofv3 passes over BASELINE configs[4] (tools/run_c5.py): kernel trace + stats, then FETCH_SIZE / WRITE_SIZE alone.
# Results under gpurun_out/prof_c5/; tools/profile_collect_c5.py turns them into profiles/<tag>_c5_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_c5
rm -rf $O && mkdir -p $O
export C5_TILES=${C5_TILES:-2}
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/tools/run_c5.py > $O/kt.log 2>&1 || { echo "kernel trace failed"; tail -5 $O/kt.log; exit 1; }
tail -12 $O/kt.log
if [ "${C5_PMC:-1}" = "1" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python3 $R/tools/run_c5.py > $O/pmc_$c.log 2>&1 || { echo "pmc $c failed"; exit 1; }
done
fi
find $O -name "*.csv" | head -20
