#!/bin/bash
# PMC passes (one counter group per run, never combined with tracing) over the ray-stage micro-benchmark: tools/pmc_tb.sh <tag>
# Summary lines for the ray kernels land in gpurun_out/<tag>/pmc_summary.txt
TAG=${1:-pmc}
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export TB_NOCHECK=1
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $O/g$i -o pmc -- python3 $R/tools/trace_bench.py > $O/g$i.log 2>&1 || { echo "group $i failed"; tail -3 $O/g$i.log; }
done
cd $R && python tools/pmc_summary.py $O/g* | grep -E "k_walk|k_trace|k_rank" > $O/pmc_summary.txt
cat $O/pmc_summary.txt
rm -rf $O/g*/
