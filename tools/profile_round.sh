#!/bin/bash
# Round profile of the default bench on the GPU box (results under gpurun_out/prof_round/, post-processed by
# tools/profile_collect.py into profiles/).  One rocprofv3 mode per run: kernel trace + stats, then PMC passes alone.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_round
rm -rf $O && mkdir -p $O
BENCH="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --big-rays 0 --no-context --pipelined-steps 0"
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $BENCH > $O/kt.log 2>&1 || echo "kernel trace failed"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 150 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- $BENCH > $O/pmc_$c.log 2>&1 || echo "pmc $c failed"
done
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_sq$i -o pmc -- $BENCH > $O/pmc_sq$i.log 2>&1 || echo "pmc sq$i failed"
done
grep -h "^{" $O/kt.log | tail -1 > $O/bench_under_kernel_trace.json
ls $O
