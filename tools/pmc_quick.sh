#!/bin/bash
# two PMC passes for k_trace: instruction counts and lane utilisation
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_q*
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_q$i -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --big-rays 0 > $R/gpurun_out/pmc_q$i.log 2>&1 || { echo "group $i failed"; tail -5 $R/gpurun_out/pmc_q$i.log; }
done
cd $R && python tools/pmc_summary.py gpurun_out/pmc_q* | grep -E "k_trace|k_voxelize|k_emit_units"
