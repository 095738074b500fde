#!/bin/bash
# PMC passes for k_trace (one counter group per run; rocprofv3 --pmc alone, no trace domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_I8 SQ_ACTIVE_INST_LDS" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_t$i -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --big-rays 0 ${EXTRA_BENCH} > $R/gpurun_out/pmc_t$i.log 2>&1 || { echo "group $i failed"; tail -5 $R/gpurun_out/pmc_t$i.log; }
done
cd $R && python tools/pmc_summary.py gpurun_out/pmc_t* | grep -E "k_trace|k_voxelize"
