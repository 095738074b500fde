mkdir -p gpurun_out/r2i
C5_TILES=3 python tools/run_c5.py > gpurun_out/r2i/c5_walk.log 2>&1; grep -E "primary rays|Octree build|VoxelGridBool build|steady" gpurun_out/r2i/c5_walk.log
C5_TILES=3 VOXHIP_TRACE_ALGO=dda python tools/run_c5.py > gpurun_out/r2i/c5_dda.log 2>&1; grep -E "primary rays" gpurun_out/r2i/c5_dda.log
TB_GRID=1024 TB_NOCHECK=1 python tools/trace_bench.py 2>&1 | tail -1
TB_GRID=1024 TB_NOCHECK=1 VOXHIP_TRACE_ALGO=dda python tools/trace_bench.py 2>&1 | tail -1
