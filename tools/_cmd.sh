mkdir -p gpurun_out/r2q
python tools/trace_bench.py > gpurun_out/r2q/tb_walk.log 2>&1; tail -1 gpurun_out/r2q/tb_walk.log
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "trace or ray or shadow or tiny or far or random or reuse or primary or c3 or c4 or c5 or multi" > gpurun_out/r2q/pytest.log 2>&1; tail -5 gpurun_out/r2q/pytest.log
