mkdir -p gpurun_out/r2l
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 900 -k "multi or cli_contract" > gpurun_out/r2l/pytest.log 2>&1; tail -25 gpurun_out/r2l/pytest.log
