mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/t8.log 2>&1; tail -15 gpurun_out/t8.log
