mkdir -p gpurun_out/g19
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/g19/pytest_gpu.txt 2>&1; tail -4 gpurun_out/g19/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
