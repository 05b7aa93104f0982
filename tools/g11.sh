mkdir -p gpurun_out/g11
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or abi_semantics" > gpurun_out/g11/pytest_golden.txt 2>&1; tail -5 gpurun_out/g11/pytest_golden.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bit_identical_to_reference_build" > gpurun_out/g11/pytest_ref.txt 2>&1; tail -5 gpurun_out/g11/pytest_ref.txt
timeout 900 python -m pytest tests/test_greedy_chain.py -m gpu -x -q > gpurun_out/g11/pytest_chain.txt 2>&1; tail -3 gpurun_out/g11/pytest_chain.txt
for v in 1 0 1 0; do CT_AMD_FUSE_QA=$v timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | head -c 120 | cut -c40-120; echo " fuse=$v"; done
