mkdir -p gpurun_out/g2
timeout 900 python -m pytest tests/test_greedy_chain.py -m gpu -x -q > gpurun_out/g2/pytest_chain.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or abi_semantics or full_7b or config2" > gpurun_out/g2/pytest_some.txt 2>&1
timeout 300 python tools/host_gap.py > gpurun_out/g2/host_gap.txt 2>&1
CT_AMD_SPEC=0 timeout 300 python tools/host_gap.py > gpurun_out/g2/host_gap_nospec.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/g2/bench.json 2> gpurun_out/g2/bench.err
CT_AMD_SPEC=0 timeout 600 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/g2/bench_nospec.json 2> gpurun_out/g2/bench_nospec.err
CT_AMD_SPEC=0 CT_AMD_HEAD_FOLD=0 timeout 600 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/g2/bench_nofold.json 2> gpurun_out/g2/bench_nofold.err
tail -15 gpurun_out/g2/pytest_chain.txt; tail -5 gpurun_out/g2/pytest_some.txt; cat gpurun_out/g2/host_gap*.txt; for f in bench bench_nospec bench_nofold; do head -c 300 gpurun_out/g2/$f.json; echo; done
