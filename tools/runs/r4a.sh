#!/bin/bash
# round 4, GPU call 1: parity on hardware for reference-quantized / arbitrary-byte weights, VALU issue rates, baseline sites + trace
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
( timeout 900 python -m pytest tests/test_weight_population.py -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_weight_population.txt
( timeout 120 tools/experiments/valu_rate.bin 2>&1 ) > $O/valu_rate.txt
( timeout 300 python tools/gpu_sites.py base 2>&1 | tail -3 ) > $O/sites_base.txt
( timeout 300 python tools/gpu_trace.py 2>&1 | tail -40 ) > $O/trace_base.txt
tail -5 $O/pytest_weight_population.txt; head -40 $O/valu_rate.txt; cat $O/sites_base.txt
