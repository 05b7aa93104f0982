#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4d; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -x -q -k "eight_stages or big_config or weight_population" 2>&1 | tail -15 ) > $O/pytest_new.txt
( CTAMD_BENCH_DEVICES=0,0,0,0 timeout 600 python bench.py --gpus 4 --steps 32 --no-cpu-baseline --no-other-configs 2>&1 | tail -2 ) > $O/bench_gpus4_onegpu.txt
cat $O/pytest_new.txt; cat $O/bench_gpus4_onegpu.txt; df -h /tmp | tail -1
