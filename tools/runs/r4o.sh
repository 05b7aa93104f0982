#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or bit_identical_to_reference_build or config2 or context" 2>&1 | tail -3 ) > $O/pytest.txt
for i in 1 2; do ( timeout 300 python tools/gpu_sites.py $1 2>&1 | tail -1 ) >> $O/sites.txt; done
( timeout 300 python tools/gpu_trace.py 2>&1 | grep -E -A4 "^attn" ) > $O/trace.txt
( timeout 600 python tools/ctx_scaling.py 2>&1 | tail -12 ) > $O/ctx.txt
cat $O/pytest.txt $O/sites.txt $O/trace.txt $O/ctx.txt
