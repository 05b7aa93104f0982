#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_weight_population.py -m gpu -x -q -k "golden or bit_identical_to_reference_build or config2 or hip_build or full_7b" 2>&1 | tail -5 ) > $O/pytest.txt
( timeout 300 python tools/gpu_sites.py split 2>&1 | tail -1 ) > $O/sites.txt
( CT_AMD_V9_SPLIT=0 timeout 300 python tools/gpu_sites.py nosplit 2>&1 | tail -1 ) >> $O/sites.txt
( timeout 300 python tools/gpu_trace.py 2>&1 | grep -E -A8 "^down|^wo" ) > $O/trace.txt
cat $O/pytest.txt $O/sites.txt $O/trace.txt
