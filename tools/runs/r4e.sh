#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or bit_identical_to_reference_build or config2" 2>&1 | tail -5 ) > $O/pytest.txt
( timeout 300 python tools/gpu_sites.py $1 2>&1 | tail -1 ) > $O/sites.txt
cat $O/pytest.txt $O/sites.txt
