#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4O; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > $O/pytest.txt
cat $O/pytest.txt
