#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
( timeout 200 tools/experiments/valu_rate.bin 2>&1 ) > $O/valu_rate.txt
cat $O/valu_rate.txt
