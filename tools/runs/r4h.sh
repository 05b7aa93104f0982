#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4h; mkdir -p $O
( timeout 300 tools/experiments/launch_cost.bin 2>&1 ) > $O/launch_cost.txt
cat $O/launch_cost.txt
