#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4p; mkdir -p $O
( timeout 300 python tools/host_overhead.py 2>&1 | tail -1 ) > $O/host.txt
( CT_AMD_PF=0 timeout 300 python tools/host_overhead.py 2>&1 | tail -1 ) >> $O/host.txt
cat $O/host.txt
