#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4g; mkdir -p $O
( CT_AMD_DBG=64 timeout 300 python tools/gpu_sites.py null 2>&1 | tail -1 ) >> $O/sites.txt
( timeout 300 python tools/gpu_sites.py base 2>&1 | tail -1 ) >> $O/sites.txt
cat $O/sites.txt
