#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4v; mkdir -p $O
( SITES_LIB=$PWD/ctransformers_amd/lib_plain/libctransformers.so timeout 300 python tools/gpu_sites.py plain 2>&1 | tail -1 ) >> $O/sites.txt
( timeout 300 python tools/gpu_sites.py base 2>&1 | tail -1 ) >> $O/sites.txt
cat $O/sites.txt
