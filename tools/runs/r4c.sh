#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
( SITES_LIB=$PWD/ctransformers_amd/lib_trace/libctransformers.so timeout 300 python tools/gpu_trace.py 2>&1 | tail -60 ) > $O/trace_pro.txt
cat $O/trace_pro.txt
