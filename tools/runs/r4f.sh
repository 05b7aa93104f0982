#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4f; mkdir -p $O
for v in exp1 exp2; do
( SITES_LIB=$PWD/ctransformers_amd/lib_$v/libctransformers.so timeout 300 python tools/gpu_sites.py $v 2>&1 | tail -1 ) >> $O/sites.txt
done
( timeout 300 python tools/gpu_sites.py base 2>&1 | tail -1 ) >> $O/sites.txt
cat $O/sites.txt
