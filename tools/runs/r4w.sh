#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4w; mkdir -p $O
for f in Q4_K_M Q8_0; do
( SITES_LIB=$PWD/ctransformers_amd/lib_plain/libctransformers.so timeout 300 python tools/prefill_quick.py plain llama-2-7b $f 2>&1 | tail -1 ) >> $O/pf.txt
( timeout 300 python tools/prefill_quick.py nt llama-2-7b $f 2>&1 | tail -1 ) >> $O/pf.txt
done
cat $O/pf.txt
