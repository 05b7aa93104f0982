#!/bin/bash
# no table lookups (exp table in decode attention, silu table in the gate+up epilogue): timing only, wrong results / baseline
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4M; mkdir -p $O
for v in exp7 base exp7 base; do
  L=""; [ $v != base ] && L=$PWD/ctransformers_amd/lib_$v/libctransformers.so
  ( SITES_LIB=$L timeout 300 python tools/gpu_sites.py $v 2>&1 | tail -1 ) >> $O/sites.txt
done
cat $O/sites.txt | cut -c1-330
