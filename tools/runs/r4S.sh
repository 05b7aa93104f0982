#!/bin/bash
# three records before the prologue for the Q6_K launches too (lib_pre3: -DV9_PRE=3) / two (base)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4S; mkdir -p $O
for v in pre3 base pre3 base; do
  L=""; [ $v != base ] && L=$PWD/ctransformers_amd/lib_$v/libctransformers.so
  ( SITES_LIB=$L timeout 300 python tools/gpu_sites.py $v 2>&1 | tail -1 ) >> $O/sites.txt
done
cat $O/sites.txt | cut -c1-330
