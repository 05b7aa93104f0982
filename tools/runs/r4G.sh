#!/bin/bash
# the slot's next record requested BEFORE the block math (math on a register copy), single-type Q4_K launches only (125 VGPRs, no spills now) / baseline
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4G; mkdir -p $O
for v in exp3 base exp3 base; do
  L=""; [ $v != base ] && L=$PWD/ctransformers_amd/lib_$v/libctransformers.so
  ( SITES_LIB=$L timeout 300 python tools/gpu_sites.py $v 2>&1 | tail -1 ) >> $O/sites.txt
done
cat $O/sites.txt | cut -c1-330
