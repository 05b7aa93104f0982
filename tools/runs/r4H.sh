#!/bin/bash
# ring of three slots for the single-type Q4_K launches / four
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4H; mkdir -p $O
for v in ns3 base ns3 base; do
  L=""; [ $v != base ] && L=$PWD/ctransformers_amd/lib_$v/libctransformers.so
  ( SITES_LIB=$L timeout 300 python tools/gpu_sites.py $v 2>&1 | tail -1 ) >> $O/sites.txt
done
cat $O/sites.txt | cut -c1-330
