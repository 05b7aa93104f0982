#!/bin/bash
# ring slots per wave of ALL single-type K-quant mat-vec launches: 2 / 3 / 4
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4I; mkdir -p $O
for v in ns2 ns3 base ns2 ns3 base; do
  L=""; [ $v != base ] && L=$PWD/ctransformers_amd/lib_$v/libctransformers.so
  ( SITES_LIB=$L timeout 300 python tools/gpu_sites.py $v 2>&1 | tail -1 ) >> $O/sites.txt
done
cat $O/sites.txt | cut -c1-330
