#!/bin/bash
# ring slots: ns4 = 4 everywhere (the round's form), new = 3 for the single-type K-quant launches, k23 = 3 for the two-type (QKV) launch as well
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4J; mkdir -p $O
for v in k23 new ns4 k23 new ns4; do
  L=""; [ $v != new ] && L=$PWD/ctransformers_amd/lib_$v/libctransformers.so
  ( SITES_LIB=$L timeout 300 python tools/gpu_sites.py $v 2>&1 | tail -1 ) >> $O/sites.txt
done
cat $O/sites.txt | cut -c1-330
