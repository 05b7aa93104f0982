#!/bin/bash
# coalesced activation requests in the mat-vec prologue (V9_EXP=6, timing only): per-site us and in-kernel stamps against the baseline
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4B; mkdir -p $O
for v in exp6 base; do
  L=""; [ $v != base ] && L=$PWD/ctransformers_amd/lib_$v/libctransformers.so
  ( SITES_LIB=$L timeout 300 python tools/gpu_sites.py $v 2>&1 | tail -1 ) >> $O/sites.txt
done
for v in trace6 trace; do
  echo "## $v" >> $O/trace.txt
  ( SITES_LIB=$PWD/ctransformers_amd/lib_$v/libctransformers.so timeout 300 python tools/gpu_trace.py 2>&1 | grep -A8 "^qkv\|^wo\|^gate_up\|^down" | grep -v lm_head ) >> $O/trace.txt
done
cat $O/sites.txt; cat $O/trace.txt
