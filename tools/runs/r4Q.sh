#!/bin/bash
# mat-vec: logical workgroup index that keeps consecutive units on one XCD (CT_AMD_DBG=128) / hardware order
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4Q; mkdir -p $O
for v in 256 512 0 256 512 0; do
  ( CT_AMD_DBG=$v timeout 300 python tools/gpu_sites.py dbg$v 2>&1 | tail -1 ) >> $O/sites.txt
done
cat $O/sites.txt | cut -c1-330
