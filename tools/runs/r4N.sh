#!/bin/bash
# chunk quantize kernel: the eight tokens of a 128-byte image line on one XCD (CT_AMD_PGQ_REMAP=1) / one workgroup per token in order (0): prompt rate and the kernel's duration
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4N; mkdir -p $O
M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
for v in 1 0 1 0; do
  ( CT_AMD_PGQ_REMAP=$v timeout 300 python tools/prefill_sweep.py $M 128 2>&1 | tail -1 | sed "s/^/remap=$v /" ) >> $O/prefill.txt
done
cd /tmp
for v in 1 0; do
CT_AMD_PGQ_REMAP=$v CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof$v -o pf -- python $GRAFT_REPO_ROOT/tools/decode_loop.py --model $M --prompt 128 --decode 2 > $GRAFT_REPO_ROOT/$O/prof$v.log 2>&1
( echo "remap=$v"; grep "pg_quantize\|matmul_pg" $(find $GRAFT_REPO_ROOT/$O/prof$v -name "*kernel_stats.csv" | head -1) | cut -c1-160 ) >> $GRAFT_REPO_ROOT/$O/kstats.txt
done
cd $GRAFT_REPO_ROOT; rm -rf $O/prof0 $O/prof1
cat $O/prefill.txt $O/kstats.txt
