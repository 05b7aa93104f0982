#!/bin/bash
# token-group size chosen per weight-type launch (the Q6_K attn_v launch beside Q4_K q / k takes 16-token groups: 256 workgroups instead of 128) (new) / per site (old)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4R; mkdir -p $O
M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
for v in new old new old; do
  L=""; [ $v = old ] && L=$PWD/ctransformers_amd/lib_old/libctransformers.so
  ( CTRANSFORMERS_AMD_LIB=$L timeout 300 python tools/prefill_sweep.py $M 128 2>&1 | tail -1 | sed "s/^/$v /" ) >> $O/prefill.txt
done
cd /tmp
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o pf -- python $GRAFT_REPO_ROOT/tools/decode_loop.py --model $M --prompt 128 --decode 2 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
grep "matmul_pg\|pg_quantize" $(find $GRAFT_REPO_ROOT/$O/prof -name "*kernel_stats.csv" | head -1) | cut -c1-200 >> $GRAFT_REPO_ROOT/$O/kstats.txt
cd $GRAFT_REPO_ROOT; rm -rf $O/prof
( timeout 1500 python -m pytest tests/test_weight_population.py tests/test_gpu_parity.py -m gpu -x -q -k "not big_config and not eight_stages" 2>&1 | tail -3 ) > $O/pytest.txt
cat $O/prefill.txt $O/kstats.txt $O/pytest.txt
