mkdir -p gpurun_out/g6; R=$PWD; M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
timeout 900 python -m pytest tests/test_greedy_chain.py -m gpu -x -q > gpurun_out/g6/pytest_chain.txt 2>&1
cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
  CT_AMD_HEAD_FOLD=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/g6/tr_$v -o t -- python $R/tools/decode_loop.py --model $M --prompt 128 --decode 40 > $R/gpurun_out/g6/tr_$v.log 2>&1
  python $R/tools/timeline.py $R/gpurun_out/g6/tr_$v > $R/gpurun_out/g6/timeline_$v.txt 2>&1
done
cd $R; find gpurun_out/g6 -name "*.csv" -delete
tail -5 gpurun_out/g6/pytest_chain.txt
for v in 1 0; do grep -E "^1[56][0-9] |per token|pick_cont|argmax|copyBuf|embed" gpurun_out/g6/timeline_$v.txt; done
for v in 1 2 0 1 2 0; do CT_AMD_HEAD_FOLD=$v timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | head -c 120 | cut -c40-120; echo " fold=$v"; done
CT_AMD_SPEC=0 timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | head -c 120 | cut -c40-120; echo " fold=1 nospec"
