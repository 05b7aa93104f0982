# round 6: attn_mm_kernel on the GPU — the parity tests of tests/test_fast_prefill.py, then prompt rates of both attention forms at 128 / 512 / 2048 tokens
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fast_prefill.py -m gpu -x -q -s -p no:cacheprovider > gpurun_out/attn_mm_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/attn_mm_tests.log
for n in 128 512 2048; do
  for mm in 0 1; do
    echo "== 7B Q4_K_M $n tokens CT_AMD_ATTN_MM=$mm" >> gpurun_out/attn_mm_rates.log
    CT_AMD_ATTN_MM=$mm timeout 300 python tools/mm8_check.py llama-2-7b Q4_K_M $n 8 2304 >> gpurun_out/attn_mm_rates.log 2>&1
  done
done
tail -5 gpurun_out/attn_mm_tests.log; cat gpurun_out/attn_mm_rates.log
