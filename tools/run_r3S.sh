cd /root/repo
O=gpurun_out/r3S; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_context_2048_gqa_64_8 or test_context_above_8192 or test_chunk_attention_8_tokens or test_config2_full_size or tiny or falcon-small or llama-70b-2l" -p no:cacheprovider > $O/pytest_ctx.log 2>&1; tail -2 $O/pytest_ctx.log
timeout 300 python tools/ctx_scaling.py llama-7b-2l > $O/ctx_7b.txt 2>&1; cut -c1-110 $O/ctx_7b.txt
timeout 300 python tools/ctx_scaling.py > $O/ctx_70b.txt 2>&1; cut -c1-110 $O/ctx_70b.txt
timeout 300 python tools/attn_trace_ctx.py > $O/attn_trace_70b.txt 2>&1; grep -E "n_kv|wave  [047]" $O/attn_trace_70b.txt
