cd /root/repo
O=gpurun_out/r3L; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_context_2048_gqa_64_8 or test_context_above_8192 or test_chunk_attention_8_tokens" -p no:cacheprovider > $O/pytest_ctx.log 2>&1; tail -2 $O/pytest_ctx.log
timeout 300 python tools/ctx_scaling.py llama-7b-2l > $O/ctx_7b_vlds.txt 2>&1; cat $O/ctx_7b_vlds.txt | cut -c1-200
CT_AMD_ATTN_VLDS=0 timeout 300 python tools/ctx_scaling.py llama-7b-2l > $O/ctx_7b_ring.txt 2>&1; cat $O/ctx_7b_ring.txt | cut -c1-200
timeout 300 python tools/ctx_scaling.py > $O/ctx_70b_vlds.txt 2>&1; cat $O/ctx_70b_vlds.txt | cut -c1-200
CT_AMD_ATTN_VLDS=0 timeout 300 python tools/ctx_scaling.py > $O/ctx_70b_ring.txt 2>&1; cat $O/ctx_70b_ring.txt | cut -c1-200
timeout 300 python tools/attn_trace_ctx.py > $O/attn_trace_70b.txt 2>&1; tail -18 $O/attn_trace_70b.txt
