timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "context" 2>&1 | tail -2
for v in 1; do echo "CT_AMD_ATTN_SHARE=$v"; CT_AMD_ATTN_SHARE=$v timeout 300 python tools/attn_trace_ctx.py llama-7b-2l 2>&1 | grep -v amdgpu | tail -4;  CT_AMD_ATTN_SHARE=$v timeout 300 python tools/attn_trace_ctx.py 2>&1 | grep -v amdgpu | tail -6; done
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d.get("prefill_2k_tok_s"), d.get("decode_tok_s_at_2k"))'
for v in 1 0 1 0; do CT_AMD_ATTN_SHARE=$v timeout 600 python bench.py --steps 64 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "$P"; done
