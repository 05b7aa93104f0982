timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or batch_structure or config2 or chunk_path" 2>&1 | tail -2
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["prefill_tok_s"], d.get("prefill_2k_tok_s"), d.get("decode_tok_s_at_2k"))'
for i in 1 2 3; do
timeout 600 python bench.py --steps 32 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "$P"
CT_AMD_PGQ_NARROW=0 timeout 600 python bench.py --steps 32 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "$P"
done
