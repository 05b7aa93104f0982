cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/g45; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "context" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for i in 1 2; do
for L in lib lib_base; do
  echo "== $L"
  CTRANSFORMERS_AMD_LIB=$PWD/ctransformers_amd/$L/libctransformers.so timeout 300 python tools/ctx_scaling.py llama-7b-2l 2>&1 | grep -o '"pos": [0-9]*\|"attn_fused@sweep": [0-9.]*' | paste - - | tr '\n' ' '; echo
  CTRANSFORMERS_AMD_LIB=$PWD/ctransformers_amd/$L/libctransformers.so timeout 300 python tools/ctx_scaling.py 2>&1 | grep -o '"pos": [0-9]*\|"attn_fused@sweep": [0-9.]*' | paste - - | tr '\n' ' '; echo
done; done 2>&1 | tee $O/ab.txt
for i in 1 2 3; do
for L in lib lib_base; do
  CTRANSFORMERS_AMD_LIB=$PWD/ctransformers_amd/$L/libctransformers.so timeout 600 python bench.py --no-cpu-baseline --no-other-configs --steps 32 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$L', d['value'], d.get('prefill_2k_tok_s'), d.get('decode_tok_s_at_2k'))"
done; done 2>&1 | tee $O/bench_ab.txt
