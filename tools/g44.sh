cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/g44; mkdir -p $O
CTRANSFORMERS_AMD_LIB=$PWD/ctransformers_amd/lib_vb6/libctransformers.so timeout 900 python -m pytest tests -m gpu -q -x -k "chain or smoke or parity_llama" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for i in 1 2 3; do
for L in lib lib_vb6 lib_base; do
  CTRANSFORMERS_AMD_LIB=$PWD/ctransformers_amd/$L/libctransformers.so timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-long-context --steps 256 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$L', d['value'], d['prefill_tok_s'])"
done; done 2>&1 | tee $O/bench_ab.txt
