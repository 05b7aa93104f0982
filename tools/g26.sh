PV=$PWD/ctransformers_amd/lib_prev/libctransformers.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_greedy_chain.py -m gpu -x -q -k "golden or bit_identical_to_reference_build or chain" 2>&1 | tail -2
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["prefill_tok_s"])'
for i in 1 2 3; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-long-context 2>/dev/null | python -c "$P"
CTRANSFORMERS_AMD_LIB=$PV timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-long-context 2>/dev/null | python -c "$P"
done
python tools/qa_trace.py 2>&1 | grep -v amdgpu | cut -c1-170 | sed -n '1p;2p;15p;16p;17p'
