#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_wide_rows.py tests/test_mpt.py tests/test_starcoder.py tests/test_weight_population.py -m gpu -x -q -k "Q8_0 or Q4_0 or q80 or q40 or gpt2 or mpt or starcoder or config3" 2>&1 | tail -3 ) > $O/pytest.txt
for n in 0 2 4; do
( CT_AMD_PFM_NTG=$n timeout 600 python bench.py --config 3 --no-cpu-baseline --no-other-configs --steps 32 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ntg_cap=$n', {k: d[k] for k in ('value','prefill_tok_s','prefill_cold_tok_s')})" ) >> $O/bench3.txt
done
cat $O/pytest.txt $O/bench3.txt
