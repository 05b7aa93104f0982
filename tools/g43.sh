cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/g43; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -k "pipeline or stage" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for S in 2 4 8; do
  D=$(python -c "print(','.join(['0'] * $S))")
  for SH in 1 0; do
  CT_AMD_PP_SHARED_STREAM=$SH CTAMD_BENCH_DEVICES=$D timeout 600 python bench.py --gpus $S --steps 64 --no-cpu-baseline --no-other-configs --no-long-context 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('stages $S shared $SH', d['value'], d['prefill_tok_s'], d['config'].get('handoff'))"
  done
done 2>&1 | tee $O/pp.txt
timeout 300 python bench.py --steps 64 --no-cpu-baseline --no-other-configs --no-long-context 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('one stage', d['value'], d['prefill_tok_s'])" | tee -a $O/pp.txt
