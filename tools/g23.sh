bash tools/pmc_mfma.sh gpurun_out/pmc_mfma 2>&1 | tail -45
timeout 900 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; tail -3 gpurun_out/bench_quick.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_quick.json") if l.startswith("{")][-1])
print(d["value"], d["prefill_tok_s"], d.get("prefill_2k_tok_s"), d.get("decode_tok_s_at_2k"), d["prefill"].get("frac"), d["data"][:160])
PY
