cd /root/repo
O=gpurun_out/r3t; rm -rf $O; mkdir -p $O
timeout 900 python bench.py --config 3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; tail -2 $O/bench_cfg3.err; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3t/bench_cfg3.json") if l.startswith("{")][-1])
print("cfg3", d["value"], "tok/s prefill", d["prefill_tok_s"], "load", d["load_s"], (d.get("roofline") or {}).get("sites"))
PY
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "q8_0 or q80 or q40 or gpt2 or config3" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
