# round 2: full GPU suite (incl. full-size config 2 / 3, tokenizers+samplers on the HIP build) + default bench
cd /root/repo
O=gpurun_out/r2e; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -16 $O/pytest.log
timeout 900 python bench.py > $O/bench_1.json 2> $O/bench_1.err; tail -2 $O/bench_1.err
python - <<'PY'
import json
for n in ("bench_1",):
    try:
        lines = [l for l in open("gpurun_out/r2e/%s.json" % n) if l.startswith("{")]
        d = json.loads(lines[-1])
        print(n, d["value"], "tok/s prefill", d["prefill_tok_s"], "load", d["load_s"], "cpu", d.get("cpu_baseline"))
    except Exception as e:
        print(n, "failed", e)
PY
