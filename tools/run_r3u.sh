# round 3: GPU suite on the multi-GPU readiness changes, bench in its two-stage (one GPU) forms, bench default with other_configs
cd /root/repo
O=gpurun_out/r3u; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
CTAMD_BENCH_DEVICES=0,0 timeout 600 python bench.py --gpus 2 --steps 64 --no-cpu-baseline > $O/bench_2_inproc.json 2> $O/bench_2_inproc.err
timeout 900 python bench.py --no-cpu-baseline > $O/bench_1.json 2> $O/bench_1.err; tail -2 $O/bench_1.err
python - <<'PY'
import json
for n in ("bench_1", "bench_2_inproc"):
    try:
        d = json.loads([l for l in open("gpurun_out/r3u/%s.json" % n) if l.startswith("{")][-1])
        print(n, d["value"], "tok/s prefill", d["prefill_tok_s"], d["config"]["parallelism"], "load", d["load_s"], "frac", (d.get("roofline") or {}).get("frac"), "other", d.get("other_configs"))
    except Exception as e:
        print(n, "failed", e)
PY
