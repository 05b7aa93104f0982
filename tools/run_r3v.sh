cd /root/repo
O=gpurun_out/r3v; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 900 python bench.py --config 3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; tail -2 $O/bench_cfg3.err; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3v/bench_cfg3.json") if l.startswith("{")][-1])
print("cfg3", d["value"], "tok/s prefill", d["prefill_tok_s"], "load", d["load_s"], (d.get("roofline") or {}).get("sites"))
PY
timeout 300 python tools/gpu_sites.py lib > $O/sites_lib.json 2> $O/sites_lib.err; cat $O/sites_lib.json
timeout 300 python tools/legacy_speed.py > $O/legacy.txt 2>&1; tail -4 $O/legacy.txt
