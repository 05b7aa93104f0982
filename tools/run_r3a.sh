# round 3, first GPU run of generation 9: GPU suite, per-site timings gen 9 vs gen 7, in-kernel trace, bench
cd /root/repo
O=gpurun_out/r3a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python tools/gpu_sites.py gen9 > $O/sites_gen9.json 2> $O/sites_gen9.err; cat $O/sites_gen9.json
timeout 300 python tools/gpu_sites.py gen7 CT_AMD_MATVEC_GEN=7 > $O/sites_gen7.json 2> $O/sites_gen7.err; cat $O/sites_gen7.json
timeout 300 python tools/gpu_trace.py > $O/trace_gen9.txt 2> $O/trace_gen9.err; cat $O/trace_gen9.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3a/bench.json") if l.startswith("{")][-1])
print("bench", d["value"], "tok/s prefill", d["prefill_tok_s"], "load", d["load_s"], "frac", (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("sites"))
PY
