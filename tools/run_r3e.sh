# round 3: generation 9 with byte scales (E20 records), Q6_K offset through the image, v_fmac_f32_dpp chain
cd /root/repo
O=gpurun_out/r3e; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 300 python tools/gpu_sites.py lib > $O/sites_lib.json 2> $O/sites_lib.err; cat $O/sites_lib.json
timeout 300 python tools/gpu_trace.py > $O/trace_lib.txt 2> $O/trace_lib.err; grep -A5 -E "^qkv|^wo|^down|^gate" $O/trace_lib.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3e/bench.json") if l.startswith("{")][-1])
print("bench", d["value"], "tok/s prefill", d["prefill_tok_s"], "load", d["load_s"], "frac", (d.get("roofline") or {}).get("frac"))
PY
