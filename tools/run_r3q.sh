cd /root/repo
O=gpurun_out/r3q; rm -rf $O; mkdir -p $O
for v in lib lib_pre1 lib_pre3 lib_pre4 lib; do
timeout 300 python tools/gpu_sites.py $v SITES_LIB=/root/repo/ctransformers_amd/$v/libctransformers.so > $O/sites_$v.json 2> $O/sites_$v.err; cat $O/sites_$v.json
done
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3q/bench.json") if l.startswith("{")][-1])
print("bench", d["value"], "tok/s prefill", d["prefill_tok_s"], "load", d["load_s"], "frac", (d.get("roofline") or {}).get("frac"))
PY
