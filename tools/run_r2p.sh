cd /root/repo
O=gpurun_out/r2p; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
timeout 600 python bench.py --no-cpu-baseline > $O/bench_1.json 2> $O/bench_1.err; tail -2 $O/bench_1.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2p/bench_1.json") if l.startswith("{")][-1])
print("decode", d["value"], "prefill", d["prefill_tok_s"], "load", d["load_s"])
PY
cd /tmp && export TMPDIR=/tmp
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/prof_prefill -o pf -- python /root/repo/tools/decode_loop.py --model $M --prompt 128 --decode 2 > /root/repo/$O/prof_prefill.log 2>&1
cd /root/repo
python tools/pf_sites.py $O/prof_prefill > $O/prefill_sites.txt 2>&1
head -14 $O/prefill_sites.txt
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
