# round 2: prompt chunks on the f16 matrix cores (kernels_pg.h): parity, bench, per-site trace of a 128-token prompt
cd /root/repo
O=gpurun_out/r2f; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_1.json 2> $O/bench_1.err; tail -2 $O/bench_1.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2f/bench_1.json") if l.startswith("{")][-1])
print("decode", d["value"], "prefill", d["prefill_tok_s"], "load", d["load_s"])
PY
M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
cd /tmp && export TMPDIR=/tmp
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/prof_prefill -o pf -- python /root/repo/tools/decode_loop.py --model $M --prompt 128 --decode 2 > /root/repo/$O/prof_prefill.log 2>&1
cd /root/repo
python tools/pf_sites.py $O/prof_prefill > $O/prefill_sites.txt 2>&1
head -24 $O/prefill_sites.txt
timeout 300 python tools/prefill_sweep.py $M 8 16 32 64 128 > $O/prefill_sweep.txt 2>&1; cat $O/prefill_sweep.txt
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
