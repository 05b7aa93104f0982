# round 2: generation 8 on hardware — parity suite, bench (v8 / v7 A-B), in-kernel trace, rocprof stats
cd /root/repo
O=gpurun_out/r2c; rm -rf $O; mkdir -p $O
export CTAMD_BENCH_MODEL=/tmp/ctamd_llama2_7b_q4km_r2.gguf
python -c "
from ctransformers_amd import synth
synth.write_llama_gguf('$CTAMD_BENCH_MODEL','llama-2-7b','Q4_K_M',seed=1234)" > $O/gen.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_v8.json 2> $O/bench_v8.err
cat $O/bench_v8.json
CT_AMD_V8=0 timeout 600 python bench.py --no-cpu-baseline --steps 64 > $O/bench_v7.json 2> $O/bench_v7.err
python - <<'PY'
import json
for n in ("v8", "v7"):
    try:
        d = json.load(open("gpurun_out/r2c/bench_%s.json" % n))
        print(n, d["value"], {k: v["us"] for k, v in d["roofline"]["sites"].items()})
    except Exception as e:
        print(n, "failed", e)
PY
timeout 300 python tools/gpu_trace.py > $O/trace_v8.txt 2>&1
cat $O/trace_v8.txt
cd /tmp && export TMPDIR=/tmp
CT_AMD_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof -o v8 -- python /root/repo/bench.py --no-cpu-baseline --steps 64 > /root/repo/$O/prof.log 2>&1
cd /root/repo
python tools/prof_summary.py $O/prof > $O/kernel_stats.txt 2>&1
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
head -12 $O/kernel_stats.txt
