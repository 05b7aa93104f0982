# round 2, closing measurement run (trimmed form of run_r2n.sh for the GPU minutes left): GPU suite, smoke, default bench with
# cpu_baseline, rocprofv3 stats of the same command, config 3, the legacy architectures at real sizes.
cd /root/repo
O=gpurun_out/r2z; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_1.json 2> $O/bench_1.err; tail -2 $O/bench_1.err
timeout 900 python bench.py --config 3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
cd /tmp && export TMPDIR=/tmp
CT_AMD_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof -o v7 -- python /root/repo/bench.py --no-cpu-baseline --steps 64 > /root/repo/$O/prof.log 2>&1
cd /root/repo
python tools/prof_summary.py $O/prof > $O/kernel_stats.txt 2>&1
timeout 400 python tools/legacy_speed.py > $O/legacy_speed.txt 2>&1
python - <<'PY'
import json
for n in ("bench_1", "bench_cfg3"):
    try:
        d = json.loads([l for l in open("gpurun_out/r2z/%s.json" % n) if l.startswith("{")][-1])
        print(n, d["value"], "tok/s prefill", d["prefill_tok_s"], "load", d["load_s"], "frac", (d.get("roofline") or {}).get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(n, "failed", e)
PY
head -16 $O/kernel_stats.txt; cat $O/legacy_speed.txt
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
