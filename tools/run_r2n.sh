# round 2, final measurement run: GPU suite, default bench (with cpu_baseline), rocprofv3 stats of the same command, PMC traffic
# passes, per-site trace of a 128-token prompt, in-kernel step trace, bench in its multi-stage forms, configs 3 / 4 / 5
cd /root/repo
O=gpurun_out/r2n; rm -rf $O; mkdir -p $O
M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 900 python bench.py > $O/bench_1.json 2> $O/bench_1.err; tail -2 $O/bench_1.err
CTAMD_BENCH_DEVICES=0,0 timeout 600 python bench.py --gpus 2 --steps 64 --no-cpu-baseline > $O/bench_2_inproc.json 2> $O/bench_2_inproc.err
CTAMD_BENCH_DEVICES=0,0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 64 --no-cpu-baseline > $O/bench_2_torchrun.json 2> $O/bench_2_torchrun.err
timeout 900 python bench.py --config 3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
cd /tmp && export TMPDIR=/tmp
CT_AMD_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof -o v7 -- python /root/repo/bench.py --no-cpu-baseline --steps 64 > /root/repo/$O/prof.log 2>&1
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /root/repo/$O/pmc_fetch -o v7 -- python /root/repo/tools/decode_loop.py --model $M --prompt 8 --decode 8 > /root/repo/$O/pmc_fetch.log 2>&1
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /root/repo/$O/pmc_write -o v7 -- python /root/repo/tools/decode_loop.py --model $M --prompt 8 --decode 8 > /root/repo/$O/pmc_write.log 2>&1
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/prof_prefill -o pf -- python /root/repo/tools/decode_loop.py --model $M --prompt 128 --decode 2 > /root/repo/$O/prof_prefill.log 2>&1
cd /root/repo
python tools/prof_summary.py $O/prof > $O/kernel_stats.txt 2>&1
python tools/pmc_traffic.py $O/pmc_fetch/v7_counter_collection.csv $O/pmc_write/v7_counter_collection.csv > $O/pmc_traffic.json 2>&1
python tools/pf_sites.py $O/prof_prefill > $O/prefill_sites.txt 2>&1
for s in gate_up down attn; do CT_AMD_PG_TRACE=$s CT_AMD_GRAPH=0 timeout 300 python tools/decode_loop.py --model $M --prompt 128 --decode 1 2>&1 | grep -E "pg_trace|attn_trace" | sed -n 3,3p | cut -c1-1800 >> $O/pg_step_trace.txt; done
timeout 300 python tools/prefill_sweep.py $M 8 16 32 64 128 > $O/prefill_sweep.txt 2>&1
timeout 1500 python bench.py --config 4 --no-cpu-baseline --steps 64 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
rm -f /tmp/ctamd_falcon_40b_q4km_r2.gguf
timeout 1800 python bench.py --config 5 --no-cpu-baseline --steps 64 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python - <<'PY'
import json
for n in ("bench_1", "bench_2_inproc", "bench_2_torchrun", "bench_cfg3", "bench_cfg4", "bench_cfg5"):
    try:
        d = json.loads([l for l in open("gpurun_out/r2n/%s.json" % n) if l.startswith("{")][-1])
        print(n, d["value"], "tok/s prefill", d["prefill_tok_s"], d["config"]["parallelism"], "load", d["load_s"], "frac", (d.get("roofline") or {}).get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(n, "failed", e)
PY
head -14 $O/kernel_stats.txt; head -12 $O/prefill_sites.txt; cat $O/prefill_sweep.txt; head -c 600 $O/pmc_traffic.json
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
