# round 3 (second session), closing measurement run: GPU suite, default bench (with cpu_baseline and other_configs), rocprofv3 stats of the same command, PMC traffic
# passes, SQ counters, in-kernel traces, per-site prefill timings, attention context scaling, bench in its two-stage forms, configs 4 / 5
cd /root/repo
O=gpurun_out/r3Z2; rm -rf $O; mkdir -p $O
M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 900 python bench.py > $O/bench_1.json 2> $O/bench_1.err; tail -2 $O/bench_1.err
CTAMD_BENCH_DEVICES=0,0 timeout 600 python bench.py --gpus 2 --steps 64 --no-cpu-baseline > $O/bench_2_inproc.json 2> $O/bench_2_inproc.err
CTAMD_BENCH_DEVICES=0,0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 64 --no-cpu-baseline > $O/bench_2_torchrun.json 2> $O/bench_2_torchrun.err
cd /tmp && export TMPDIR=/tmp
CT_AMD_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof -o v9 -- python /root/repo/bench.py --no-cpu-baseline --no-other-configs --steps 64 > /root/repo/$O/prof.log 2>&1
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /root/repo/$O/pmc_fetch -o v9 -- python /root/repo/tools/decode_loop.py --model $M --prompt 8 --decode 8 > /root/repo/$O/pmc_fetch.log 2>&1
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /root/repo/$O/pmc_write -o v9 -- python /root/repo/tools/decode_loop.py --model $M --prompt 8 --decode 8 > /root/repo/$O/pmc_write.log 2>&1
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d /root/repo/$O/pmc_sq -o p -- python /root/repo/tools/decode_loop.py --model $M --prompt 8 --decode 6 > /root/repo/$O/pmc_sq.log 2>&1
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/prof_prefill -o pf -- python /root/repo/tools/decode_loop.py --model $M --prompt 128 --decode 2 > /root/repo/$O/prof_prefill.log 2>&1
cd /root/repo
python tools/prof_summary.py $O/prof > $O/kernel_stats.txt 2>&1
python tools/pmc_traffic.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) > $O/pmc_traffic.json 2>&1
f=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1); python tools/pmc_sq.py $f matvec_v9 > $O/sq_counters.txt 2>&1
python tools/pf_sites.py $O/prof_prefill > $O/prefill_sites.txt 2>&1
timeout 300 python tools/gpu_sites.py final > $O/sites.json 2> $O/sites.err
timeout 300 python tools/gpu_trace.py > $O/trace.txt 2> $O/trace.err
timeout 300 python tools/ctx_scaling.py > $O/ctx_scaling_70b.txt 2>&1
timeout 300 python tools/ctx_scaling.py llama-7b-2l > $O/ctx_scaling_7b.txt 2>&1
timeout 300 python tools/attn_trace_ctx.py > $O/attn_trace_70b.txt 2>&1
timeout 300 python tools/prefill_sweep.py $M 8 16 32 64 128 > $O/prefill_sweep.txt 2>&1
timeout 300 python tools/legacy_speed.py > $O/legacy_speed.txt 2>&1
timeout 1500 python bench.py --config 4 --no-cpu-baseline --steps 64 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
rm -f /tmp/ctamd_falcon_40b_q4km_r2.gguf
timeout 1800 python bench.py --config 5 --no-cpu-baseline --steps 64 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python - <<'PY'
import json
for n in ("bench_1", "bench_2_inproc", "bench_2_torchrun", "bench_cfg4", "bench_cfg5"):
    try:
        d = json.loads([l for l in open("gpurun_out/r3Z2/%s.json" % n) if l.startswith("{")][-1])
        print(n, d["value"], "tok/s prefill", d["prefill_tok_s"], d["config"]["parallelism"], "load", d["load_s"], "frac", (d.get("roofline") or {}).get("frac"), "tokfrac", d["token_roofline"]["frac_of_8TBps"], "cpu", (d.get("cpu_baseline") or {}).get("value"), "other", [(o.get("config"), o.get("decode_tok_s"), o.get("prefill_tok_s")) for o in d.get("other_configs") or []])
    except Exception as e:
        print(n, "failed", e)
PY
head -14 $O/kernel_stats.txt; head -12 $O/prefill_sites.txt; cat $O/prefill_sweep.txt; head -c 700 $O/pmc_traffic.json; cat $O/ctx_scaling_70b.txt $O/ctx_scaling_7b.txt
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
