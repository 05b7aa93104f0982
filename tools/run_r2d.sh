# round 2: in-process pipeline on hardware (two stages on the one GPU), bench in its three launch forms, PMC traffic pass
cd /root/repo
O=gpurun_out/r2d; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 900 python bench.py > $O/bench_1.json 2> $O/bench_1.err; tail -2 $O/bench_1.err
CTAMD_BENCH_DEVICES=0,0 timeout 600 python bench.py --gpus 2 --steps 64 > $O/bench_2_inproc.json 2> $O/bench_2_inproc.err; tail -2 $O/bench_2_inproc.err
CTAMD_BENCH_DEVICES=0,0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 64 > $O/bench_2_torchrun.json 2> $O/bench_2_torchrun.err; tail -2 $O/bench_2_torchrun.err
python - <<'PY'
import json
for n in ("bench_1", "bench_2_inproc", "bench_2_torchrun"):
    try:
        lines = [l for l in open("gpurun_out/r2d/%s.json" % n) if l.startswith("{")]
        d = json.loads(lines[-1])
        print(n, d["value"], "tok/s prefill", d["prefill_tok_s"], d["config"]["parallelism"], d["config"]["layer_ranges"], "load", d["load_s"], "cached", d["config"]["model_cached"], "cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"))
    except Exception as e:
        print(n, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp
M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /root/repo/$O/pmc_fetch -o v7 -- python /root/repo/tools/decode_loop.py --model $M --prompt 8 --decode 8 > /root/repo/$O/pmc_fetch.log 2>&1
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /root/repo/$O/pmc_write -o v7 -- python /root/repo/tools/decode_loop.py --model $M --prompt 8 --decode 8 > /root/repo/$O/pmc_write.log 2>&1
cd /root/repo
python tools/pmc_traffic.py $O/pmc_fetch/v7_counter_collection.csv $O/pmc_write/v7_counter_collection.csv > $O/pmc_traffic.json 2>&1
head -c 1500 $O/pmc_traffic.json
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
