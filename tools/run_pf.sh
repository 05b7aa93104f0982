cd /root/repo
O=gpurun_out/pf10; rm -rf $O; mkdir -p $O
export CTAMD_BENCH_MODEL=/tmp/l7b.gguf
timeout 1200 python -m pytest tests -x -q -m gpu --durations=6 > $O/pytest.log 2>&1
tail -12 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench_pf.json 2> $O/bench_pf.err; python -c "import json,sys; d=json.load(open(sys.argv[1])); print(\"prefill\", d[\"prefill_tok_s\"], \"cold\", d[\"prefill_cold_tok_s\"], \"decode\", d[\"value\"])" $O/bench_pf.json
timeout 300 python tools/prefill_sweep.py /tmp/l7b.gguf 8 16 64
