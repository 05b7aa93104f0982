cd /root/repo
O=gpurun_out/pf9; rm -rf $O; mkdir -p $O
export CTAMD_BENCH_MODEL=/tmp/l7b.gguf
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden and (q4km or q5km) and not falcon or batch_structure or (bit_identical_to_reference and (llama-small-Q4_K_M or llama-tiny or llama-7b-2l-Q4_K_M or llama-70b-2l)) or full_7b or long_context or pipeline" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench_pf.json 2> $O/bench_pf.err; python -c "import json,sys; d=json.load(open(sys.argv[1])); print(\"prefill\", d[\"prefill_tok_s\"], \"cold\", d[\"prefill_cold_tok_s\"], \"decode\", d[\"value\"])" $O/bench_pf.json
CT_AMD_PF_CHUNK=32 timeout 300 python bench.py --no-cpu-baseline > $O/bench_gx2.json 2> $O/bench_gx2.err; python -c "import json,sys; print(\"chunk32 prefill\", json.load(open(sys.argv[1]))[\"prefill_tok_s\"])" $O/bench_gx2.json
cd /tmp && export TMPDIR=/tmp
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/prof -o pf -- python /root/repo/tools/decode_loop.py --model /tmp/l7b.gguf --prompt 128 --decode 2 > /root/repo/$O/prof.log 2>&1
cd /root/repo
python tools/pf_sites.py $O/prof > $O/pf_sites.txt 2>&1
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
cat $O/pf_sites.txt
