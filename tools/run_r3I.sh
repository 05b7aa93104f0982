cd /root/repo
O=gpurun_out/r3I; rm -rf $O; mkdir -p $O
M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
python tools/prefill_sweep.py $M 128 > $O/sweep.txt 2>&1; tail -1 $O/sweep.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_config2_full_size or test_chunk_path_repeatable or llama-70b-2l or falcon-40b-2l or falcon-small or llama-7b-2l or tiny" -p no:cacheprovider > $O/pytest.log 2>&1; tail -2 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/prof_prefill -o pf -- python /root/repo/tools/decode_loop.py --model $M --prompt 128 --decode 2 > /root/repo/$O/prof_prefill.log 2>&1
cd /root/repo
python tools/pf_sites.py $O/prof_prefill > $O/prefill_sites.txt 2>&1; head -12 $O/prefill_sites.txt
