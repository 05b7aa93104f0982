cd /root/repo
O=gpurun_out/r3P; rm -rf $O; mkdir -p $O
M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
OLD=/root/repo/ctransformers_amd/lib_old/libctransformers.so
for i in 1 2; do
  CTRANSFORMERS_AMD_LIB=$OLD python tools/prefill_sweep.py $M 128 2>&1 | tail -1 | sed 's/^/old: /' >> $O/ab.txt
  python tools/prefill_sweep.py $M 128 2>&1 | tail -1 | sed 's/^/new: /' >> $O/ab.txt
done
cat $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_config2_full_size or test_chunk_path_repeatable or llama-70b-2l or falcon-40b-2l or falcon-small or llama-7b-2l or tiny" -p no:cacheprovider > $O/pytest.log 2>&1; tail -2 $O/pytest.log
