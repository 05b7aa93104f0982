# A/B: chunk kernel with 8 waves per workgroup (one workgroup per CU) against 4 (two per CU)
cd /root/repo
O=gpurun_out/r3E; rm -rf $O; mkdir -p $O
M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
python tools/prefill_sweep.py $M 32 128 > $O/sweep_nw8.txt 2>&1; tail -2 $O/sweep_nw8.txt
CTRANSFORMERS_AMD_LIB=/root/repo/ctransformers_amd/lib_ab4/libctransformers.so python tools/prefill_sweep.py $M 32 128 > $O/sweep_nw4.txt 2>&1; tail -2 $O/sweep_nw4.txt
CTRANSFORMERS_AMD_LIB=/root/repo/ctransformers_amd/lib_ab4/libctransformers.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_config2_full_size or test_chunk_path_repeatable or llama-70b-2l or falcon-40b-2l or falcon-small" -p no:cacheprovider > $O/pytest_nw4.log 2>&1; tail -2 $O/pytest_nw4.log
