# round 3 (second session), run A: the new GPU cases (Falcon-7B widths), full-size parity of configs 5 / 4, a 2048-context prompt at the 70B widths
cd /root/repo
O=gpurun_out/r3A; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "falcon-7b-2l" -p no:cacheprovider > $O/pytest_falcon7b.log 2>&1; echo "rc=$?" >> $O/pytest_falcon7b.log
tail -3 $O/pytest_falcon7b.log
# config 5 first (the 49 GB file also serves the 2k-context prompt), then config 4
CTAMD_BENCH_BIG=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_big_config_full_size and 5" -p no:cacheprovider -n 0 > $O/pytest_big5.log 2>&1; echo "rc=$?" >> $O/pytest_big5.log
tail -3 $O/pytest_big5.log
timeout 600 python tools/prefill_2k.py /tmp/ctamd_llama2_70b_q5km_r2.gguf 2048 128 llama-2-70b Q5_K_M > $O/prefill_2k_70b.txt 2>&1
tail -5 $O/prefill_2k_70b.txt
rm -f /tmp/ctamd_llama2_70b_q5km_r2.gguf
CTAMD_BENCH_BIG=1 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_big_config_full_size and 4" -p no:cacheprovider -n 0 > $O/pytest_big4.log 2>&1; echo "rc=$?" >> $O/pytest_big4.log
tail -3 $O/pytest_big4.log
rm -f /tmp/ctamd_falcon_40b_q4km_r2.gguf
