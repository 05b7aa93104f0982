cd /root/repo
O=gpurun_out/r3O; rm -rf $O; mkdir -p $O
python tools/prefill_sweep.py /tmp/ctamd_llama2_7b_q80_r2.gguf:llama-2-7b:Q8_0 128 > $O/sweep_q80.txt 2>&1; tail -1 $O/sweep_q80.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_wide_rows.py tests/test_mpt.py tests/test_starcoder.py -m gpu -q -x -k "Q8_0 or Q4_0 or q80 or q40 or wide or mpt or starcoder or gpt2 or config3" -p no:cacheprovider > $O/pytest.log 2>&1; tail -2 $O/pytest.log
python tools/legacy_speed.py > $O/legacy.txt 2>&1; tail -2 $O/legacy.txt
