cd /root/repo
O=gpurun_out/r3D; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_mpt.py tests/test_starcoder.py -m gpu -q -k "falcon-7b-2l or mpt or starcoder or gpt2" -p no:cacheprovider > $O/pytest_legacy.log 2>&1; echo "rc=$?" >> $O/pytest_legacy.log
tail -5 $O/pytest_legacy.log
