# round 3: file-layout types (F16, Q4_1, Q5_0, Q5_1) in every graph on the GPU against oracle/_ref
cd /root/repo
O=gpurun_out/r3raw5; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "Q4_1 or Q5_0 or Q5_1 or F16 or F32" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
