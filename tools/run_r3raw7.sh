# round 3: file-layout kernels with the request rings pinned slot by slot (sched_fence): speed of full 7B files, parity subset
cd /root/repo
O=gpurun_out/r3raw7; rm -rf $O; mkdir -p $O
timeout 600 python tools/scratch/raw32_speed.py Q4_1 Q5_0 Q5_1 F16 > $O/speed.txt 2>&1; cat $O/speed.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "Q4_1 or Q5_0 or Q5_1 or F16 or F32" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
