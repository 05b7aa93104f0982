# round 3: Q4_1 / Q5_0 / Q5_1 in every graph (llama, falcon, gpt2, starcoder, mpt) on the GPU against oracle/_ref, decode speed per ftype
cd /root/repo
O=gpurun_out/r3raw2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "Q4_1 or Q5_0 or Q5_1 or F16" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python tools/scratch/raw32_speed.py > $O/speed.txt 2>&1; cat $O/speed.txt
