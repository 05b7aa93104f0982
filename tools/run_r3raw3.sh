# round 3: matvec_raw32_kernel workgroup size A/B (the prologue is recomputed per workgroup: NT / 16 rows share it), then parity at the default
cd /root/repo
O=gpurun_out/r3raw3; rm -rf $O; mkdir -p $O
for nt in 128 256 512; do echo "CT_AMD_RAW_NT=$nt"; CT_AMD_RAW_NT=$nt timeout 300 python tools/scratch/raw32_speed.py Q4_1 Q5_0; done > $O/nt_ab.txt 2>&1
cat $O/nt_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "Q4_1 or Q5_0 or Q5_1 or F16" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
