# round 3 (second session), run B: the new GPU cases — Falcon-7B widths (rows of 142 32-blocks), MPT-30B heads (112)
cd /root/repo
O=gpurun_out/r3B; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "falcon-7b-2l or mpt-30b-2l" -p no:cacheprovider > $O/pytest_new_shapes.log 2>&1; echo "rc=$?" >> $O/pytest_new_shapes.log
tail -5 $O/pytest_new_shapes.log
timeout 300 python tools/legacy_speed.py > $O/legacy_speed.txt 2>&1; tail -4 $O/legacy_speed.txt
