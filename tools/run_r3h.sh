# round 3: decode attention generation 9 (workgroup per head x 16 channels, cursor by scalar load, speculative K / V requests)
cd /root/repo
O=gpurun_out/r3i; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 300 python tools/gpu_sites.py lib > $O/sites_lib.json 2> $O/sites_lib.err; cat $O/sites_lib.json

timeout 300 python tools/gpu_trace.py > $O/trace_lib.txt 2> $O/trace_lib.err; grep -A4 -E "^attn" $O/trace_lib.txt
timeout 600 python tools/ctx_scaling.py > $O/ctx_scaling.txt 2>&1; cat $O/ctx_scaling.txt | tail -12
