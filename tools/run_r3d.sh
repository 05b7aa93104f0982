# round 3: generation 9 with early norm weights everywhere, live-wave rendezvous in the prologue
cd /root/repo
O=gpurun_out/r3d; rm -rf $O; mkdir -p $O
timeout 300 python tools/gpu_sites.py lib > $O/sites_lib.json 2> $O/sites_lib.err; cat $O/sites_lib.json
timeout 300 python tools/gpu_trace.py > $O/trace_lib.txt 2> $O/trace_lib.err; grep -A5 -E "^qkv|^wo|^down|^gate" $O/trace_lib.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
