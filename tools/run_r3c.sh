# round 3: activation requests fenced from the weight stream by a barrier (lib), + whole ring before the prologue (lib_xb4), no barrier (lib_nob)
cd /root/repo
O=gpurun_out/r3c; rm -rf $O; mkdir -p $O
for v in lib lib_xb4 lib_nob; do
  timeout 300 python tools/gpu_sites.py $v SITES_LIB=/root/repo/ctransformers_amd/$v/libctransformers.so > $O/sites_$v.json 2> $O/sites_$v.err; cat $O/sites_$v.json
done
for v in lib lib_xb4; do
  SITES_LIB=/root/repo/ctransformers_amd/$v/libctransformers.so timeout 300 python tools/gpu_trace.py > $O/trace_$v.txt 2> $O/trace_$v.err; echo "== $v"; grep -A5 -E "^qkv|^wo|^down|^gate" $O/trace_$v.txt
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
