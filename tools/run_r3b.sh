# round 3: generation 9 prologue / ring-prefetch variants (separate builds), per-site timings and in-kernel traces
cd /root/repo
O=gpurun_out/r3b; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for v in lib lib_p2pre4 lib_pre4; do
  timeout 300 python tools/gpu_sites.py $v SITES_LIB=/root/repo/ctransformers_amd/$v/libctransformers.so > $O/sites_$v.json 2> $O/sites_$v.err; cat $O/sites_$v.json
done
for v in lib lib_p2pre4; do
  SITES_LIB=/root/repo/ctransformers_amd/$v/libctransformers.so timeout 300 python tools/gpu_trace.py > $O/trace_$v.txt 2> $O/trace_$v.err; echo "== $v"; grep -A3 -E "^qkv|^wo|^down" $O/trace_$v.txt
done
