cd /root/repo
O=gpurun_out/r3f; rm -rf $O; mkdir -p $O
for v in lib_fake lib; do
timeout 300 python tools/gpu_sites.py $v SITES_LIB=/root/repo/ctransformers_amd/$v/libctransformers.so > $O/sites_$v.json 2> $O/sites_$v.err; cat $O/sites_$v.json
done
SITES_LIB=/root/repo/ctransformers_amd/lib_fake/libctransformers.so timeout 300 python tools/gpu_trace.py > $O/trace_fake.txt 2> $O/trace_fake.err; grep -A5 -E "^qkv|^wo|^down|^gate" $O/trace_fake.txt
