cd /root/repo
O=gpurun_out/r3o; rm -rf $O; mkdir -p $O
for v in lib_e20 lib lib_e20 lib; do
timeout 300 python tools/gpu_sites.py $v SITES_LIB=/root/repo/ctransformers_amd/$v/libctransformers.so > $O/sites_$v.json 2> $O/sites_$v.err; cat $O/sites_$v.json
done
