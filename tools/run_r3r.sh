cd /root/repo
O=gpurun_out/r3r; rm -rf $O; mkdir -p $O
for v in pf1 pf0 pf1 pf0; do
timeout 300 python tools/gpu_sites.py $v CT_AMD_ATTN_PF=${v#pf} > $O/sites_$v.json 2> $O/sites_$v.err; cat $O/sites_$v.json
done
