cd /root/repo
O=gpurun_out/r3Q; rm -rf $O; mkdir -p $O
for v in 0 1; do
  echo "## CT_AMD_ATTN_NWV4=$v" >> $O/ab.txt
  CT_AMD_ATTN_NWV4=$v timeout 300 python tools/ctx_scaling.py llama-7b-2l 2>/dev/null | cut -c1-110 >> $O/ab.txt
  CT_AMD_ATTN_NWV4=$v timeout 300 python tools/ctx_scaling.py 2>/dev/null | cut -c1-110 >> $O/ab.txt
done
cat $O/ab.txt
