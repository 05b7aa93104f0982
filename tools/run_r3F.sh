cd /root/repo
O=gpurun_out/r3F; rm -rf $O; mkdir -p $O
for cfg in "7 2" "7 4" "5 2" "5 4" "4 2" "4 4" "3 4" "7 2"; do
  set -- $cfg
  TAG="nwv$1_pb$2" CT_AMD_ATTN_NWV=$1 CT_AMD_ATTN_PB=$2 python tools/scratch/attn_ab.py 192 2>/dev/null | tail -1 >> $O/attn_ab.txt
done
cat $O/attn_ab.txt
