cd /root/repo
O=gpurun_out/r3R; rm -rf $O; mkdir -p $O
for i in 1 2; do
TAG=shallow7 python tools/scratch/attn_ab.py 192 2>/dev/null | tail -1 >> $O/ab.txt
TAG=deep7 CT_AMD_ATTN_DEEP_CTX=0 python tools/scratch/attn_ab.py 192 2>/dev/null | tail -1 >> $O/ab.txt
TAG=deep4 CT_AMD_ATTN_DEEP_CTX=0 CT_AMD_ATTN_NWV4=1 python tools/scratch/attn_ab.py 192 2>/dev/null | tail -1 >> $O/ab.txt
done
cut -c1-120 $O/ab.txt
