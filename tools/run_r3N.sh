cd /root/repo
O=gpurun_out/r3N; rm -rf $O; mkdir -p $O
TAG=shallow python tools/scratch/attn_ab.py 192 2>/dev/null | tail -1 >> $O/ab.txt
TAG=deep CT_AMD_ATTN_DEEP_CTX=0 python tools/scratch/attn_ab.py 192 2>/dev/null | tail -1 >> $O/ab.txt
TAG=shallow python tools/scratch/attn_ab.py 192 2>/dev/null | tail -1 >> $O/ab.txt
TAG=deep CT_AMD_ATTN_DEEP_CTX=0 python tools/scratch/attn_ab.py 192 2>/dev/null | tail -1 >> $O/ab.txt
cat $O/ab.txt
