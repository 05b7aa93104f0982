cd /root/repo
O=gpurun_out/r3m; rm -rf $O; mkdir -p $O
timeout 300 python tools/attn_trace_ctx.py > $O/attn_trace_70b.txt 2>&1; cat $O/attn_trace_70b.txt

