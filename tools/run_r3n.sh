cd /root/repo
O=gpurun_out/r3n; rm -rf $O; mkdir -p $O
timeout 300 python tools/gpu_sites.py lib > $O/sites_lib.json 2> $O/sites_lib.err; cat $O/sites_lib.json
ATTN_CTX=512 timeout 300 python tools/attn_trace_ctx.py llama-7b-2l > $O/attn_trace_7b.txt 2>&1; head -12 $O/attn_trace_7b.txt
