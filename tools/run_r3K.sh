cd /root/repo
O=gpurun_out/r3K; rm -rf $O; mkdir -p $O
M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
for site in gate_up qkv; do
  CT_AMD_GRAPH=0 CT_AMD_PG_TRACE=$site timeout 300 python tools/decode_loop.py --model $M --shape llama-2-7b --prompt 128 --decode 1 2>&1 | grep pg_trace | sed -n '3p;40p' > $O/trace_$site.txt
  cat $O/trace_$site.txt | cut -c1-1500
done
