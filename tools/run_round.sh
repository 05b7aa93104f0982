cd /root/repo
SITES_PROMPT=384 python tools/gpu_trace.py 2>/dev/null | grep -A5 "^attn"
python tools/gpu_sites.py p384 SITES_PROMPT=384 2>/dev/null
