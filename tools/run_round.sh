cd /root/repo
python tools/gpu_sites.py "$1" 2>/dev/null
python tools/gpu_trace.py 2>/dev/null | grep -A1 -E "^(wo|gate_up|down|lm_head|qkv)"
