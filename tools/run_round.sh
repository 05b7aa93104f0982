cd /root/repo
python tools/gpu_sites.py "$1" 2>/dev/null
python tools/gpu_trace.py 2>/dev/null | grep -A4 -E "^(gate_up)"
