cd /root/repo
timeout 200 python tools/host_overhead.py 2>/dev/null
