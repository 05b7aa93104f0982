cd /root/repo
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
timeout 120 python tools/gpu_sites.py base SITES_LIB=/root/repo/ctransformers_amd/lib/libbase.so 2>/dev/null | cut -c1-120
timeout 120 python tools/gpu_sites.py new 2>/dev/null | cut -c1-120
done
