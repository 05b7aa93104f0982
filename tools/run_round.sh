cd /root/repo
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
timeout 120 python tools/gpu_sites.py base384 SITES_PROMPT=384 SITES_LIB=/root/repo/ctransformers_amd/lib/libbase.so 2>/dev/null | cut -c1-200
timeout 120 python tools/gpu_sites.py new384 SITES_PROMPT=384 2>/dev/null | cut -c1-200
done
timeout 120 python tools/gpu_sites.py base64 SITES_LIB=/root/repo/ctransformers_amd/lib/libbase.so 2>/dev/null | cut -c1-200
timeout 120 python tools/gpu_sites.py new64 2>/dev/null | cut -c1-200
