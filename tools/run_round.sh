cd /root/repo
for i in 1 2; do
timeout 120 python tools/gpu_sites.py base SITES_LIB=/root/repo/ctransformers_amd/lib/libbase.so 2>/dev/null | cut -c1-330
timeout 120 python tools/gpu_sites.py new 2>/dev/null | cut -c1-330
done
