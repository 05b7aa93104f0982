cd /root/repo
for i in 1 2; do
python tools/gpu_sites.py base SITES_LIB=/root/repo/ctransformers_amd/lib/libbase.so 2>/dev/null
python tools/gpu_sites.py new 2>/dev/null
done
