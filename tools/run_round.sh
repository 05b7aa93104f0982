cd /root/repo
timeout 40 tools/experiments/overlap_probe 2000
