cd /root/repo
export CTAMD_BENCH_MODEL=/tmp/l7b.gguf
timeout 600 python bench.py --no-cpu-baseline
