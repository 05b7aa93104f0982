cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "40b or 70b" 2>&1 | tail -8
