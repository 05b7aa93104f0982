cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "40b or 70b" 2>&1 | tail -4
timeout 600 python tools/big_models.py falcon-40b 2>&1 | tail -1
rm -f /tmp/falcon-40b_Q4_K_M.gguf
timeout 800 python tools/big_models.py llama-2-70b 2>&1 | tail -1
rm -f /tmp/llama-2-70b_Q5_K_M.gguf
