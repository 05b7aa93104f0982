cd /root/repo
timeout 800 python tools/big_models.py llama-2-70b 2>&1 | tail -2
rm -f /tmp/llama-2-70b_Q5_K_M.gguf
