cd /root/repo
export CTAMD_BENCH_MODEL=/tmp/l7b.gguf
python -c "
import sys; sys.path.insert(0,'.')
from ctransformers_amd import synth
synth.write_llama_gguf('/tmp/l7b.gguf','llama-2-7b','Q4_K_M',seed=1234)"
CTAMD_FORCE_PIPELINE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 64 --warmup 8 2>&1 | tail -5 | cut -c1-1200
