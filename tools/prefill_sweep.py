"""Prompt throughput of the HIP library against the ABI's batch_size (the reference's default is 8): tokens/s of a 256-token
prompt evaluated in chunks of that size, third pass (steady state).  usage: prefill_sweep.py <model.gguf>[:shape:ftype] [batch sizes...]   (shape/ftype: generate the synthetic file if missing)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth  # noqa: E402
from ctransformers_amd.llm import LLM, Config  # noqa: E402

path, shape, ftype = (sys.argv[1].split(":") + ["llama-2-7b", "Q4_K_M"])[:3]
if not os.path.exists(path):
    (synth.write_falcon_gguf if shape.startswith("falcon") else synth.write_llama_gguf)(path, shape, ftype, seed=1234)
sizes = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8, 16, 32, 64, 128]
N = 256
for bs in sizes:
    m = LLM(path, config=Config(context_length=512, batch_size=bs, gpu_layers=1000))
    toks = synth.prompt_tokens(N, m.vocab_size)
    m.eval(toks)
    m._context = []
    m.eval(toks)          # chunk shapes are captured into hipGraphs on their second use
    m._context = []
    t0 = time.perf_counter()
    m.eval(toks)
    dt = time.perf_counter() - t0
    print("batch_size %4d: %8.1f tok/s  (%.2f ms per chunk)" % (bs, N / dt, dt / ((N + bs - 1) // bs) * 1e3), flush=True)
    del m
