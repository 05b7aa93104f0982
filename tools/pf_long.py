"""Prompt rate of a long prompt at a given context (which decides how many probability rows attn_chunk_long_kernel keeps in LDS: 16 up to 2048, 8 above)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth
from ctransformers_amd.llm import LLM, Config
shape = sys.argv[1] if len(sys.argv) > 1 else "llama-7b-2l"
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
p = "/tmp/%s.gguf" % shape
if not os.path.exists(p):
    synth.write_llama_gguf(p, shape, "Q5_K_M" if "70b" in shape else "Q4_K_M", seed=5)
m = LLM(p, config=Config(context_length=ctx, batch_size=128))
toks = synth.prompt_tokens(n, m.vocab_size)
best = 1e9
for rep in range(3):
    m._context = []
    t0 = time.perf_counter(); m.eval(toks); _ = m.logits[0]; dt = time.perf_counter() - t0
    best = min(best, dt)
print("%s ctx %d prompt %d: %.1f tok/s (%.2f ms)" % (shape, ctx, n, n / best, best * 1e3))
