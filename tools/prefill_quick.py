"""Steady-state prompt rate of a 128-token prompt (third and later passes) for a synthetic model; SITES_LIB picks a variant build.
usage: python tools/prefill_quick.py <tag> [shape] [ftype]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth
from ctransformers_amd.llm import LLM, Config
tag = sys.argv[1] if len(sys.argv) > 1 else ""
shape = sys.argv[2] if len(sys.argv) > 2 else "llama-2-7b"
ftype = sys.argv[3] if len(sys.argv) > 3 else "Q4_K_M"
p = {"Q4_K_M": "/tmp/ctamd_llama2_7b_q4km_r2.gguf", "Q8_0": "/tmp/ctamd_llama2_7b_q80_r2.gguf"}.get(ftype, "/tmp/ctamd_%s_%s.gguf" % (shape, ftype)) if shape == "llama-2-7b" else "/tmp/ctamd_%s_%s.gguf" % (shape, ftype)
if not os.path.exists(p):
    (synth.write_falcon_gguf if shape.startswith("falcon") else synth.write_llama_gguf)(p, shape, ftype, seed=1234)
m = LLM(p, config=Config(context_length=512, batch_size=128), lib=os.environ.get("SITES_LIB") or None)
toks = synth.prompt_tokens(128, m.vocab_size)
for _ in range(3):
    m._context = []
    m.eval(toks)
ts = []
for _ in range(4):
    m._context = []
    t0 = time.perf_counter()
    m.eval(toks)
    ts.append(time.perf_counter() - t0)
print("%s %s %s prefill tok/s %.0f" % (tag, shape, ftype, 128 / min(ts)))
