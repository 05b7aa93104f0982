import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctransformers_amd import synth
from ctransformers_amd.llm import LLM, Config
p = "/tmp/f16_7b.gguf"
if not os.path.exists(p): synth.write_llama_gguf(p, "llama-2-7b", "F16", seed=1)
t0 = time.perf_counter(); m = LLM(p, config=Config(context_length=256, batch_size=8)); tl = time.perf_counter() - t0
m.eval(synth.prompt_tokens(8, m.vocab_size)); tok = m.sample(top_k=1, repetition_penalty=1.0)
for _ in range(4): m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
t0 = time.perf_counter()
for _ in range(32): m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
dt = (time.perf_counter() - t0) / 32
print("Llama-2-7B F16 (13.5 GB), CT_AMD_F16_NT=%s: load %.1f s, decode %.1f tok/s (%.2f ms/token, %.2f TB/s)" % (os.environ.get("CT_AMD_F16_NT", "256"), tl, 1 / dt, dt * 1e3, 13.48e9 / dt / 1e12))
