"""Decode speed of full-size Llama-2-7B files of ftype Q4_1 / Q5_0 / Q5_1 (kernels_raw32.h: file layout, two launches per site)."""
import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctransformers_amd import synth
from ctransformers_amd.llm import LLM, Config
for ft in sys.argv[1:] or ["Q4_1", "Q5_0", "Q5_1"]:
    p = "/tmp/%s_7b.gguf" % ft
    if not os.path.exists(p): synth.write_llama_gguf(p, "llama-2-7b", ft, seed=1)
    gb = os.path.getsize(p) / 1e9
    t0 = time.perf_counter(); m = LLM(p, config=Config(context_length=256, batch_size=8)); tl = time.perf_counter() - t0
    m.eval(synth.prompt_tokens(8, m.vocab_size)); tok = m.sample(top_k=1, repetition_penalty=1.0)
    for _ in range(4): m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
    t0 = time.perf_counter()
    for _ in range(32): m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
    dt = (time.perf_counter() - t0) / 32
    print("Llama-2-7B %s (%.2f GB): load %.1f s, decode %.1f tok/s (%.2f ms/token, %.2f TB/s)" % (ft, gb, tl, 1 / dt, dt * 1e3, gb * 1e9 / dt / 1e12), flush=True)
    del m
    os.remove(p)
