"""A/B helper: decode tok/s of the 7B bench model at ctx 512 (128-token prompt, N greedy steps) + the attention site's sweep time."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ctransformers_amd import synth, measure
from ctransformers_amd.llm import LLM, Config
p = "/tmp/ctamd_llama2_7b_q4km_r2.gguf"
if not os.path.exists(p):
    synth.write_llama_gguf(p, "llama-2-7b", "Q4_K_M", seed=1234)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 192
m = LLM(p, config=Config(context_length=512, batch_size=128))
m.eval(synth.prompt_tokens(128, m.vocab_size))
tok = m.sample(top_k=1, repetition_penalty=1.0)
for _ in range(16): m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
t0 = time.perf_counter()
for _ in range(N): m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
dt = (time.perf_counter() - t0) / N
sites = measure.profile_sites(m._lib, m._llm, 4)
d = {s["site"]: round(s["ms"] * 1e3 / s["launches"], 2) for s in sites if s["site"].endswith("@sweep")}
print(json.dumps(dict(tag=os.environ.get("TAG", ""), tok_s=round(1 / dt, 1), ms=round(dt * 1e3, 4), sweep_us=d)), flush=True)
