"""Where the per-token wall time goes on the host side (decode loop of bench.py): eval() vs sample()."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth
from ctransformers_amd.llm import LLM, Config
p = os.environ.get("CTAMD_BENCH_MODEL", "/tmp/ctamd_llama2_7b_q4km_r2.gguf")
if not os.path.exists(p):
    synth.write_llama_gguf(p, "llama-2-7b", "Q4_K_M", seed=1234)
m = LLM(p, config=Config(context_length=512, batch_size=128))
m.eval(synth.prompt_tokens(128, 32000))
tok = m.sample(top_k=1, repetition_penalty=1.0)
for _ in range(8):
    m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
te = ts = 0.0
N = 128
for _ in range(N):
    t0 = time.perf_counter(); m.eval([tok]); t1 = time.perf_counter(); tok = m.sample(top_k=1, repetition_penalty=1.0); t2 = time.perf_counter()
    te += t1 - t0; ts += t2 - t1
# prefill-style chunk: 64 tokens in one eval (no per-token host work)
ctx0 = list(m._context)
t0 = time.perf_counter(); m.eval(synth.prompt_tokens(64, 32000)); tc = (time.perf_counter() - t0) / 64
print(json.dumps(dict(eval_us=round(te / N * 1e6, 1), sample_us=round(ts / N * 1e6, 1), chunked_eval_us_per_token=round(tc * 1e6, 1))))
