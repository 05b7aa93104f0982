"""Per-launch-site decode timings (HIP events inside the library) for the synthetic 7B model. Usage: gpu_sites.py [tag] [ENV=VAL ...]"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for kv in sys.argv[2:]:
    k, v = kv.split("="); os.environ[k] = v
from tools import synth
from ctransformers_amd.llm import LLM, Config

p = os.environ.get("CTAMD_BENCH_MODEL", "/tmp/ctamd_llama2_7b_q4km_r2.gguf")
if not os.path.exists(p):
    synth.write_llama_gguf(p, "llama-2-7b", "Q4_K_M", seed=1234)
m = LLM(p, config=Config(context_length=int(os.environ.get("SITES_CTX", "512")), batch_size=512), lib=os.environ.get("SITES_LIB") or None)
NP = int(os.environ.get("SITES_PROMPT", "64"))
m.eval(synth.prompt_tokens(NP, 32000))
tok = m.sample(top_k=1, repetition_penalty=1.0)
for _ in range(8):
    m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
t0 = time.perf_counter()
for _ in range(64):
    m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
dt = (time.perf_counter() - t0) / 64
from ctransformers_amd import measure
sites = measure.profile_sites(m._lib, m._llm, 10)
tot = sum(s["ms"] for s in sites) / 10
print(json.dumps(dict(tag=sys.argv[1] if len(sys.argv) > 1 else "", ms_per_token=round(dt * 1e3, 3), tok_s=round(1 / dt, 1),
                      eager_sum_ms=round(tot, 3),
                      sites={s["site"]: round(s["ms"] * 1e3 / s["launches"], 2) for s in sites})))
