"""Host share of a decode step: the eval + sample loop of bench.py against the same token steps queued back to back on the device
(ctamd_decode_burst: HIP events around the burst, no host round trip between the steps)."""
import ctypes, json, os, sys, time
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
for _ in range(16):
    m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
f = m._lib.ctamd_decode_burst
f.restype, f.argtypes = ctypes.c_double, [ctypes.c_void_p, ctypes.c_int]
res = []
for rep in range(3):
    N = 64
    t0 = time.perf_counter()
    for _ in range(N):
        m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
    loop_us = (time.perf_counter() - t0) / N * 1e6
    burst_us = f(m._llm, 64)
    res.append(dict(loop_us_per_token=round(loop_us, 1), burst_us_per_token=round(burst_us, 1), host_gap_us=round(loop_us - burst_us, 1)))
print(json.dumps(res))
