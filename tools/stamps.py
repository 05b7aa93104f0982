"""Where a decode step's wall time goes between the graphs (CT_AMD_STAMPS=1: a 1-thread kernel stamps the 100 MHz wall clock at the start
and end of every token step): time inside a step, and the gap from the end of one step to the start of the next — for the eval + sample
loop and for steps queued back to back (ctamd_decode_burst)."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CT_AMD_STAMPS"] = "1"
from tools import synth
from ctransformers_amd.llm import LLM, Config
import numpy as np
p = os.environ.get("CTAMD_BENCH_MODEL", "/tmp/ctamd_llama2_7b_q4km_r2.gguf")
if not os.path.exists(p):
    synth.write_llama_gguf(p, "llama-2-7b", "Q4_K_M", seed=1234)
m = LLM(p, config=Config(context_length=512, batch_size=128))
rd = m._lib.ctamd_read_stamps
rd.restype, rd.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
bf = m._lib.ctamd_decode_burst
bf.restype, bf.argtypes = ctypes.c_double, [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 4000)()


def stamps():
    n = rd(m._llm, buf, 4000)
    a = np.array(buf[:n], dtype=np.uint64)
    return (a >> np.uint64(4)).astype(np.int64), (a & np.uint64(15)).astype(np.int64)


def report(label):
    t, tag = stamps()
    inside, gaps = [], []
    for i in range(len(t) - 1):
        if tag[i] == 1 and tag[i + 1] == 2:
            inside.append((t[i + 1] - t[i]) / 100.0)
        if tag[i] == 2 and tag[i + 1] == 1:
            gaps.append((t[i + 1] - t[i]) / 100.0)
    inside, gaps = inside[4:], gaps[4:]
    print(json.dumps(dict(what=label, steps=len(inside), inside_us=round(float(np.mean(inside)), 1), gap_us_mean=round(float(np.mean(gaps)), 1),
                          gap_us_median=round(float(np.median(gaps)), 1), gap_us_max=round(float(np.max(gaps)), 1))))


m.eval(synth.prompt_tokens(128, 32000))
tok = m.sample(top_k=1, repetition_penalty=1.0)
for _ in range(16):
    m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
stamps()
t0 = time.perf_counter()
for _ in range(64):
    m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
print("loop us/token %.1f" % ((time.perf_counter() - t0) / 64 * 1e6))
report("eval + sample loop")
us = bf(m._llm, 64)
print("burst us/token %.1f" % us)
report("burst")
