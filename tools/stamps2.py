"""Per-site times inside queued token steps, without a profiler (CT_AMD_STAMPS=2: a 1-thread kernel stamps the wall clock behind every launch
site): the average time from the previous stamp to the site's stamp, for a burst of steps (ctamd_decode_burst)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CT_AMD_STAMPS"] = "2"
from tools import synth
from ctransformers_amd.llm import LLM, Config
import numpy as np
p = os.environ.get("CTAMD_BENCH_MODEL", "/tmp/ctamd_llama2_7b_q4km_r2.gguf")
m = LLM(p, config=Config(context_length=512, batch_size=128))
rd = m._lib.ctamd_read_stamps
rd.restype, rd.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
bf = m._lib.ctamd_decode_burst
bf.restype, bf.argtypes = ctypes.c_double, [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 40000)()
m.eval(synth.prompt_tokens(128, 32000))
tok = m.sample(top_k=1, repetition_penalty=1.0)
for _ in range(int(os.environ.get("PRE", "16"))):
    m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
rd(m._llm, buf, 40000)
us = bf(m._llm, 40)
n = rd(m._llm, buf, 40000)
a = np.array(buf[:n], dtype=np.uint64)
t, tag = (a >> np.uint64(4)).astype(np.int64), (a & np.uint64(15)).astype(np.int64)
names = {1: "step start (gap from the previous step's end)", 2: "step end (pick_cont / advance)", 3: "qkv", 4: "attn", 5: "wo", 6: "gate_up", 7: "down", 8: "lm_head"}
acc = {}
first = int(np.argmax(tag == 1))
seen_steps = 0
for i in range(first + 1, n):
    if tag[i] == 1:
        seen_steps += 1
    if seen_steps < 3:
        continue
    acc.setdefault(int(tag[i]), []).append((t[i] - t[i - 1]) / 100.0)
out = {names[k]: [round(float(np.mean(v)), 2), len(v)] for k, v in sorted(acc.items())}
layer = sum(np.mean(acc[k]) for k in (3, 4, 5, 6, 7))
print(json.dumps(dict(burst_us_per_token=round(us, 1), per_layer_us=round(float(layer), 2), sites=out)))
