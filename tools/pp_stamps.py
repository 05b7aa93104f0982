"""Where a decode step goes in the in-process pipeline on ONE device (CT_AMD_STAMPS=1, CT_AMD_DEVICES=0,0,..): per stage the time inside
its token step, the gap from stage s's end to stage s + 1's start (the hop), and from the last stage's end to stage 0's next start."""
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
m = LLM(p, config=Config(context_length=512, batch_size=128, gpu_layers=1000))
L = m._lib
L.ctamd_n_stages.restype, L.ctamd_n_stages.argtypes = ctypes.c_int, [ctypes.c_void_p]
S = L.ctamd_n_stages(m._llm)
L.ctamd_handoff.restype, L.ctamd_handoff.argtypes = ctypes.c_char_p, [ctypes.c_void_p]
HAND = L.ctamd_handoff(m._llm).decode()
rd = L.ctamd_read_stamps_stage
rd.restype, rd.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 40000)()
def stamps(s):
    n = rd(m._llm, s, buf, 40000)
    a = np.array(buf[:n], dtype=np.uint64)
    return (a >> np.uint64(4)).astype(np.int64), (a & np.uint64(15)).astype(np.int64)
m.eval(synth.prompt_tokens(128, 32000))
tok = m.sample(top_k=1, repetition_penalty=1.0)
for _ in range(8):
    m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
for s in range(S): stamps(s)
N = 48
t0 = time.perf_counter()
for _ in range(N):
    m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
loop = (time.perf_counter() - t0) / N * 1e6
st = [stamps(s) for s in range(S)]
starts = [t[tag == 1] for t, tag in st]
ends = [t[tag == 2] for t, tag in st]
inside = [float(np.mean(ends[s][:N] - starts[s][:N])) / 100 for s in range(S)]
hops = [float(np.mean(starts[s + 1][:N] - ends[s][:N])) / 100 for s in range(S - 1)]
wrap = float(np.mean(starts[0][1:N] - ends[S - 1][:N - 1])) / 100
print(json.dumps(dict(stages=S, handoff=HAND, loop_us_per_token=round(loop, 1), inside_us=[round(x, 1) for x in inside],
                      hop_us=[round(x, 1) for x in hops], last_end_to_next_start_us=round(wrap, 1))))
