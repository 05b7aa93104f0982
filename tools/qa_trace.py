"""In-kernel timeline of workgroup 0 of the fused QKV + attention launch (last layer of one decode step), cycles from its first stamp."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth
from ctransformers_amd.llm import LLM, Config
p = os.environ.get("CTAMD_BENCH_MODEL", "/tmp/ctamd_llama2_7b_q4km_r2.gguf")
if not os.path.exists(p):
    synth.write_llama_gguf(p, "llama-2-7b", "Q4_K_M", seed=1234)
m = LLM(p, config=Config(context_length=512, batch_size=512))
m.eval(synth.prompt_tokens(int(os.environ.get("SITES_PROMPT", "200")), 32000))
lib = m._lib
lib.ctamd_trace_site.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
buf = (ctypes.c_uint64 * 512)()
for rep in range(3):
    lib.ctamd_trace_site(m._llm, b"qa", buf, 512)
rows = [[buf[16 * w + k] for k in range(16)] for w in range(16)]
t0 = min(r[0] for r in rows if r[0])
print("fused qkv + attention, n_kv=%d (cycles of s_memtime, 100 MHz x ...: see gpu.h clock64_dev)" % rows[0][7])
names = ((0, "entry"), (1, "rows done"), (2, "sweep/pre-barrier"), (3, "exchange landed"), (8, "scores"), (9, "max"), (4, "softmax"), (6, "exit"))
for w in range(16):
    r = rows[w]
    v = [buf[256 + 16 * w + k] for k in range(16)]
    print("  wave %2d: " % w + " ".join("%s=%6d" % (n, r[k] - t0 if r[k] else -1) for k, n in names) +
          " | v9: first requests=%d prologue end=%d loop end=%d (x arrived %d, scale %d, quantized %d, images %d)" % tuple((v[k] - t0 if v[k] else -1) for k in (1, 2, 3, 8, 9, 10, 11)))
