"""In-kernel timeline (s_memtime) of workgroup 0 for each mat-vec launch site of one decode step."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth
from ctransformers_amd.llm import LLM, Config
p = os.environ.get("CTAMD_BENCH_MODEL", "/tmp/ctamd_llama2_7b_q4km_r2.gguf")
if not os.path.exists(p):
    synth.write_llama_gguf(p, "llama-2-7b", "Q4_K_M", seed=1234)
m = LLM(p, config=Config(context_length=int(os.environ.get("SITES_CTX", "512")), batch_size=512), lib=os.environ.get("SITES_LIB") or None)
m.eval(synth.prompt_tokens(int(os.environ.get("SITES_PROMPT", "64")), 32000))
lib = m._lib
lib.ctamd_trace_site.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
buf = (ctypes.c_uint64 * 256)()
for site in ("qkv", "wo", "gate_up", "down", "lm_head"):
    for rep in range(2):
        lib.ctamd_trace_site(m._llm, site.encode(), buf, 256)
    rows = [[buf[16 * w + k] for k in range(16)] for w in range(16)]
    t0 = min(r[0] for r in rows if r[0])
    print(site)
    for w in (0, 1, 2, 3, 5, 10, 15):
        r = rows[w]
        print("  wave %2d: " % w + " ".join("%s=%6d" % (n, r[k] - t0 if r[k] else -1) for k, n in enumerate(("entry", "loads", "prolog", "math", "barrier", "chain", "exit"))) + "  pro(x arrived, scale known, quantized, images written):" + " ".join("%d" % (r[k] - t0) for k in range(8, 12) if r[k] > t0))
for rep in range(2):
    lib.ctamd_trace_site(m._llm, b"attn", buf, 256)
rows = [[buf[16 * w + k] for k in range(8)] for w in range(16)]
t0 = rows[0][0]
print("attn (n_kv=%d)" % rows[0][7])
for w in (0, 1, 3):
    r = rows[w]
    print("  wave %2d: " % w + " ".join("%s=%6d" % (n, r[k] - t0 if r[k] else -1) for k, n in enumerate(("entry", "pos", "scores", "max", "softmax", "pv_fma", "exit"))))
