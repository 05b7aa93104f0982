"""In-kernel timeline of the decode attention launch at several context lengths (2-layer model at Llama-2-70B widths)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth
from ctransformers_amd.llm import LLM, Config
shape = sys.argv[1] if len(sys.argv) > 1 else "llama-70b-2l"
p = "/tmp/%s.gguf" % shape
if not os.path.exists(p):
    synth.write_llama_gguf(p, shape, "Q5_K_M" if "70b" in shape else "Q4_K_M", seed=5)
m = LLM(p, config=Config(context_length=int(os.environ.get("ATTN_CTX", "2048")), batch_size=512))
lib = m._lib
lib.ctamd_trace_site.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
buf = (ctypes.c_uint64 * 256)()
done = 0
for target in [t for t in (128, 512, 2000) if t < int(os.environ.get("ATTN_CTX", "2048"))]:
    m.eval(synth.prompt_tokens(target - done, m.vocab_size)); done = target
    tok = m.sample(top_k=1, repetition_penalty=1.0); m.eval([tok]); done += 1
    for rep in range(3):
        lib.ctamd_trace_site(m._llm, b"attn", buf, 256)
    rows = [[buf[16 * w + k] for k in range(8)] for w in range(16)]
    t0 = min(r[0] for r in rows if r[0])
    print("n_kv", rows[0][7])
    for w in (0, 3, 4, 5, 7):
        r = rows[w]
        print("  wave %2d: " % w + " ".join("%s=%6d" % (n, r[k] - t0 if r[k] else -1) for k, n in enumerate(("entry", "cursor", "scores", "max", "softmax", "pv_fma", "exit"))))
