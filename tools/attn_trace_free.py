"""In-kernel timeline (workgroup 0) of the long-context decode attention at 2001 positions: the bit-identical form and the order-free one
(CT_AMD_DECODE_ATTN=fast).  Stamps: entry, cursor known + first requests out, scores done, row gathered + max, softmax done, V*P fma done, exit."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth
from ctransformers_amd.llm import LLM, Config
for shape in sys.argv[1:] or ["llama-7b-2l", "llama-70b-2l"]:
    p = "/tmp/%s.gguf" % shape
    if not os.path.exists(p):
        synth.write_llama_gguf(p, shape, "Q5_K_M" if "70b" in shape else "Q4_K_M", seed=5)
    for knob in ("exact", "fast"):
        os.environ["CT_AMD_DECODE_ATTN"] = knob
        m = LLM(p, config=Config(context_length=2048, batch_size=512))
        lib = m._lib
        lib.ctamd_trace_site.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
        buf = (ctypes.c_uint64 * 256)()
        m.eval(synth.prompt_tokens(2000, m.vocab_size))
        tok = m.sample(top_k=1, repetition_penalty=1.0); m.eval([tok])
        for rep in range(3):
            lib.ctamd_trace_site(m._llm, b"attn", buf, 256)
        rows = [[buf[16 * w + k] for k in range(8)] for w in range(16)]
        t0 = min(r[0] for r in rows if r[0])
        print(shape, knob, "n_kv", rows[0][7], flush=True)
        for w in range(8):
            r = rows[w]
            if r[0]:
                print("  wave %2d: " % w + " ".join("%s=%6d" % (n, r[k] - t0 if r[k] else -1) for k, n in enumerate(("entry", "cursor", "scores", "max", "softmax", "pv_fma", "exit"))))
        del m
