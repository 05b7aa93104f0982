"""fp16 K / V cache rows of chosen layers after one prompt, order-free prompt kernels against the bit-identical ones (a layer-0 row is a pure function of
the embedding rows and the layer's QKV launch: what differs there is the launch itself, not an amplified earlier difference).
usage: python tools/mm8_kv_check.py [shape] [ftype] [n_prompt]"""
import ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tools import synth, mm8_check

LAYERS = (0, 1, 2, 4, 8, 16, 31)


def worker(mode, shape, ftype, n_prompt):
    os.environ["CT_AMD_PREFILL"] = mode
    from ctransformers_amd.llm import LLM, Config
    m = LLM(mm8_check.model_path(shape, ftype), config=Config(context_length=512, batch_size=n_prompt))
    hp = synth.LLAMA_SHAPES[shape]
    G = hp["n_embd"] // hp["n_head"] * hp["n_head_kv"]
    m.eval(synth.prompt_tokens(n_prompt, m.vocab_size))
    f = m._lib.ctamd_debug_read_kv
    f.restype, f.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    out = {}
    for il in LAYERS:
        if il >= hp["n_layer"]:
            continue
        k = np.zeros(512 * G, np.uint16); v = np.zeros((512 + 512) * G, np.uint16)   # (rows are padded: the call returns the stride)
        vs = f(m._llm, il, k.ctypes.data, v.ctypes.data)
        out["k%d" % il] = k.reshape(-1, 512, hp["n_embd"] // hp["n_head"])[:, :n_prompt, :].copy()
        out["v%d" % il] = v[:G * vs].reshape(G, vs)[:, :n_prompt].copy()
    np.savez("/tmp/mm8_kv_%s.npz" % mode, **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]))
        sys.exit(0)
    shape = sys.argv[1] if len(sys.argv) > 1 else "llama-2-7b"
    ftype = sys.argv[2] if len(sys.argv) > 2 else "Q4_K_M"
    n_prompt = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    mm8_check.model_path(shape, ftype)
    # KV_A / KV_B = "mode[:chunk]" override the pair (e.g. "exact:64" against "exact:128": the bit-identical kernels do not depend on the chunking)
    pair = [os.environ.get("KV_A", "exact"), os.environ.get("KV_B", "fast")]
    for i, spec in enumerate(pair):
        mode, _, chunk = spec.partition(":")
        env = dict(os.environ)
        if chunk:
            env["CT_AMD_PF_CHUNK"] = chunk
        subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", mode, shape, ftype, str(n_prompt)], check=True, env=env)
        os.replace("/tmp/mm8_kv_%s.npz" % mode, "/tmp/mm8_kv_%d.npz" % i)
    os.replace("/tmp/mm8_kv_0.npz", "/tmp/mm8_kv_exact.npz"); os.replace("/tmp/mm8_kv_1.npz", "/tmp/mm8_kv_fast.npz")
    a, b = np.load("/tmp/mm8_kv_exact.npz"), np.load("/tmp/mm8_kv_fast.npz")
    for key in a.files:
        x, y = a[key].view(np.float16).astype(np.float64), b[key].view(np.float16).astype(np.float64)
        d = np.abs(x - y)
        nz = d > 0
        ulp = np.abs(a[key].astype(np.int64) - b[key].astype(np.int64))   # fp16 bit patterns of the same sign differ by their ulp distance
        if key == "k1" and os.environ.get("KV_PER_POS"):
            per = d.max(axis=(0, 2)) / np.abs(x).max()   # [pos]
            print("k1 per-position max diff (x1000):", " ".join("%d" % int(1000 * v) for v in per))
        print("%-4s %9d values, %7d differ (%.4f %%), max |diff| / max |x| %.3g, max ulp distance %d, share of differing values more than 1 ulp apart %.4f" %
              (key, x.size, int(nz.sum()), 100.0 * nz.mean(), d.max() / np.abs(x).max(), int(ulp.max()), float((ulp[nz] > 1).mean()) if nz.any() else 0.0))
