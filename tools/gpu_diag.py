"""GPU-box diagnostic: per-step bit-identity vs the reference build for several engine variants."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ctransformers_amd import synth
from ctransformers_amd.llm import LLM, Config
from oracle import ref

def run(shape, ft, n_prompt, n, env, seed=7, threads=4):
    for k, v in env.items(): os.environ[k] = str(v)
    p = "/tmp/diag-%s-%s.gguf" % (shape, ft)
    hp = synth.write_llama_gguf(p, shape, ft, seed=seed)
    cfg = dict(context_length=max(64, n_prompt + n + 8), batch_size=64, threads=threads)
    r = ref.open_llm(p, **cfg); m = LLM(p, config=Config(**cfg))
    toks = synth.prompt_tokens(n_prompt, hp["n_vocab"])
    r.eval(toks); m.eval(toks)
    flags = ""
    for i in range(n):
        a = r.logits.to_numpy(); b = m.logits.to_numpy()
        flags += "." if np.array_equal(a, b) else "X"
        t = int(a.argmax()); r.eval([t]); m.eval([t])
    print(shape, ft, env, flags, flush=True)
    for k in env: os.environ.pop(k, None)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "a"
    if which == "b":
        run("llama-7b-2l", "Q4_K_M", 1, 6, {}, seed=11, threads=16)
        run("llama-7b-2l", "Q4_K_M", 1, 6, {"CT_AMD_DESIGN": 1}, seed=11, threads=16)
        run("llama-7b-2l", "Q4_K_M", 1, 6, {"CT_AMD_GRAPH": 0, "CT_AMD_MAXWG": 1}, seed=11, threads=16)
        run("llama-7b-2l", "Q4_K_M", 8, 6, {}, seed=11, threads=16)
    if which == "a":
        run("llama-small", "Q4_K_M", 12, 40, {})
        run("llama-small", "Q4_K_M", 12, 40, {"CT_AMD_DESIGN": 1})
        run("llama-small", "Q4_K_M", 12, 40, {"CT_AMD_GRAPH": 0, "CT_AMD_MAXWG": 1})
        run("llama-small", "Q5_K_M", 12, 40, {})
        run("llama-tiny", "Q4_K_M", 12, 40, {})
        run("llama-small", "Q4_K_M", 1, 40, {})
        run("llama-small", "Q4_K_M", 12, 40, {"CT_AMD_EXACT": 0})
