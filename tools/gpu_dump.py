"""Dump per-launch scratch buffers for a fixed scenario (run once on the GPU box, once on the CPU emulation)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ctransformers_amd import synth
from ctransformers_amd.llm import LLM, Config
out, lib, n = sys.argv[1], (sys.argv[2] if sys.argv[2] != "hip" else None), int(sys.argv[3])
os.makedirs(out, exist_ok=True)
os.environ["CT_AMD_DUMP"] = out
shape = os.environ.get("DUMP_SHAPE", "llama-small")
p = "/tmp/dump-%s.gguf" % shape
hp = synth.write_llama_gguf(p, shape, "Q4_K_M", seed=int(os.environ.get("DUMP_SEED", "7")))
m = LLM(p, config=Config(context_length=64, batch_size=64), lib=lib)
toks = synth.prompt_tokens(int(os.environ.get("DUMP_PROMPT", "12")), hp["n_vocab"])
m.eval(toks)
seq = [int(x) for x in os.environ.get("DUMP_TOKENS", "").split(",") if x]
for i in range(n):
    t = seq[i] if i < len(seq) else m.sample(top_k=1, repetition_penalty=1.0)
    print(t, end=",", flush=True)
    m.eval([t])
print()
