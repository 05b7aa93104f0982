"""Attention cost vs context length on a 2-layer model at Llama-2-70B widths (GQA 64/8, head_dim 128)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ctransformers_amd import measure
from tools import synth
from ctransformers_amd.llm import LLM, Config
shape = sys.argv[1] if len(sys.argv) > 1 else "llama-70b-2l"
p = "/tmp/%s.gguf" % shape
if not os.path.exists(p):
    synth.write_llama_gguf(p, shape, "Q5_K_M" if "70b" in shape else "Q4_K_M", seed=5)
m = LLM(p, config=Config(context_length=2048, batch_size=512))
done = 0
for target in (64, 512, 1024, 2000):
    m.eval(synth.prompt_tokens(target - done, m.vocab_size))
    done = target
    tok = m.sample(top_k=1, repetition_penalty=1.0)
    m.eval([tok]); done += 1
    sites = measure.profile_sites(m._lib, m._llm, 4)
    d = {s["site"]: round(s["ms"] * 1e3 / s["launches"], 2) for s in sites if s["site"].endswith("@sweep")}
    print(json.dumps(dict(shape=shape, pos=done, sweep_us=d)))
