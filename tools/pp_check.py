"""Pipeline correctness probe: N stages on one device against the single-stage result of the same library (greedy tokens + logits)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tools import synth
from ctransformers_amd.llm import LLM, Config
shape, nl, ns = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
p = "/tmp/pp_check_%s_%d.gguf" % (shape, nl)
if not os.path.exists(p):
    synth.write_llama_gguf(p, shape, "Q5_K_M" if "70b" in shape else "Q4_K_M", seed=9, overrides=dict(n_layer=nl))
cfg = dict(context_length=128, batch_size=128)
toks = synth.prompt_tokens(72, 32000)
def run(devs):
    if devs: os.environ["CT_AMD_DEVICES"] = devs
    else: os.environ.pop("CT_AMD_DEVICES", None)
    m = LLM(p, config=Config(**cfg))
    m.eval(toks)
    out = [m.logits.to_numpy().copy()]
    for _ in range(6):
        t = m.sample(top_k=1, repetition_penalty=1.0)
        m.eval([t])
        out.append(m.logits.to_numpy().copy())
    return out
a = run("")
b = run(",".join(["0"] * ns))
print(shape, nl, ns, os.environ.get("CT_AMD_HANDOFF"), os.environ.get("CT_AMD_GRAPH"), os.environ.get("CT_AMD_FUSE_QA"),
      [bool(np.array_equal(x, y)) for x, y in zip(a, b)], "nan" if any(np.isnan(y).any() for y in b) else "")
