"""Decode attention at long contexts, bit-identical form against the order-free one (CT_AMD_DECODE_ATTN=fast): per-site sweep times at 2001 positions
on the two-layer models at the 7B / 70B / Falcon-40B widths, and whole-token decode rates behind 2000 positions."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ctransformers_amd import measure
from tools import synth
from ctransformers_amd.llm import LLM, Config

for shape in sys.argv[1:] or ["llama-7b-2l", "llama-70b-2l", "falcon-40b-2l"]:
    falcon = shape.startswith("falcon")
    p = "/tmp/%s.gguf" % shape
    if not os.path.exists(p):
        (synth.write_falcon_gguf if falcon else synth.write_llama_gguf)(p, shape, "Q5_K_M" if "70b" in shape else "Q4_K_M", seed=5)
    for knob in ("exact", "fast"):
        os.environ["CT_AMD_DECODE_ATTN"] = knob
        m = LLM(p, config=Config(context_length=2048, batch_size=512))
        for target in (1024, 2000):
            m.eval(synth.prompt_tokens(target - (0 if target == 1024 else 1025), m.vocab_size))
            tok = m.sample(top_k=1, repetition_penalty=1.0)
            m.eval([tok])
            sites = measure.profile_sites(m._lib, m._llm, 4)
            d = {s["site"]: round(s["ms"] * 1e3 / s["launches"], 2) for s in sites if s["site"].endswith("@sweep")}
            print(json.dumps(dict(shape=shape, decode_attn=knob, pos=target + 1, sweep_us=d)), flush=True)
        del m
