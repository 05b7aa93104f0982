"""Tiny driver for rocprofv3: load a model, prefill, decode N tokens greedily."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth  # noqa: E402
from ctransformers_amd.llm import LLM, Config  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", required=True)
ap.add_argument("--shape", default="")
ap.add_argument("--ftype", default="Q4_K_M")
ap.add_argument("--prompt", type=int, default=16)
ap.add_argument("--decode", type=int, default=32)
ap.add_argument("--ctx", type=int, default=512)
ap.add_argument("--batch", type=int, default=0, help="batch_size of the prompt evaluation (default: the whole prompt)")
a = ap.parse_args()
if a.shape and not os.path.exists(a.model):
    synth.write_llama_gguf(a.model, a.shape, a.ftype, seed=1234)
m = LLM(a.model, config=Config(context_length=a.ctx, batch_size=a.batch or a.prompt))
m.eval(synth.prompt_tokens(a.prompt, m.vocab_size))
tok = m.sample(top_k=1, repetition_penalty=1.0)
for _ in range(a.decode):
    m.eval([tok])
    tok = m.sample(top_k=1, repetition_penalty=1.0)
print("done", tok)
