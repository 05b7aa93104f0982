"""A/B of the two prompt-chunk forms on one model file: logits after prompts of several lengths with CT_AMD_PG=1 (f16 matrix cores,
kernels_pg.h) and CT_AMD_PG=0 (int8 form, kernels_pfm.h), fresh handles, repeated.  usage: pg_check.py model.gguf [lengths...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctransformers_amd import synth
from ctransformers_amd.llm import LLM, Config

path = sys.argv[1]
LIB = os.environ.get("PG_CHECK_LIB")   # a library built with different options (default: the product build)
REPS = int(os.environ.get("PG_CHECK_REPS", "3"))
lens = [int(x) for x in sys.argv[2:]] or [2, 9, 16, 17, 24, 32, 33, 64, 128]
def run(pg, n, rep):
    os.environ["CT_AMD_PG"] = pg
    m = LLM(path, None, config=Config(context_length=512, batch_size=128), lib=LIB)
    m.eval(synth.prompt_tokens(n, m.vocab_size))
    out = m.logits.to_numpy().copy()
    del m
    return out
for n in lens:
    base = run("0", n, 0)
    res = []
    for rep in range(REPS):
        a = run("1", n, rep)
        res.append("same" if np.array_equal(a, base) else "DIFF(max %.3g, %d of %d)" % (np.abs(a - base).max(), int((a != base).sum()), a.size))
    print("prompt %4d: %s" % (n, "  ".join(res)), flush=True)
