"""Repeat-run check of the prompt-chunk path against the reference CPU build on the full 7B file: the LDS-DMA staging of kernels_pg.h
once produced RARE wrong stage data that single runs of the parity tests did not catch.  Fresh handle per repetition,
several prompt lengths (one group, ragged groups, full chunk, two chunks); every logits vector must equal the reference's.
usage: stress_chunks.py [reps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth
from ctransformers_amd.llm import LLM, Config
from oracle import ref

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
p = os.environ.get("CTAMD_BENCH_MODEL", "/tmp/ctamd_llama2_7b_q4km_r2.gguf")
if not os.path.exists(p):
    synth.write_llama_gguf(p, "llama-2-7b", "Q4_K_M", seed=1234)
lens = (17, 24, 33, 96, 128, 200)
want = {}
r = ref.open_llm(p, context_length=512, batch_size=128, threads=16)
for n in lens:
    r.reset(); r._context = []
    r.eval(synth.prompt_tokens(n, 32000))
    want[n] = r.logits.to_numpy().copy()
del r
bad = 0
for rep in range(reps):
    m = LLM(p, None, config=Config(context_length=512, batch_size=128))
    row = []
    for n in lens:
        m._context = []
        m.eval(synth.prompt_tokens(n, 32000))
        ok = np.array_equal(m.logits.to_numpy(), want[n])
        bad += not ok
        row.append("%d:%s" % (n, "same" if ok else "DIFF"))
    del m
    print("rep %d  %s" % (rep, "  ".join(row)), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
