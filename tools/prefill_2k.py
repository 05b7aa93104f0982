#!/usr/bin/env python3
"""Prompt throughput at a 2k context (BASELINE config 5 asks for a 2k-context prefill): context 2048, a prompt of n tokens evaluated
three times from position 0 (cold, graph capture, steady state) — 128-token chunks, the later ones attending to up to 2k positions.
usage (GPU box): prefill_2k.py [model.gguf] [n_tokens] [batch_size] [shape ftype]     (default: the 7B Q4_K_M bench file; `llama-2-70b
Q5_K_M` = config 5's model on one GPU).  Prints a text summary and one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth  # noqa: E402
from ctransformers_amd.llm import LLM, Config  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/ctamd_llama2_7b_q4km_r2.gguf"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 128   # the reference's batch: the V*P dot of a token runs to the end of ITS batch
shape = sys.argv[4] if len(sys.argv) > 4 else "llama-2-7b"
ftype = sys.argv[5] if len(sys.argv) > 5 else "Q4_K_M"
if not os.path.exists(path):
    (synth.write_falcon_gguf if shape.startswith("falcon") else synth.write_llama_gguf)(path, shape, ftype, seed=1234)
m = LLM(path, config=Config(context_length=2048, batch_size=bs, gpu_layers=1000))
toks = synth.prompt_tokens(n, m.vocab_size)
ts = []
for _ in range(3):
    m._context = []
    t0 = time.perf_counter(); m.eval(toks); ts.append(time.perf_counter() - t0)
print("%s %s, context 2048, batch_size %d, %d-token prompt in 128-token chunks: %.0f / %.0f / %.0f tok/s (cold / capture / steady); steady %.1f ms" % (
    shape, ftype, bs, n, n / ts[0], n / ts[1], n / ts[2], ts[2] * 1e3))
first = {}
for k in (128, 512, 1024):
    if k > n:
        continue
    m._context = []
    t0 = time.perf_counter(); m.eval(toks[:k]); dt = time.perf_counter() - t0
    first[k] = round(k / dt, 1)
    print("  first %4d tokens: %.0f tok/s" % (k, k / dt))
print(json.dumps(dict(shape=shape, ftype=ftype, context=2048, batch_size=bs, n_prompt=n, tok_s_cold=round(n / ts[0], 1), tok_s_capture=round(n / ts[1], 1),
                      tok_s_steady=round(n / ts[2], 1), ms_steady=round(ts[2] * 1e3, 2), first_tokens_tok_s=first)))
