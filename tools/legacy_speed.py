#!/usr/bin/env python3
"""Decode / prefill rates of the legacy-GGML architectures at real sizes (synthetic weights): MPT-7B Q4_0 and StarCoderBase-1B Q8_0.
Not a BASELINE config — the numbers go into DESIGN.md §8 as a record of where these paths stand.  Run on the GPU box:
    python tools/legacy_speed.py > gpurun_out/legacy_speed.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth  # noqa: E402
from ctransformers_amd.llm import LLM, Config  # noqa: E402


def run(name, path, model_type, n_vocab, n_prompt=128, n_decode=128):
    t0 = time.perf_counter()
    m = LLM(path, model_type, config=Config(context_length=512, batch_size=n_prompt))
    load = time.perf_counter() - t0
    prompt = synth.prompt_tokens(n_prompt, n_vocab)
    m.eval(prompt); m._context = []
    m.eval(prompt); m._context = []
    t0 = time.perf_counter()
    m.eval(prompt)
    pre = time.perf_counter() - t0
    tok = m.sample(top_k=1, repetition_penalty=1.0)
    for _ in range(8):
        m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
    t0 = time.perf_counter()
    for _ in range(n_decode):
        m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
    dec = time.perf_counter() - t0
    print("%s: file %.2f GB, load %.2f s, prefill %d tok in %.1f ms = %.0f tok/s, decode %.1f tok/s (%.3f ms/token)" % (
        name, os.path.getsize(path) / 1e9, load, n_prompt, pre * 1e3, n_prompt / pre, n_decode / dec, dec / n_decode * 1e3), flush=True)


if __name__ == "__main__":
    p = "/tmp/ctamd_mpt7b_q40.bin"
    if not os.path.exists(p):
        synth.write_mpt_ggml(p, dict(synth.MPT_SHAPES["mpt-7b-2l"], n_layer=32), seed=5, ftype=2)
    run("MPT-7B Q4_0", p, "mpt", 50432)
    p = "/tmp/ctamd_starcoder1b_q80.bin"
    if not os.path.exists(p):
        synth.write_gpt2_ggml(p, dict(synth.GPT2_SHAPES["starcoder-1b"], n_ctx=2048), seed=5, ftype=7, pieces=synth.STARCODER_PIECES)
    run("StarCoderBase-1B Q8_0", p, "starcoder", 49152)
