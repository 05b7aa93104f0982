#!/usr/bin/env python3
"""Headline benchmark: greedy decode tokens/s (and prefill tok/s) of synthetic Llama-2-7B Q4_K_M through the C ABI.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
A "step" is one decoded token = one pass of the hot path (`llm.eval([token])` -> ctransformers_llm_batch_eval) with
all weights and the KV cache resident in HBM; K steps are timed after a 128-token prefill and W untimed warm-up
steps, bracketed by barrier + device synchronisation, MAX over ranks.

Workload = BASELINE.json configs[1]: Llama-2-7B GGUF Q4_K_M (synthetic weights at the real shapes / tensor-type mix),
all layers on the GPU(s), 128-token prefill + 256-token decode, context 512.

Extra objects on the JSON line:
  roofline      dominant kernel (the K=4096 weight mat-vec launch: QKV / Wo / gate+up / lm_head sites) —
                algorithmic weight bytes per launch / HIP-event time per launch, vs 8 TB/s HBM3E peak
  cpu_baseline  the REAL reference CPU build (oracle/_ref) on this box's host cores, bounded sample of the same job
  prefill       the 128-token prompt through the prompt-chunk kernels (third pass = steady state; the cold first pass beside it)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from ctransformers_amd import measure, synth  # noqa: E402
from ctransformers_amd.llm import LLM, Config  # noqa: E402

N_PROMPT, N_DECODE, N_CTX = 128, 256, 512
MODEL = os.environ.get("CTAMD_BENCH_MODEL", "/tmp/ctamd_llama2_7b_q4km.gguf")
SHAPE, FTYPE = os.environ.get("CTAMD_BENCH_SHAPE", "llama-2-7b"), "Q4_K_M"


def ensure_model(rank):
    if rank == 0 and not os.path.exists(MODEL):
        tmp = MODEL + ".tmp%d" % os.getpid()
        synth.write_llama_gguf(tmp, SHAPE, FTYPE, seed=1234)
        os.replace(tmp, MODEL)


def cpu_baseline(n_vocab):
    """The reference CPU build on this box's host cores, bounded sample (16-token prefill + 24 greedy decode steps).
    When oracle/_ref did not travel with the snapshot, the scalar C restatement (oracle/mirror.c) is timed instead on a
    smaller sample and reported as kind "port"."""
    from oracle import mirror, ref
    if ref.available():
        threads = min(16, os.cpu_count() or 1)
        r = ref.open_llm(MODEL, context_length=N_CTX, batch_size=16, threads=threads)
        t0 = time.perf_counter()
        r.eval(synth.prompt_tokens(16, n_vocab))     # includes the first touch of the mmap'ed weights (page cache warm from the GPU load)
        t_first = time.perf_counter() - t0
        r._context = []
        t0 = time.perf_counter()
        r.eval(synth.prompt_tokens(16, n_vocab))     # the same 16-token batch again: the reference's prompt rate
        t_prefill = time.perf_counter() - t0
        ts = []
        for _ in range(24):
            tok = r.sample(top_k=1, repetition_penalty=1.0)
            t0 = time.perf_counter()
            r.eval([tok])
            ts.append(time.perf_counter() - t0)
        return dict(value=round(1.0 / float(np.median(ts)), 3), unit="tokens/s", cores=threads, kind="reference",
                    prefill_tok_s=round(16.0 / min(t_first, t_prefill), 2),
                    sample="reference AVX2 build (oracle/_ref), threads=%d, same synthetic 7B file, 16-token prefill (one batch, "
                           "best of two: prefill_tok_s) then 24 greedy decode steps, median step time" % threads)
    if not mirror.available():
        return None
    o = mirror.MirrorLlama(MODEL, N_CTX)
    lg = o.eval(synth.prompt_tokens(2, n_vocab), 0)
    t0 = time.perf_counter()
    o.eval([int(np.argmax(lg))], 2)
    dt = time.perf_counter() - t0
    return dict(value=round(1.0 / dt, 4), unit="tokens/s", cores=1, kind="port",
                sample="scalar C restatement (oracle/mirror.c), 1 thread, same synthetic 7B file, 2-token prefill then ONE decode step")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=N_DECODE)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 or world > 1 or os.environ.get("CTAMD_FORCE_PIPELINE") == "1":   # the env switch lets a 1-GPU box run the N > 1 code path with world_size 1
        from ctransformers_amd import pipeline
        return pipeline.bench_main(a, MODEL, SHAPE, FTYPE)

    ensure_model(rank)
    t0 = time.perf_counter()
    llm = LLM(MODEL, config=Config(context_length=N_CTX, batch_size=N_PROMPT, gpu_layers=1000))
    load_s = time.perf_counter() - t0
    n_vocab = llm.vocab_size
    prompt = synth.prompt_tokens(N_PROMPT, n_vocab)
    # prefill (timed separately; reported, not the headline value): once cold (first use of every kernel: code-object load,
    # LDS opt-ins), a second pass (graph capture), then the same prompt again from position 0 — the steady-state number
    t0 = time.perf_counter()
    llm.eval(prompt)
    prefill_cold_s = time.perf_counter() - t0
    llm._context = []
    llm.eval(prompt)            # second pass: the library captures the hipGraph of each chunk shape on its second use
    llm._context = []
    t0 = time.perf_counter()
    llm.eval(prompt)
    prefill_s = time.perf_counter() - t0
    tok = llm.sample(top_k=1, repetition_penalty=1.0)
    for _ in range(a.warmup):
        llm.eval([tok])
        tok = llm.sample(top_k=1, repetition_penalty=1.0)
    steps = min(a.steps, N_CTX - N_PROMPT - a.warmup - 1)
    # llm.eval() returns only after the library synchronised its stream and copied the logits to the host, so the
    # wall clock below brackets exactly `steps` complete decode steps (device sync on both sides).
    t0 = time.perf_counter()
    for _ in range(steps):
        llm.eval([tok])
        tok = llm.sample(top_k=1, repetition_penalty=1.0)
    dt = time.perf_counter() - t0
    tok_s = steps / dt

    sites = measure.profile_sites(llm._lib, llm._llm, 8)
    roof = measure.roofline(sites)
    wbytes = synth.weight_bytes_per_token(MODEL)
    kv_avg = 2 * 32 * (N_PROMPT + a.warmup + steps / 2.0) * 4096 * 2 if SHAPE == "llama-2-7b" else 0
    out = dict(metric="decode_tokens_per_s", value=round(tok_s, 2), unit="tokens/s", n_gpus=1, steps=steps, warmup=a.warmup,
               ms_per_step=round(dt / steps * 1e3, 4), higher_is_better=True, scaling="weak", vs_baseline=None, dtype="int8",
               data="synthetic",
               config=dict(workload="Llama-2-7B GGUF Q4_K_M, all layers on 1xMI355X, 128-tok prefill + 256-tok greedy decode, ctx 512",
                           shape=SHAPE, ftype=FTYPE, n_prompt=N_PROMPT, parallelism="1 GPU"),
               prefill_tok_s=round(N_PROMPT / prefill_s, 1), prefill_cold_tok_s=round(N_PROMPT / prefill_cold_s, 1), load_s=round(load_s, 2),
               token_roofline=dict(bytes_per_token=int(wbytes + kv_avg), frac_of_8TBps=round(tok_s * (wbytes + kv_avg) / measure.HBM_PEAK, 4)),
               roofline=roof)
    # prompt chunks: 2 ops per weight of the 2-D matrices per token, against the dense int8 MFMA
    # floor of the guide; the bound is VALU issue (the exact f32 chain step per block, AVX lane, row and token), not MFMA
    pf_flop = 2 * 6.607e9 if SHAPE == "llama-2-7b" else None
    out["prefill"] = dict(tok_s=out["prefill_tok_s"], cold_tok_s=out["prefill_cold_tok_s"], chunk_tokens=128,
                          kernel="matvec_pfm_kernel<TYPE,TOK,GU> (int8 MFMA, exact), one hipGraph per chunk shape",
                          int8_tops=round(out["prefill_tok_s"] * pf_flop / 1e12, 1) if pf_flop else None, mfma_peak_tops=3944,
                          bound="valu")
    if not a.no_cpu_baseline:
        del llm
        out["cpu_baseline"] = cpu_baseline(n_vocab)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
