#!/usr/bin/env python3
"""Headline benchmark: greedy decode tokens/s (and prefill tok/s) of synthetic Llama-2-7B Q4_K_M through the C ABI.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
A "step" is one decoded token = one pass of the hot path (`llm.eval([token])` -> ctransformers_llm_batch_eval) with
all weights and the KV cache resident in HBM; K steps are timed after a 128-token prefill and W untimed warm-up
steps, bracketed by barrier + device synchronisation, MAX over ranks.

Workload = BASELINE.json configs[1]: Llama-2-7B GGUF Q4_K_M (synthetic weights at the real shapes / tensor-type mix),
all layers on the GPU(s), 128-token prefill + 256-token decode, context 512.

N > 1 (north_star: whole layers spread over the GPUs of the node, activations handed from GPU to GPU): the product path is the
in-process pipeline of the library (ctransformers_amd/csrc/pipeline.cc, CT_AMD_DEVICES): ONE process drives a stage per GPU,
hand-off in-stream over xGMI (a kernel stores the rows into the next stage's peer-mapped buffer, the next stage's stream waits on a sequence word).  `python bench.py --gpus N` therefore brings the N stages up itself.  When the
driver launches N ranks with torch.distributed.run, rank 0 drives the N stages and the other ranks take part in the barriers
and the MAX reduction only (backend gloo: they own no GPU work); where a rank cannot see N devices the one-process-per-GPU
RCCL pipeline of tools/rccl_pipeline.py runs instead.  Decode of one sequence is serial over the stages (strong
scaling: the model is fixed), so the roofline denominator stays ONE GPU's HBM.

`--config 3|4|5` runs the other single-GPU-capable BASELINE configs instead (3: Llama-2-7B Q8_0; 4: Falcon-40B Q4_K_M, 25 GB; 5:
Llama-2-70B Q5_K_M, 49 GB — each fits one MI355X's 288 GB; with --gpus N their layers are spread over N GPUs); the default run is
config 2.  The 40B / 70B files take minutes to synthesise on first use.

Extra objects on the JSON line:
  roofline      dominant kernel (the K=4096 weight mat-vec launches: QKV / Wo / gate+up / lm_head sites) —
                algorithmic weight bytes per launch / HIP-event time per launch, vs 8 TB/s HBM3E peak
  cpu_baseline  the REAL reference CPU build (oracle/_ref) on this box's host cores, bounded sample of the same job
  prefill       the 128-token prompt through the prompt-chunk kernels (third pass = steady state; the cold first pass beside it)
  other_configs config 3 (always) and configs 4 / 5 (when the scratch disk has 53 GB free; CTAMD_BENCH_BIG=0 / 1 forces) as child runs: decode tok/s,
                per-token roofline fraction, prefill
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from ctransformers_amd import measure
from tools import synth  # noqa: E402
from ctransformers_amd.llm import LLM, Config  # noqa: E402

N_PROMPT, N_DECODE, N_CTX = 128, 256, int(os.environ.get("CTAMD_BENCH_CTX", "512"))   # (CTAMD_BENCH_CTX: layout experiments only; the headline is 512)
def _ref_quantizer():
    """BASELINE.md 3 quantizes with ggml_quantize_chunk: where the reference build travelled with the snapshot (oracle/_ref exports it) the
    block pools of the synthetic files come from it; else from this repo's numpy quantizers (tools/synth.py)."""
    try:
        from oracle import ref
        return "reference" if ref.available() else None
    except Exception:   # noqa: BLE001
        return None


QUANTIZER = _ref_quantizer()
_QTAG = "refq" if QUANTIZER else "r2"
MODEL = os.environ.get("CTAMD_BENCH_MODEL", "/tmp/ctamd_llama2_7b_q4km_%s.gguf" % _QTAG)
SHAPE, FTYPE = os.environ.get("CTAMD_BENCH_SHAPE", "llama-2-7b"), os.environ.get("CTAMD_BENCH_FTYPE", "Q4_K_M")
GEN_VERSION = "synth-%s:%s:%s:seed1234" % (_QTAG, SHAPE, FTYPE)
CONFIG = 2
CONFIGS = {2: ("llama-2-7b", "Q4_K_M", "/tmp/ctamd_llama2_7b_q4km_%s.gguf" % _QTAG), 3: ("llama-2-7b", "Q8_0", "/tmp/ctamd_llama2_7b_q80_%s.gguf" % _QTAG),
           4: ("falcon-40b", "Q4_K_M", "/tmp/ctamd_falcon_40b_q4km_%s.gguf" % _QTAG), 5: ("llama-2-70b", "Q5_K_M", "/tmp/ctamd_llama2_70b_q5km_%s.gguf" % _QTAG)}
N_PROMPT_2K, N_CTX_2K, N_DECODE_2K = 2048, 2304, 32
SETTLE_S = 0.75   # untimed prompt evaluations in front of the steady-state measurements (main())


def shape_dims():
    return synth.FALCON_SHAPES[SHAPE] if SHAPE.startswith("falcon") else synth.LLAMA_SHAPES[SHAPE]


def _fingerprint(path):
    """Size + SHA-256 of the first and last MiB (header, tensor table, tail of the data): cheap, and it changes with the shape, the
    tensor-type mix and the generator."""
    h = hashlib.sha256()
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        h.update(f.read(1 << 20))
        if size > (2 << 20):
            f.seek(size - (1 << 20))
            h.update(f.read(1 << 20))
    return dict(size=size, sha256_head_tail=h.hexdigest(), generator=GEN_VERSION)


def ensure_model():
    """The synthetic model file lives outside the repo (4 GB).  A file found at MODEL is accepted only with its stamp (written
    next to it by the run that generated it: size, head/tail hash, generator version); anything else is regenerated.
    Returns True when a cached file was reused."""
    stamp = MODEL + ".stamp.json"
    if os.path.exists(MODEL) and os.path.exists(stamp):
        try:
            if json.load(open(stamp)) == _fingerprint(MODEL):
                return True
        except (OSError, ValueError):
            pass
    tmp = MODEL + ".tmp%d" % os.getpid()
    if SHAPE.startswith("falcon"):
        synth.write_falcon_gguf(tmp, SHAPE, FTYPE, seed=1234, quantizer=QUANTIZER)
    else:
        synth.write_llama_gguf(tmp, SHAPE, FTYPE, seed=1234, quantizer=QUANTIZER)
    os.replace(tmp, MODEL)
    json.dump(_fingerprint(MODEL), open(stamp, "w"))
    return False


def _cpu_worker(threads, n_vocab):
    """One reference-build run (child process of cpu_baseline): prints {"decode": tok/s, "prefill": tok/s}."""
    from oracle import ref
    r = ref.open_llm(MODEL, context_length=N_CTX, batch_size=N_PROMPT, threads=threads)
    prompt = synth.prompt_tokens(N_PROMPT, n_vocab)
    r.eval(prompt[:8])                       # throw-away: first touch of the mmap'ed weights
    r._context = []
    t0 = time.perf_counter()
    r.eval(prompt)
    t_prefill = time.perf_counter() - t0
    ts = []
    for _ in range(24):
        tok = r.sample(top_k=1, repetition_penalty=1.0)
        t0 = time.perf_counter()
        r.eval([tok])
        ts.append(time.perf_counter() - t0)
    print(json.dumps(dict(decode=round(1.0 / float(np.median(ts)), 3), prefill=round(N_PROMPT / t_prefill, 2))), flush=True)


def cpu_baseline(n_vocab):
    """BASELINE.md §3 on a bounded sample: the reference CPU build on this box's host cores — 128-token prefill with
    batch_size=128, then 24 greedy decode steps (median), with every core and with the library default (threads=-1 -> cores / 2,
    models/llm.h:129-130).  Each run is a child process with a time limit: ggml's spinning thread pool can take minutes with
    hundreds of threads on a many-core host, and this must stay a bounded sample — a run that does not finish is retried with
    fewer threads and reported as such.  The headline `value` is the best decode rate found (its thread count in `threads`).
    When oracle/_ref did not travel with the snapshot, the scalar C restatement (oracle/mirror.c) is timed instead on a smaller
    sample and reported as kind "port"."""
    import subprocess
    from oracle import mirror, ref
    cores = os.cpu_count() or 1
    if ref.available():
        def run(threads, limit):
            try:
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(threads), "--cpu-vocab", str(n_vocab),
                                    "--config", str(CONFIG)],
                                   capture_output=True, text=True, timeout=limit)
                lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
                return json.loads(lines[-1]) if lines else None
            except subprocess.TimeoutExpired:
                return None
        tried, best = [], None
        # thread counts in the region where ggml's pool scales (its barriers spin: hundreds of threads are slower than a few dozen),
        # then every core and the library default for the record, each under its own limit
        plan = [(t, 45) for t in (16, 32, 8) if t <= cores] or [(cores, 45)]
        if cores > 32:
            plan.append((cores, 25))
        for t, limit in plan:
            r = run(t, limit)
            tried.append(dict(threads=t, decode_tok_s=r["decode"] if r else None, prefill_tok_s=r["prefill"] if r else None,
                              note=None if r else "did not finish within the %d s limit" % limit))
            if r and (best is None or r["decode"] > best[1]["decode"]):
                best = (t, r)
        rd = run(-1, 30)
        if best is None:
            return dict(value=None, unit="tokens/s", cores=cores, kind="reference", runs=tried, sample="no reference run finished within its limit")
        return dict(value=best[1]["decode"], unit="tokens/s", cores=cores, threads=best[0], kind="reference", prefill_tok_s=best[1]["prefill"],
                    runs=tried, default_threads=dict(threads=max(1, cores // 2), value=rd["decode"] if rd else None,
                                                     prefill_tok_s=rd["prefill"] if rd else None,
                                                     note=None if rd else "did not finish within the 30 s limit"),
                    sample="reference AVX2 build (oracle/_ref: gcc -O3 -mavx2 -mfma -mf16c, the reference's own CT_INSTRUCTIONS=avx2 flags), same "
                           "synthetic file, host has %d cores: 128-token prefill (batch_size=128) then 24 greedy decode steps, median step "
                           "time; value = threads=%d, the best of the thread counts in `runs`" % (cores, best[0]))
    if not mirror.available():
        return None
    o = mirror.MirrorLlama(MODEL, N_CTX)
    lg = o.eval(synth.prompt_tokens(2, n_vocab), 0)
    t0 = time.perf_counter()
    o.eval([int(np.argmax(lg))], 2)
    dt = time.perf_counter() - t0
    return dict(value=round(1.0 / dt, 4), unit="tokens/s", cores=1, kind="port",
                sample="scalar C restatement (oracle/mirror.c), 1 thread, same synthetic file, 2-token prefill then ONE decode step")


def other_configs():
    """The other single-GPU-capable BASELINE configs, each as a child `bench.py --config N` run (own process, own model file):
    config 3 (Llama-2-7B Q8_0, 7 GB: seconds to synthesise) always, configs 4 / 5 (Falcon-40B Q4_K_M 25 GB, Llama-2-70B Q5_K_M 49 GB on
    ONE GPU) when CTAMD_BENCH_BIG=1.  Reported, not the headline."""
    import shutil
    import subprocess
    big = os.environ.get("CTAMD_BENCH_BIG")
    if big is None:   # default: on, when the scratch disk takes the 49 GB file (one big file at a time; pooled synthesis: tens of seconds each)
        big = "1" if shutil.disk_usage(os.path.dirname(CONFIGS[5][2])).free > 53e9 else "0"
    todo = [(3, 240)] + ([(4, 600), (5, 900)] if big == "1" else [])
    res = []
    for cfg, limit in todo:
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", str(cfg), "--no-cpu-baseline", "--no-other-configs", "--steps", "64"],
                               capture_output=True, text=True, timeout=limit)
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            d = json.loads(lines[-1]) if lines else None
        except subprocess.TimeoutExpired:
            d = None
        if d is None:
            res.append(dict(config=cfg, note="did not finish within %d s" % limit))
            continue
        res.append(dict(config=cfg, workload=d["config"]["workload"], decode_tok_s=d["value"], ms_per_step=d["ms_per_step"], steps=d["steps"],
                        prefill_tok_s=d["prefill_tok_s"], prefill_2k_tok_s=d.get("prefill_2k_tok_s"), decode_tok_s_at_2k=d.get("decode_tok_s_at_2k"),
                        prefill_fast_tok_s=(d.get("prefill_fast") or {}).get("tok_s"), prefill_fast_2k_tok_s=(d.get("prefill_fast") or {}).get("tok_s_2k"),
                        decode_fast_attn_tok_s_at_2k=(d.get("prefill_fast") or {}).get("decode_tok_s_at_2k"),
                        prefill_fast_logits_rel_diff=(d.get("prefill_fast") or {}).get("logits_rel_diff_vs_default"),
                        load_s=d["load_s"], frac_of_8TBps_per_token=d["token_roofline"]["frac_of_8TBps"],
                        bytes_per_token=d["token_roofline"]["bytes_per_token"], model_cached=d["config"]["model_cached"]))
        if (cfg in (4, 5) or big == "1") and os.environ.get("CTAMD_BENCH_KEEP_BIG") != "1":   # 7 / 25 / 49 GB of scratch disk: one file at a time, none left behind
            for f in (CONFIGS[cfg][2], CONFIGS[cfg][2] + ".stamp.json"):
                try:
                    os.remove(f)
                except OSError:
                    pass
    return res


def fast_prefill(prompt, n_vocab, exact_logits, exact_greedy, pf_flop, long_context):
    """The order-free prompt kernels (CT_AMD_PREFILL=fast: ctransformers_amd/csrc/kernels_mm8.h) on fresh handles of the same file: steady-state prompt
    rates at 128, 512 and 2048 tokens, and how far their results are from the default (bit-identical) kernels' — logits of the 128-token prompt and the
    greedy continuation.  Opt-in: the default keeps the reference's bits (DESIGN.md 5b says why 1e-3 is out of reach for any other summation order)."""
    os.environ["CT_AMD_PREFILL"] = "fast"
    os.environ["CT_AMD_DECODE_ATTN"] = "fast"   # (only the first handle below decodes beyond 1024 positions: the order-free decode attention's rate at 2k)
    d2k = None
    try:
        h = LLM(MODEL, config=Config(context_length=N_CTX_2K if long_context else 640, batch_size=2048, gpu_layers=1000))
        rates = {}
        for n in (128, 512) + ((N_PROMPT_2K,) if long_context else ()):
            p = synth.prompt_tokens(n, n_vocab)
            h._context = []
            h.eval(p)
            h._context = []
            h.eval(p)
            h._context = []
            t0 = time.perf_counter()
            h.eval(p)
            rates[n] = round(n / (time.perf_counter() - t0), 1)
        if long_context:   # token steps behind the 2048-token prompt, as `decode_tok_s_at_2k` measures them for the default kernels
            tk = h.sample(top_k=1, repetition_penalty=1.0)
            for _ in range(4):
                h.eval([tk])
                tk = h.sample(top_k=1, repetition_penalty=1.0)
            t0 = time.perf_counter()
            for _ in range(N_DECODE_2K):
                h.eval([tk])
                tk = h.sample(top_k=1, repetition_penalty=1.0)
            d2k = round(N_DECODE_2K / (time.perf_counter() - t0), 2)
        del h
        h = LLM(MODEL, config=Config(context_length=N_CTX, batch_size=N_PROMPT, gpu_layers=1000))   # the headline's own handle shape for the comparison
        h.eval(prompt)
        t = h.sample(top_k=1, repetition_penalty=1.0)
        lg = np.array(h.logits.to_numpy(), copy=True)
        same = 0
        for want in exact_greedy:
            if int(t) != want:
                break
            same += 1
            h.eval([t])
            t = h.sample(top_k=1, repetition_penalty=1.0)
        del h
    finally:
        os.environ.pop("CT_AMD_PREFILL", None)
        os.environ.pop("CT_AMD_DECODE_ATTN", None)
    tops = round(rates[128] * pf_flop / 1e12, 1) if pf_flop else None
    tops_512 = round(rates[512] * pf_flop / 1e12, 1) if pf_flop else None
    return dict(tok_s=rates[128], tok_s_512=rates[512], tok_s_2k=rates.get(N_PROMPT_2K), chunk_tokens=512, opt_in="CT_AMD_PREFILL=fast",
                decode_tok_s_at_2k=d2k, decode_opt_in="CT_AMD_DECODE_ATTN=fast (attn_decode9_free_kernel: the V*P steps of a channel split over the workgroup's waves)",
                kernel="mm8_kernel<TYPE,NTT,KS> (v_mfma_i32_32x32x32_i8: Q4_K / Q5_K scales as two int8 digit planes inside the accumulation, Q6_K masked K-chunks, "
                       "Q8_0 float scales; Q8_K / Q8_0 activations as the reference quantizes them; f32 sums in free order) + attn_mm_kernel<HD> "
                       "(v_mfma_f32_32x32x16_f16 for K.Q and V.P, the reference's fp16 / f32 roundings, exact max and double sum)",
                tops=tops, tops_512=tops_512, mfma_i8_peak_tops=5000, frac=round(tops / 5000.0, 4) if tops else None,
                frac_of_f16_peak=round(tops / 2500.0, 4) if tops else None, frac_512_of_f16_peak=round(tops_512 / 2500.0, 4) if tops_512 else None,
                logits_rel_diff_vs_default=float(np.abs(lg - exact_logits).max() / np.abs(exact_logits).max()),
                greedy_steps_compared=len(exact_greedy), greedy_steps_identical=same,
                note="same quantization points and exact integer dots as the reference, another f32 summation order: K / V rows of the first layer within an fp16 ulp, "
                     "logits within the reference's own int8 quantization noise (~4e-2 of the largest logit on this synthetic model; the reference CPU build differs from ITSELF "
                     "by 3e-2 when its batch_size changes: profiles/r06_reference_self_difference.txt), NOT within 1e-3 — hence opt-in; "
                     "MFMA-busy counters: profiles/r06_mm8_pmc_128tok.txt, profiles/r06_mm8_pmc_512tok.txt")


def stage_ranges(llm):
    import ctypes
    L = llm._lib
    L.ctamd_n_stages.restype, L.ctamd_n_stages.argtypes = ctypes.c_int, [ctypes.c_void_p]
    L.ctamd_stage_range.restype = ctypes.c_int
    L.ctamd_stage_range.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    out = []
    for s in range(L.ctamd_n_stages(llm._llm)):
        a, b = ctypes.c_int(), ctypes.c_int()
        if L.ctamd_stage_range(llm._llm, s, ctypes.byref(a), ctypes.byref(b)) == 0:
            out.append([a.value, b.value])
    return L.ctamd_n_stages(llm._llm), out


def visible_gpus():
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=N_DECODE)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the child runs of configs 3, 4 and 5")
    ap.add_argument("--no-long-context", action="store_true", help="skip the 2048-token prompt / decode-at-2k measurement")
    ap.add_argument("--no-fast-prefill", action="store_true", help="skip the order-free prompt kernels' numbers (prefill_fast)")
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configs[N-1] (default 2: the headline)")
    ap.add_argument("--cpu-worker", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-vocab", type=int, default=32000, help=argparse.SUPPRESS)
    a = ap.parse_args()
    global SHAPE, FTYPE, MODEL, GEN_VERSION, CONFIG
    CONFIG = a.config
    if a.config != 2:
        SHAPE, FTYPE, MODEL = CONFIGS[a.config]
        GEN_VERSION = "synth-%s:%s:%s:seed1234" % (_QTAG, SHAPE, FTYPE)
    if a.cpu_worker is not None:
        _cpu_worker(a.cpu_worker, a.cpu_vocab)
        return 0
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = max(1, a.gpus)
    devices = os.environ.get("CTAMD_BENCH_DEVICES", str(n_gpus) if n_gpus > 1 else "")   # e.g. "0,0": two stages on one GPU (1-GPU box)
    group = None
    rccl_ranks = None
    if world > 1:
        if os.environ.get("CTAMD_FORCE_RCCL_PIPELINE") == "1" or (n_gpus > 1 and not os.environ.get("CTAMD_BENCH_DEVICES") and visible_gpus() < n_gpus):
            from tools import rccl_pipeline as pipeline   # one process per GPU, RCCL point-to-point hand-off
            return pipeline.bench_main(a, MODEL, SHAPE, FTYPE)
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # every rank joins ONE RCCL all-reduce on its own GPU (how many ranks RCCL saw is reported as config.rccl_ranks); the data path of
        # the in-process pipeline itself needs no collective: rank 0 drives every stage, the hand-off is in-stream (csrc/pipeline.cc)
        local = int(os.environ.get("LOCAL_RANK", str(rank)))
        if torch.cuda.is_available() and torch.cuda.device_count() > local:
            try:
                torch.cuda.set_device(local)
                dist.init_process_group(backend="nccl", rank=rank, world_size=world)
                one = torch.ones(1, device="cuda:%d" % local)
                dist.all_reduce(one)
                torch.cuda.synchronize()
                rccl_ranks = int(one.item())
                dist.destroy_process_group()
            except Exception as e:   # noqa: BLE001 — reported, never fatal: the bench itself runs without RCCL
                rccl_ranks = "failed: %s" % str(e)[:120]
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        group = dist

    def barrier():
        if group is not None:
            group.barrier()

    if rank != 0:   # rank 0 drives every stage (in-process pipeline); the others only keep the contract's barriers
        for _ in range(5):
            barrier()
        group.destroy_process_group()
        return 0

    cached = ensure_model()
    barrier()
    if devices:
        os.environ["CT_AMD_DEVICES"] = devices
    t0 = time.perf_counter()
    llm = LLM(MODEL, config=Config(context_length=N_CTX, batch_size=N_PROMPT, gpu_layers=1000))
    load_s = time.perf_counter() - t0
    n_stages, ranges = stage_ranges(llm)
    n_vocab = llm.vocab_size
    prompt = synth.prompt_tokens(N_PROMPT, n_vocab)
    # prefill (timed separately; reported, not the headline value): once cold (first use of every kernel: code-object load,
    # LDS opt-ins), a second pass (graph capture), then the same prompt again from position 0 — the steady-state number
    t0 = time.perf_counter()
    llm.eval(prompt)
    prefill_cold_s = time.perf_counter() - t0
    llm._context = []
    llm.eval(prompt)            # second pass: the library captures the hipGraph of each chunk shape on its second use
    # settle: SETTLE_S seconds of untimed prompt evaluations before anything steady-state is timed.  The first process on a fresh box measured 4 - 6 % low
    # once in six runs of round 6 (decode 720 against 752 - 768 tok/s, the same box 767 minutes later: profiles/r06_bench_default_last_run.json) with only
    # ~45 ms of GPU work in front of the timed loop; the W warm-up steps of the contract follow as before.
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < SETTLE_S:
        llm._context = []
        llm.eval(prompt)
    llm._context = []
    t0 = time.perf_counter()
    llm.eval(prompt)
    prefill_s = time.perf_counter() - t0
    tok = llm.sample(top_k=1, repetition_penalty=1.0)
    exact_logits = np.array(llm.logits.to_numpy(), copy=True)   # (the order-free prompt kernels are compared with these below)
    exact_greedy = [int(tok)]
    for _ in range(a.warmup):
        llm.eval([tok])
        tok = llm.sample(top_k=1, repetition_penalty=1.0)
        exact_greedy.append(int(tok))
    steps = min(a.steps, N_CTX - N_PROMPT - a.warmup - 1)
    # llm.eval() returns only after the library synchronised its stream(s); the logits stay in HBM (lazy outputs: they cross the
    # bus when ctransformers_llm_logits_data is called) and a greedy sample() returns the device-side first-maximum — 4 bytes per
    # step.  That is the reference's generate() loop (eval + sample, ctransformers/llm.py:503-540); the wall clock below brackets
    # exactly `steps` complete decode steps (device sync on both sides).
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        llm.eval([tok])
        tok = llm.sample(top_k=1, repetition_penalty=1.0)
    dt = time.perf_counter() - t0
    barrier()
    tok_s = steps / dt

    issue = None
    if n_stages > 1:   # host time of the one issuing thread per stage and decode step (csrc/pipeline.cc:eval_stages) against the step itself
        import ctypes
        f = llm._lib.ctamd_stage_issue_us
        f.restype, f.argtypes = ctypes.c_double, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong)]
        ev = ctypes.c_longlong(0)
        us = [f(llm._llm, s_, ctypes.byref(ev)) for s_ in range(n_stages)]
        issue = dict(evals=int(ev.value), us_per_eval_by_stage=[round(u / max(1, ev.value), 2) for u in us],
                     us_per_eval_total=round(sum(us) / max(1, ev.value), 2), step_us=round(dt / steps * 1e6, 1),
                     note="averaged over every eval of the handle (three prompt passes + decode steps); the issuing thread binds when its total approaches step_us")
    import ctypes as _ct2
    llm._lib.ctamd_qa_launches.restype, llm._lib.ctamd_qa_launches.argtypes = _ct2.c_longlong, [_ct2.c_void_p]
    fused_qa = int(llm._lib.ctamd_qa_launches(llm._llm)) > 0
    llm._lib.ctamd_spec_hits.restype, llm._lib.ctamd_spec_hits.argtypes = _ct2.c_longlong, [_ct2.c_void_p, _ct2.POINTER(_ct2.c_longlong)]
    _sl = _ct2.c_longlong(0)
    spec_hits = int(llm._lib.ctamd_spec_hits(llm._llm, _ct2.byref(_sl)))
    sites = measure.profile_sites(llm._lib, llm._llm, 8)
    roof = measure.roofline(sites, fused=fused_qa)
    wbytes = synth.weight_bytes_per_token(MODEL)
    hd = shape_dims()   # K + V rows of every layer, fp16, at the average position of the timed steps
    kv_avg = 2 * hd["n_layer"] * (N_PROMPT + a.warmup + steps / 2.0) * (hd["n_embd"] // hd["n_head"] * hd["n_head_kv"]) * 2
    import ctypes as _ct
    llm._lib.ctamd_handoff.restype, llm._lib.ctamd_handoff.argtypes = _ct.c_char_p, [_ct.c_void_p]
    handoff_mode = llm._lib.ctamd_handoff(llm._llm).decode()
    HANDOFF = {"none": "none (one stage)",
               "flag": "in-stream: a kernel on the producer's stream stores the [tokens][n_embd] f32 rows into the next stage's peer-mapped buffer (xGMI) and publishes a "
                       "sequence number; the next stage's stream waits on it with hipStreamWaitValue32 (csrc/pipeline.cc) — no event, no SDMA copy, no RCCL call on this path",
               "stream": "the stages share ONE device and ONE stream: a kernel stores the [tokens][n_embd] f32 rows into the next stage's buffer, stream order is the "
                         "hand-off (no event, no wait; stages on distinct devices take the flag form)",
               "event": "hipMemcpyPeerAsync of the [tokens][n_embd] f32 rows + event record / stream wait (CT_AMD_HANDOFF=event: the round-4 form)"}[handoff_mode]
    par = "1 GPU" if n_stages == 1 else "pp%d in-process (one stage per device, hand-off: %s)" % (n_stages, handoff_mode)
    out = dict(metric="decode_tokens_per_s", value=round(tok_s, 2), unit="tokens/s", n_gpus=n_gpus, steps=steps, warmup=a.warmup,
               ms_per_step=round(dt / steps * 1e3, 4), higher_is_better=True, scaling="weak" if n_gpus == 1 else "strong", vs_baseline=None,
               dtype="int8 dot products, f32 accumulation chain (bit-identical to the reference CPU build)",
               data=("synthetic (random-init weights at the real shapes and tensor-type mix; quantized blocks drawn from a pool of 8192 per type "
                     + ("produced by the reference's own ggml_quantize_chunk (oracle/_ref, BASELINE.md 3)" if QUANTIZER else
                        "produced by this repo's numpy quantizer, tools/synth.py — oracle/_ref did not travel, so not ggml_quantize_chunk")
                     + "; synthetic prompt tokens)"),
               config=dict(workload=("Llama-2-7B GGUF Q4_K_M, all layers on %d x MI355X, 128-tok prefill + %d warm-up + %d timed greedy decode steps (positions %d..%d; BASELINE "
                                     "configs[1] decodes 256: --steps 256 --warmup 0), ctx 512" % (n_gpus, a.warmup, steps, N_PROMPT + a.warmup, N_PROMPT + a.warmup + steps - 1))
                           if SHAPE == "llama-2-7b" and FTYPE == "Q4_K_M" else "BASELINE config %d: %s %s, all layers on %d x MI355X, 128-tok prefill + %d warm-up + %d timed greedy decode steps, ctx 512" % (a.config, SHAPE, FTYPE, n_gpus, a.warmup, steps),
                           shape=SHAPE, ftype=FTYPE, n_prompt=N_PROMPT, parallelism=par, stages=n_stages, layer_ranges=ranges,
                           devices=os.environ.get("CT_AMD_DEVICES", "0"), ranks=world, model_cached=cached,
                           handoff=HANDOFF, rccl_ranks=rccl_ranks, settle_s=SETTLE_S,
                           decode_form=dict(launches_per_layer=4 if fused_qa else 5, fused_qkv_attention=fused_qa,
                                            greedy_chain=dict(steps_served_by_a_queued_step=spec_hits, steps_queued_ahead=int(_sl.value),
                                                              note="the timed loop is eval + sample(top_k=1): after a device-side greedy pick the engine queues the "
                                                                   "next token step behind the one it waits for; every step is computed in full, one more than asked "
                                                                   "for at the end (CT_AMD_SPEC=0 turns it off)"))),
               prefill_tok_s=round(N_PROMPT / prefill_s, 1), prefill_cold_tok_s=round(N_PROMPT / prefill_cold_s, 1), load_s=round(load_s, 2),
               token_roofline=dict(bytes_per_token=int(wbytes + kv_avg), frac_of_8TBps=round(tok_s * (wbytes + kv_avg) / measure.HBM_PEAK, 4),
                                   note="one sequence: the stages of a pipeline are serial, the denominator is ONE GPU's HBM"),
               roofline=roof)
    # prompt chunks: 2 ops per weight of the 2-D matrices per token, against the dense int8 MFMA
    # floor of the guide; the bound is VALU issue (the exact f32 chain step per block, AVX lane, row and token), not MFMA
    pf_flop = 2 * 6.607e9 if SHAPE == "llama-2-7b" else None
    kq = FTYPE.endswith("_K_M")
    out["prefill"] = dict(tok_s=out["prefill_tok_s"], cold_tok_s=out["prefill_cold_tok_s"], chunk_tokens=128,
                          kernel=("matmul_pg_kernel<TYPE,TG,8,GU> (exact integer sums on v_mfma_f32_16x16x32_f16, f32 chain on VALU)" if kq
                                  else "matvec_pfm_kernel<GU> (lane sums on v_mfma_i32_4x4x4_16b_i8 with the 1.5*2^23 addend, packed f32 chain, Q8_0 activation images)") + ", one hipGraph per chunk shape",
                          tops=round(out["prefill_tok_s"] * pf_flop / 1e12, 1) if pf_flop else None, mfma_f16_peak_tops=2500,
                          bound=("valu + mfma issue" if kq else "valu issue") + " (the exact f32 chain step per block, AVX lane, row and token)")
    out["prefill"]["frac"] = round(out["prefill"]["tops"] / 2500.0, 4) if out["prefill"]["tops"] else None
    out["prefill"]["frac_note"] = "tops / dense f16 MFMA peak; MFMA-busy counters of the chunk kernels: profiles/r06_prefill_mfma_pmc.txt"
    if n_gpus == 1 and not a.no_long_context:
        # BASELINE configs[4] says "2k-ctx prefill": a 2048-token prompt at context 2304 (batch_size 128, as the 128-token prompt), then 32
        # greedy steps at positions 2048.. — for every config, on a fresh handle (the context length is fixed at load)
        del llm
        llm = None
        h = LLM(MODEL, config=Config(context_length=N_CTX_2K, batch_size=N_PROMPT, gpu_layers=1000))
        p2k = synth.prompt_tokens(N_PROMPT_2K, n_vocab)
        h.eval(p2k)                 # cold + graph capture of the chunk shapes beyond position 128
        h._context = []
        h.eval(p2k)
        h._context = []
        t0 = time.perf_counter()
        h.eval(p2k)
        t2k = time.perf_counter() - t0
        tk = h.sample(top_k=1, repetition_penalty=1.0)
        for _ in range(4):
            h.eval([tk])
            tk = h.sample(top_k=1, repetition_penalty=1.0)
        t0 = time.perf_counter()
        for _ in range(N_DECODE_2K):
            h.eval([tk])
            tk = h.sample(top_k=1, repetition_penalty=1.0)
        td = time.perf_counter() - t0
        out["prefill_2k_tok_s"] = round(N_PROMPT_2K / t2k, 1)
        out["decode_tok_s_at_2k"] = round(N_DECODE_2K / td, 2)
        out["long_context"] = dict(n_prompt=N_PROMPT_2K, context_length=N_CTX_2K, batch_size=N_PROMPT, decode_steps=N_DECODE_2K,
                                   decode_positions="%d..%d" % (N_PROMPT_2K + 4, N_PROMPT_2K + 4 + N_DECODE_2K - 1))
        del h
    if n_gpus == 1 and n_stages == 1 and not a.no_fast_prefill:
        out["prefill_fast"] = fast_prefill(prompt, n_vocab, exact_logits, exact_greedy, pf_flop, not a.no_long_context)
    if issue is not None:
        out["config"]["host_issue"] = issue
    if n_stages > 1:
        # A pipeline evaluates a prompt in micro-batches (CT_AMD_PP_MB; default 64 tokens up to four stages, 32 beyond: stage s works on micro-batch c while stage
        # s - 1 works on c + 1).  Larger micro-batches mean fewer passes over the weights and less overlap: the best size depends on
        # the stage count, so the prompt rate is reported per size (a fresh handle each: the size is read at load).
        out["prefill"]["micro_batch_tokens"] = int(os.environ.get("CT_AMD_PP_MB", "64" if n_stages <= 4 else "32"))   # pipeline.cc's default
        by_mb = {str(out["prefill"]["micro_batch_tokens"]): out["prefill_tok_s"]}
        del llm
        llm = None
        for mb in (32, 64, 128):
            if str(mb) in by_mb:
                continue
            os.environ["CT_AMD_PP_MB"] = str(mb)
            h = LLM(MODEL, config=Config(context_length=N_CTX, batch_size=N_PROMPT, gpu_layers=1000))
            for _ in range(2):
                h._context = []
                h.eval(prompt)
            h._context = []
            t0 = time.perf_counter()
            h.eval(prompt)
            by_mb[str(mb)] = round(N_PROMPT / (time.perf_counter() - t0), 1)
            del h
        os.environ["CT_AMD_PP_MB"] = str(out["prefill"]["micro_batch_tokens"])
        out["prefill"]["tok_s_by_micro_batch"] = by_mb
    if n_gpus == 1 and a.config == 2 and not a.no_other_configs:
        del llm
        llm = None
        out["other_configs"] = other_configs()
    if not a.no_cpu_baseline and n_gpus == 1:
        if llm is not None:
            del llm
        out["cpu_baseline"] = cpu_baseline(n_vocab)
    barrier()
    print(json.dumps(out), flush=True)
    barrier()
    if group is not None:
        group.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
