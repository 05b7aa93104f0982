"""Multi-GPU layer pipeline (DESIGN.md row e): one process per GPU, each owning a contiguous block of
transformer layers (weights + that block's KV cache resident on it; token embedding on stage 0, final norm + lm_head on
the last stage).  The only exchange is the `[n_tokens, n_embd]` f32 residual stream from stage s to stage s+1 — a
point-to-point send/recv (`torch.distributed`, backend "nccl" = RCCL over the direct xGMI link; "gloo" in the CPU
tests) — plus one int32 (the greedily sampled token) from the last stage back to stage 0 during decode.  There is no
collective on the data path.

What the reference does instead: `gpu_layers` / `tensor_split` split tensors inside ONE process with peer copies per
mat-mul (reference models/ggml/llama.cpp:1938-2070, ggml-cuda.cu:5798-6119).  north_star asks for the layer pipeline.

Numerics: a stage runs exactly the per-layer launch sequence of the single-GPU engine, and the hand-off is a bit copy
of the f32 residual stream, so pipeline results are bit-identical to the single-GPU path for the same chunking
(`micro_batch` here plays the role of the reference's `batch_size`, which its results depend on — DESIGN.md §2).
"""
import ctypes
import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import gguf as G
from ctransformers_amd.llm import load_library


def partition_layers(n_layer, world, head_cost=0.5, embed_cost=0.0):
    """Contiguous layer ranges, one per rank, balancing per-token HBM bytes: the last stage also streams the lm_head
    (about `head_cost` layers' worth of bytes for Llama-2-7B Q4_K_M: 107 MB vs 122 MB per layer)."""
    if world < 1 or n_layer < world:
        raise ValueError("need 1 <= world <= n_layer (got world=%d, n_layer=%d)" % (world, n_layer))
    total = n_layer + head_cost + embed_cost
    bounds, acc, begin = [], 0.0, 0
    for r in range(world):
        left_ranks = world - r
        if r == world - 1:
            end = n_layer
        else:
            target = (total - acc) / left_ranks - (embed_cost if r == 0 else 0.0)
            cnt = int(round(target))
            cnt = max(1, min(cnt, n_layer - begin - (left_ranks - 1)))
            end = begin + cnt
        bounds.append((begin, end))
        acc += (end - begin) + (embed_cost if r == 0 else 0.0)
        begin = end
    return bounds


def model_dims(path):
    f = G.GGUFFile(path)
    arch = f.kv["general.architecture"]
    return dict(arch=arch, n_layer=int(f.kv[arch + ".block_count"]), n_embd=int(f.kv[arch + ".embedding_length"]),
                n_vocab=len(f.kv["tokenizer.ggml.tokens"]))


class HipStage:
    """One pipeline stage over the C library's stage entry points (include/ctransformers_amd_ext.h).  `device` is the
    torch device the hand-off tensors live on; with the HIP build that is cuda:<ordinal>.  (The CPU test-suite passes the
    emulator build of the same sources, whose "device" pointers are host pointers, together with device="cpu".)"""

    def __init__(self, path, layer_begin, layer_end, context_length=512, device="cuda:0", lib=None):
        self._lib = lib if lib is not None else load_library()
        L = self._lib
        L.ctamd_stage_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.ctamd_stage_create.restype = ctypes.c_void_p
        L.ctamd_stage_eval.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p]
        L.ctamd_stage_eval.restype = ctypes.c_int
        L.ctamd_stage_eval_batched.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.ctamd_stage_eval_batched.restype = ctypes.c_int
        for name in ("ctamd_n_layer", "ctamd_n_embd"):
            getattr(L, name).argtypes = [ctypes.c_void_p]
            getattr(L, name).restype = ctypes.c_int
        L.ctransformers_llm_logits_data.argtypes = [ctypes.c_void_p]
        L.ctransformers_llm_logits_data.restype = ctypes.POINTER(ctypes.c_float)
        L.ctransformers_llm_vocab_size.argtypes = [ctypes.c_void_p]
        L.ctransformers_llm_vocab_size.restype = ctypes.c_int
        L.ctransformers_llm_delete.argtypes = [ctypes.c_void_p]
        self.device = torch.device(device)
        ordinal = self.device.index if self.device.type == "cuda" and self.device.index is not None else 0
        self._h = L.ctamd_stage_create(os.fsencode(path), int(context_length), int(layer_begin), int(layer_end), ordinal)
        if not self._h:
            raise RuntimeError("failed to create pipeline stage [%d,%d) from '%s'" % (layer_begin, layer_end, path))
        self.n_layer = L.ctamd_n_layer(self._h)
        self.n_embd = L.ctamd_n_embd(self._h)
        self.n_vocab = L.ctransformers_llm_vocab_size(self._h)
        self.first = layer_begin == 0
        self.last = layer_end == self.n_layer

    def forward(self, tokens, n_past, x_in=None, batch=0):
        """tokens: this chunk's ids (stage 0) or just its length as `[0]*n` elsewhere.  Returns the [n, n_embd] hand-off
        tensor, or on the last stage the logits of the chunk's last token as a float32 CPU tensor.  batch > 0: the chunk
        is evaluated as the reference would in batches of that size (the chunk has to start on a batch boundary)."""
        n = len(tokens)
        ids = (ctypes.c_int * n)(*[int(t) for t in tokens])
        x_out = None
        if not self.first:
            if x_in is None or tuple(x_in.shape) != (n, self.n_embd) or x_in.dtype != torch.float32:
                raise ValueError("stage input must be float32 [n_tokens, n_embd]")
            x_in = x_in.contiguous()
        if not self.last:
            x_out = torch.empty((n, self.n_embd), dtype=torch.float32, device=self.device)
        rc = self._lib.ctamd_stage_eval_batched(self._h, ids, n, int(n_past), None if self.first else x_in.data_ptr(),
                                                None if self.last else x_out.data_ptr(), int(batch))
        if rc != 0:
            raise RuntimeError("stage eval failed")
        if not self.last:
            return x_out
        p = self._lib.ctransformers_llm_logits_data(self._h)
        return torch.from_numpy(np.ctypeslib.as_array(p, shape=(self.n_vocab,)).copy())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ctransformers_llm_delete(self._h)
            self._h = None

    def __del__(self):
        self.close()


class Pipeline:
    """SPMD driver: every rank calls the same methods with the same arguments.  `stage` is any object with
    first/last/n_embd and forward(tokens, n_past, x_in) (HipStage in the product; the tests also inject the oracle)."""

    def __init__(self, stage, rank, world, device, group=None, stage_device=None):
        # `device`: where the tensors handed to torch.distributed live (cuda:<local rank> for RCCL, cpu for gloo).
        # `stage_device`: where the stage wants its rows; differs from `device` only when a gloo group drives HIP stages
        # (hand-off staged through host memory: the 1-GPU CI form of the N-GPU run).
        self.stage, self.rank, self.world, self.device, self.group = stage, rank, world, torch.device(device), group
        self.stage_device = torch.device(stage_device) if stage_device is not None else self.device

    def _sync(self):
        if self.stage_device.type == "cuda":
            torch.cuda.current_stream(self.stage_device).synchronize()

    def eval_chunk(self, tokens, n_past, batch=0):
        """One chunk through this rank's stage.  Returns logits on the last rank, None elsewhere.  batch: see prefill."""
        n = len(tokens)
        x_in = None
        if self.rank > 0:
            x_in = torch.empty((n, self.stage.n_embd), dtype=torch.float32, device=self.device)
            dist.recv(x_in, src=self.rank - 1, group=self.group)
            x_in = x_in.to(self.stage_device)
            self._sync()  # the stage runs on the library's own stream: the hand-off must have landed
        out = self.stage.forward(tokens, n_past, x_in, batch) if batch else self.stage.forward(tokens, n_past, x_in)
        if self.rank < self.world - 1:
            out = out.to(self.device)
            dist.send(out, dst=self.rank + 1, group=self.group)
            self._keep = out  # keep the buffer alive until the next call has synchronised
            return None
        return out

    def prefill(self, prompt, n_past=0, micro_batch=32, batch_size=0):
        """Micro-batched prompt evaluation: stage s works on chunk c while stage s-1 already works on c+1.
        By default a micro-batch is evaluated as ONE reference batch (results = the reference with batch_size = micro_batch).
        batch_size > 0 (a divisor of micro_batch) reproduces the reference run with that batch size instead, whatever the
        micro-batch: the pipelining granularity no longer decides the bits."""
        if batch_size and micro_batch % batch_size:
            raise ValueError("micro_batch must be a multiple of batch_size")
        logits = None
        for start in range(0, len(prompt), micro_batch):
            chunk = prompt[start:start + micro_batch]
            logits = self.eval_chunk(chunk, n_past + start, batch_size if batch_size and batch_size < len(chunk) else 0)
        return logits

    def _return_token(self, logits):
        """Greedy sample on the last rank, hand the id back to stage 0 (the only rank that needs it)."""
        if self.world == 1:
            return int(torch.argmax(logits))
        buf = torch.zeros(1, dtype=torch.int32, device=self.device)
        if self.rank == self.world - 1:
            buf[0] = int(torch.argmax(logits))
            dist.send(buf, dst=0, group=self.group)
            return int(buf[0])
        if self.rank == 0:
            dist.recv(buf, src=self.world - 1, group=self.group)
            return int(buf[0])
        return 0

    def generate_greedy(self, prompt, n_new, micro_batch=32):
        """Greedy continuation.  Returns the new token ids on ranks 0 and world-1 (placeholders elsewhere) and, on the
        last rank, the logits behind the final token."""
        logits = self.prefill(prompt, 0, micro_batch)
        out, pos = [], len(prompt)
        for _ in range(n_new):
            tok = self._return_token(logits)
            out.append(tok)
            logits = self.eval_chunk([tok], pos)
            pos += 1
        return out, logits


def init_distributed(backend=None):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    if not dist.is_initialized():
        kw = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local, device


def bench_main(args, model_path, shape, ftype, n_prompt=128, n_ctx=512):
    """`bench.py --gpus N` for N > 1: pipeline decode tokens/s (strong scaling: the model is fixed, layers are sharded)."""
    from . import synth
    rank, world, local, device = init_distributed("nccl")
    if rank == 0 and not os.path.exists(model_path):
        tmp = model_path + ".tmp%d" % os.getpid()
        synth.write_llama_gguf(tmp, shape, ftype, seed=1234)
        os.replace(tmp, model_path)
    dist.barrier()
    dims = model_dims(model_path)
    l0, l1 = partition_layers(dims["n_layer"], world)[rank]
    t0 = time.perf_counter()
    stage = HipStage(model_path, l0, l1, context_length=n_ctx, device=device)
    load_s = time.perf_counter() - t0
    pipe = Pipeline(stage, rank, world, device)
    prompt = synth.prompt_tokens(n_prompt, dims["n_vocab"])
    steps = min(args.steps, n_ctx - n_prompt - args.warmup - 1)

    def barrier_sync():
        torch.cuda.synchronize(device)
        dist.barrier()
        torch.cuda.synchronize(device)

    barrier_sync()
    t0 = time.perf_counter()
    logits = pipe.prefill(prompt, 0, micro_batch=32)
    barrier_sync()
    prefill_s = time.perf_counter() - t0
    pos = n_prompt
    for _ in range(args.warmup):
        tok = pipe._return_token(logits)
        logits = pipe.eval_chunk([tok], pos)
        pos += 1
    barrier_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        tok = pipe._return_token(logits)
        logits = pipe.eval_chunk([tok], pos)
        pos += 1
    barrier_sync()
    dt_local = time.perf_counter() - t0
    from ctransformers_amd import measure
    roof = measure.roofline(measure.profile_sites(stage._lib, stage._h, 8)) if rank == 0 else None  # after the timed region
    dt = torch.tensor([dt_local, prefill_s, load_s], dtype=torch.float64, device=device)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt_s, prefill_s, load_s = [float(v) for v in dt.tolist()]
    if rank == 0:
        wbytes = synth.weight_bytes_per_token(model_path)
        out = dict(metric="decode_tokens_per_s", value=round(steps / dt_s, 2), unit="tokens/s", n_gpus=world, steps=steps,
                   warmup=args.warmup, ms_per_step=round(dt_s / steps * 1e3, 4), higher_is_better=True, scaling="strong",
                   vs_baseline=None, dtype="int8", data="synthetic",
                   config=dict(workload="Llama-2-7B GGUF Q4_K_M, layers pipelined over %d MI355X (RCCL p2p hand-off), "
                                        "128-tok prefill + 256-tok greedy decode, ctx 512" % world,
                               shape=shape, ftype=ftype, n_prompt=n_prompt, parallelism="pp%d" % world,
                               layer_ranges=partition_layers(dims["n_layer"], world)),
                   prefill_tok_s=round(n_prompt / prefill_s, 1), load_s=round(load_s, 2),
                   token_roofline=dict(bytes_per_token=int(wbytes), frac_of_8TBps=round(steps / dt_s * wbytes / 8.0e12, 4),
                                       note="stages are serial for one sequence: the denominator stays ONE GPU's HBM"),
                   roofline=roof, cpu_baseline=None)
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0
