"""Malformed model files must make ctransformers_llm_create return NULL (the Python side raises) — never kill the process; the reference
does the same through its loaders' exceptions (models/llm.cc:36-65).  Plus the ABI corners the advisor listed: reset() on legacy
models, last_n_tokens = 0."""
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN
from ctransformers_amd.llm import LLM, Config


def _try_open(path, lib):
    return LLM(path, config=Config(context_length=32, batch_size=8, threads=1), lib=lib)


def _patched(tmp_path, name, edit):
    raw = bytearray(open(os.path.join(GOLDEN, "tiny-q4km.gguf"), "rb").read())
    raw = edit(raw)
    p = str(tmp_path / name)
    open(p, "wb").write(bytes(raw))
    return p


def _set_u32_kv(raw, key, value):
    k = key.encode()
    i = raw.find(struct.pack("<Q", len(k)) + k)
    assert i >= 0
    off = i + 8 + len(k)
    assert struct.unpack_from("<I", raw, off)[0] == 4   # GGUF type u32
    struct.pack_into("<I", raw, off + 4, value)
    return raw


@pytest.mark.parametrize("case", ["truncated", "huge_tensor_count", "huge_kv_count", "zero_heads", "heads_not_dividing", "huge_dim", "garbage_tail"])
def test_malformed_gguf_is_refused(emu_lib, tmp_path, case):
    def edit(raw):
        if case == "truncated":
            return raw[:len(raw) // 3]
        if case == "huge_tensor_count":
            struct.pack_into("<Q", raw, 8, 1 << 60)
        elif case == "huge_kv_count":
            struct.pack_into("<Q", raw, 16, 1 << 61)
        elif case == "zero_heads":
            _set_u32_kv(raw, "llama.attention.head_count", 0)
        elif case == "heads_not_dividing":
            _set_u32_kv(raw, "llama.attention.head_count", 7)
        elif case == "huge_dim":
            k = b"token_embd.weight"
            i = raw.find(struct.pack("<Q", len(k)) + k)
            struct.pack_into("<Q", raw, i + 8 + len(k) + 4, 1 << 50)   # ne[0]
        elif case == "garbage_tail":
            return raw[:4096] + bytes(np.random.default_rng(1).integers(0, 256, 4096, dtype=np.uint8))
        return raw
    p = _patched(tmp_path, case + ".gguf", edit)
    with pytest.raises(Exception):
        _try_open(p, emu_lib)
    # the process is alive and a good file still loads
    m = _try_open(os.path.join(GOLDEN, "tiny-q4km.gguf"), emu_lib)
    assert m.vocab_size == 512


def test_reset_semantics(emu_lib):
    """reference models/llm.h:106: Reset() forgets the logits of legacy models (sample() then returns EOS, logits_size() 0); GGUF
    models keep theirs."""
    g = np.load(os.path.join(GOLDEN, "gpt2-tiny-q40.npz"))
    m = LLM(os.path.join(GOLDEN, "gpt2-tiny-q40.bin"), "gpt2", config=Config(context_length=96, batch_size=8, threads=1), lib=emu_lib)
    m.eval(list(g["prompt"]))
    assert len(m.logits) == 512
    m.reset()
    assert len(m.logits) == 0 and m.sample(top_k=1) == m.eos_token_id
    m.eval(list(g["prompt"]))
    assert np.array_equal(m.logits.to_numpy(), g["logits"][0])
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    m = LLM(os.path.join(GOLDEN, "tiny-q4km.gguf"), config=Config(context_length=96, batch_size=8, threads=1), lib=emu_lib)
    m.eval(list(g["prompt"]))
    m.reset()
    assert len(m.logits) == 512


def test_last_n_tokens_zero_means_whole_context(emu_lib, ref):
    """`last_n_tokens=0` slices [-0:] in the reference's Python (llm.py:443): the repetition penalty sees the whole context."""
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    path = os.path.join(GOLDEN, "tiny-q4km.gguf")
    m = LLM(path, config=Config(context_length=96, batch_size=8, threads=1), lib=emu_lib)
    r = ref.open_llm(path, context_length=96, batch_size=8, threads=1)
    m.eval(list(g["prompt"]))
    r.eval(list(g["prompt"]))
    for seed in (1, 2, 3):
        assert m.sample(top_k=50, top_p=0.9, temperature=0.9, repetition_penalty=1.5, last_n_tokens=0, seed=seed) == \
            r.sample(top_k=50, top_p=0.9, temperature=0.9, repetition_penalty=1.5, last_n_tokens=0, seed=seed)


def test_greedy_pick_without_fetching_logits_and_edited_logits(emu_lib):
    """A top_k = 1 step without repetition penalty is answered by the device-side first-maximum (4 bytes cross the bus); once the caller
    has looked at the logits — the reference's Python hands out a WRITABLE view — sampling runs on that host copy, edits included."""
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    m = LLM(os.path.join(GOLDEN, "tiny-q4km.gguf"), config=Config(context_length=96, batch_size=8, threads=1), lib=emu_lib)
    m.eval(list(g["prompt"]))
    t1 = m.sample(top_k=1, repetition_penalty=1.0, temperature=0.7, top_p=0.5, seed=3)   # nothing fetched yet: the device-side pick
    assert t1 == int(g["greedy"][0])
    lg = m.logits                      # fetches the outputs
    assert t1 == int(np.argmax(lg.to_numpy())) and np.array_equal(lg.to_numpy(), g["logits"][0])
    lg[t1] = -1e30                     # the caller bans the token in place
    t2 = m.sample(top_k=1, repetition_penalty=1.0)
    assert t2 != t1 and t2 == int(np.argmax(lg.to_numpy()))
    m.eval([t1])                       # a new eval: back to the device-side pick, equal to the reference's next greedy token
    assert m.sample(top_k=1, repetition_penalty=1.0) == int(g["greedy"][1])
    assert m.sample(top_k=0, repetition_penalty=1.3, last_n_tokens=0) == int(g["greedy"][1])   # k <= 1, no tokens to penalise


def test_edge_requests_match_reference_build(emu_lib, ref):
    """The C ABI at its edges, the same raw calls on this library and on the reference build: an empty request, a request that runs
    into the end of the context (the reference clamps n_past to n_ctx - N per batch, models/llm.h:126 — positions are overwritten),
    a batch size larger than the context, and a single token at the last position."""
    import ctypes
    path = os.path.join(GOLDEN, "tiny-q4km.gguf")
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    toks = [int(t) for t in g["long_prompt"]]
    cfg = dict(context_length=32, batch_size=8, threads=1)
    m = LLM(path, config=Config(**cfg), lib=emu_lib)
    r = ref.open_llm(path, **cfg)

    def raw(h, tokens, n_past, batch):
        arr = (ctypes.c_int * max(1, len(tokens)))(*tokens)
        return bool(h.ctransformers_llm_batch_eval(arr, len(tokens), n_past, batch, 1))

    for h in (m, r):
        assert raw(h, toks[:20], 0, 8)
    assert np.array_equal(m.logits.to_numpy(), r.logits.to_numpy())
    for h in (m, r):
        assert raw(h, [], 20, 8)                      # nothing to do: true, outputs untouched
    assert np.array_equal(m.logits.to_numpy(), r.logits.to_numpy())
    for h in (m, r):
        assert raw(h, toks[20:38], 20, 8)             # 18 tokens from position 20 of a 32-position context: the last batches are clamped
    assert np.array_equal(m.logits.to_numpy(), r.logits.to_numpy())
    for h in (m, r):
        assert raw(h, toks[:5], 31, 64)               # batch_size above n_ctx, n_past at the last position
    assert np.array_equal(m.logits.to_numpy(), r.logits.to_numpy())
    for h in (m, r):
        assert raw(h, toks[7:8], 31, 8)               # one token at the last position
    assert np.array_equal(m.logits.to_numpy(), r.logits.to_numpy())
    assert np.array_equal(m.embeddings.to_numpy(), r.embeddings.to_numpy())
