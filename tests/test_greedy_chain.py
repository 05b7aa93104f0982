"""Greedy chains on the GPU: the head launch picks the token and prepares the next step, the engine queues that step
before it is asked for.  What the caller sees must be the reference's results whatever it does next: feed the pick back (the queued
step is the eval), feed something else, move n_past backwards, fetch or edit the logits, stop."""
import ctypes
import os

import numpy as np
import pytest

from tools import synth
from ctransformers_amd.llm import LLM, Config

pytestmark = pytest.mark.gpu


def spec_counts(m):
    f = m._lib.ctamd_spec_hits
    f.restype, f.argtypes = ctypes.c_longlong, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]
    n = ctypes.c_longlong(0)
    hits = int(f(m._llm, ctypes.byref(n)))
    return hits, int(n.value)


@pytest.mark.parametrize("shape,ftype,n_prompt,n_decode", [
    ("llama-7b-2l", "Q4_K_M", 33, 40),    # Q6_K head of 32000 rows: the pick over 256 workgroups
    ("llama-small", "Q8_0", 20, 50),      # 32-block head launch
    ("llama-70b-2l", "Q5_K_M", 9, 12),
    ("falcon-small", "Q4_K_M", 24, 30),   # LayerNorm head (Q8_0 lm_head), fused-QKV graph
    ("llama-tiny", "Q5_K_M", 5, 50),      # runs to the end of the context: the last step must not be followed by a guess
])
def test_greedy_chain_is_the_reference(ref, tmp_path, shape, ftype, n_prompt, n_decode):
    p = str(tmp_path / "m.gguf")
    hp = (synth.write_falcon_gguf if shape.startswith("falcon") else synth.write_llama_gguf)(p, shape, ftype, seed=33)
    ctx = n_prompt + n_decode + (0 if shape == "llama-tiny" else 6)
    cfg = dict(context_length=ctx, batch_size=64)
    r = ref.open_llm(p, threads=8, **cfg)
    m = LLM(p, config=Config(**cfg))
    toks = synth.prompt_tokens(n_prompt, hp["n_vocab"])
    r.eval(toks)
    m.eval(toks)
    for i in range(n_decode):
        t = m.sample(top_k=1, repetition_penalty=1.0)           # the device-side pick, BEFORE anything is fetched: arms the chain
        a = r.logits.to_numpy()
        assert t == int(a.argmax()), "step %d: pick" % i
        b = m.logits.to_numpy()                                   # fetched while the guessed next step may be running
        assert np.array_equal(a, b), "step %d: logits" % i
        assert np.array_equal(r.embeddings.to_numpy(), m.embeddings.to_numpy()), "step %d: embeddings" % i
        if len(m._context) >= ctx:
            break
        r.eval([t])
        m.eval([t])
    hits, launched = spec_counts(m)
    assert hits >= min(n_decode, ctx - n_prompt) - 2, (hits, launched)   # every step but the first was served by a queued step
    assert launched - hits <= 1
    f = m._lib.ctamd_qa_launches
    f.restype, f.argtypes = ctypes.c_longlong, [ctypes.c_void_p]
    # K-quant files, the llama graph and (round 6: attn_qkv rows reordered at load, LayerNorm form of the kernel) the falcon graph: the fused QKV + attention launch
    assert (int(f(m._llm)) > 0) == (ftype != "Q8_0")


def test_wrong_guesses_and_rollbacks(ref, tmp_path):
    p = str(tmp_path / "m.gguf")
    hp = synth.write_llama_gguf(p, "llama-7b-2l", "Q4_K_M", seed=34)
    cfg = dict(context_length=128, batch_size=64)
    r = ref.open_llm(p, threads=8, **cfg)
    m = LLM(p, config=Config(**cfg))
    toks = synth.prompt_tokens(20, hp["n_vocab"])
    r.eval(toks)
    m.eval(toks)
    rng = np.random.default_rng(5)
    for i in range(30):
        t = m.sample(top_k=1, repetition_penalty=1.0)
        assert t == int(r.logits.to_numpy().argmax())
        mode = (0, 0, 1, 0, 0, 2, 3)[i % 7]   # two plain steps in a row: the second is served by the queued guess
        if mode == 1:      # the caller feeds another token than the pick: the guess is dropped
            t = int(rng.integers(0, hp["n_vocab"]))
        elif mode == 2:    # the caller edits the logits, then samples on the host copy
            lg = m.logits
            lg[t] = -1e30
            t2 = m.sample(top_k=1, repetition_penalty=1.0)
            a = r.logits.to_numpy().copy()
            a[t] = -1e30
            assert t2 == int(a.argmax())
            t = t2
        elif mode == 3:    # rollback: two positions back, replay them, then go on
            back = m._context[-2:]
            m._context = m._context[:-2]
            r._context = r._context[:-2]
            m.eval(back)
            r.eval(back)
            assert np.array_equal(r.logits.to_numpy(), m.logits.to_numpy())
            t = int(r.logits.to_numpy().argmax())
        r.eval([t])
        m.eval([t])
        if mode != 0:   # (fetching the logits makes the next pick a host-side one: plain steps compare the picks only, so that chains form)
            assert np.array_equal(r.logits.to_numpy(), m.logits.to_numpy()), "step %d mode %d" % (i, mode)
    # a prompt right after a greedy pick (the queued guess is simply overtaken)
    t = m.sample(top_k=1, repetition_penalty=1.0)
    more = synth.prompt_tokens(9, hp["n_vocab"])
    r.eval(more)
    m.eval(more)
    assert np.array_equal(r.logits.to_numpy(), m.logits.to_numpy())
    hits, launched = spec_counts(m)
    assert launched > hits > 0


def test_chain_off_equals_chain_on(tmp_path, monkeypatch):
    """CT_AMD_SPEC=0 / CT_AMD_HEAD_FOLD=0 (separate argmax launch, no queued steps): the same tokens and logits."""
    p = str(tmp_path / "m.gguf")
    hp = synth.write_llama_gguf(p, "llama-7b-2l", "Q4_K_M", seed=35)
    toks = synth.prompt_tokens(17, hp["n_vocab"])
    outs = []
    for spec, fold in (("1", "1"), ("0", "1"), ("0", "0")):
        monkeypatch.setenv("CT_AMD_SPEC", spec)
        monkeypatch.setenv("CT_AMD_HEAD_FOLD", fold)
        m = LLM(p, config=Config(context_length=96, batch_size=64))
        m.eval(toks)
        seq, lgs = [], []
        for _ in range(30):
            t = m.sample(top_k=1, repetition_penalty=1.0)
            seq.append(t)
            m.eval([t])
        lgs = m.logits.to_numpy().copy()
        outs.append((seq, lgs, spec_counts(m)))
        del m
    assert outs[0][0] == outs[1][0] == outs[2][0]
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][1], outs[2][1])
    assert outs[0][2][0] >= 28 and outs[1][2] == (0, 0) and outs[2][2] == (0, 0)
