"""Host-side text path of the ABI (csrc/host_text.cc) against vectors produced by the reference build (tests/golden/make_golden.py
`tokenizers`): sentencepiece with real score-ordered merges and byte fallback (reference llama.cpp:3030-3042, :3080-3210), falcon's
byte-level BPE (:3228-3388), the legacy gpt2 tokenizer (models/common.cc).  Runs on the emulator build here and on the HIP build on
the GPU box (same host code, the product binary)."""
import json
import os

import pytest

from conftest import GOLDEN, has_gpu
from ctransformers_amd.llm import LLM, Config


def _open(path, lib, model_type=None):
    return LLM(path, model_type, config=Config(context_length=32, batch_size=8, threads=1), lib=lib)


def _check_all(lib):
    spm = json.load(open(os.path.join(GOLDEN, "spm_golden.json")))
    m = _open(os.path.join(GOLDEN, "spm-vocab.gguf"), lib)
    for text, ids in spm["tokenize"].items():
        assert m.tokenize(text) == ids, repr(text)
        assert m.detokenize(ids) == spm["detokenize"][text], repr(text)
    # merges really happened: "hello world" is two pieces, not eleven characters
    assert len(spm["tokenize"]["hello world"]) == 3
    bpe = json.load(open(os.path.join(GOLDEN, "falcon_bpe.json")))
    m = _open(os.path.join(GOLDEN, "falcon-tiny-q4km.gguf"), lib)
    for text, ids in bpe.items():
        assert m.tokenize(text) == ids, repr(text)
    g2 = json.load(open(os.path.join(GOLDEN, "gpt2_host.json")))
    m = _open(os.path.join(GOLDEN, "gpt2-tiny-q40.bin"), lib, "gpt2")
    for text, ids in g2["tokenize"].items():
        assert m.tokenize(text) == ids, repr(text)
    assert m.detokenize([300, 10]) == g2["detok_300_10"]


def test_tokenizers_match_reference_emulator_build(emu_lib):
    _check_all(emu_lib)


@pytest.mark.gpu
def test_tokenizers_and_samplers_match_reference_hip_build():
    """The product binary on the GPU box: tokenizers as above, and both sampler chains replayed on the golden logits (the library's
    logits buffer is written through the ABI's in-place mutation)."""
    import numpy as np
    _check_all(None)   # None: the HIP library
    for name, mt in (("tiny-q4km", None), ("gpt2-tiny-q40", "gpt2")):
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        path = os.path.join(GOLDEN, name + (".bin" if mt else ".gguf"))
        m = LLM(path, mt, config=Config(context_length=96, batch_size=8, threads=1))
        if mt:   # legacy sampler: expectations taken right after the prompt
            exp = json.load(open(os.path.join(GOLDEN, "gpt2_host.json")))["samples"]
            m.eval(list(g["prompt"]))
            for k, p, temp, pen, seed, tok in exp:
                assert m.sample(top_k=int(k), top_p=p, temperature=temp, repetition_penalty=pen, seed=int(seed)) == int(tok)
        else:    # llama chain on the final golden logits
            ctx = list(g["context"])
            m.eval(ctx[:1])
            final = g["logits"][-1]
            buf = m.logits
            for i in range(len(buf)):
                buf[i] = float(final[i])
            m._context = ctx
            for k, p, temp, pen, seed, expect in g["samples"]:
                assert m.sample(top_k=int(k), top_p=float(p), temperature=float(temp), repetition_penalty=float(pen), last_n_tokens=64,
                                seed=int(seed)) == int(expect)
