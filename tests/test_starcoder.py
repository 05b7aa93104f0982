"""StarCoder / GPT-BigCode legacy files: the reference's starcoder loader (models/llms/starcoder.cc) reads the
container, tensors and graph of its gpt2 loader and differs on the host side — the StarChat / fill-in-the-middle markers present
in the vocabulary are split out of the text before the word regex (models/common.h:76-101).  Vectors: tests/golden/make_golden.py
`starcoder-tiny-q80` + `tokenizers`, produced by the reference build with model_type "starcoder", "gpt_bigcode" and "gpt2"."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from ctransformers_amd.llm import LLM, Config

PATH = os.path.join(GOLDEN, "starcoder-tiny-q80.bin")


def _open(lib, model_type, **kw):
    cfg = dict(context_length=96, batch_size=8, threads=1)
    cfg.update(kw)
    return LLM(PATH, model_type, config=Config(**cfg), lib=lib)


def _check(lib, greedy_steps, light=False):
    g = np.load(os.path.join(GOLDEN, "starcoder-tiny-q80.npz"))
    host = json.load(open(os.path.join(GOLDEN, "starcoder_host.json")))
    for mt in ("starcoder", "gpt_bigcode", "gpt2"):
        m = _open(lib, mt)
        assert m.model_type == host[mt]["model_type"] and m.eos_token_id == host[mt]["eos"] == 500 and m.context_length == 96
        for text, ids in host[mt]["tokenize"].items():
            assert m.tokenize(text) == ids, (mt, text)
    # StarChat end marker: an end-of-sequence token wherever special pieces are registered (reference models/llm.h:78-89
    # LLM::IsEosToken; `<|end|>` is token 507 of this vocabulary), not under the plain gpt2 type
    for mt, expect in (("starcoder", True), ("gpt_bigcode", True), ("gpt2", False)):
        m = _open(lib, mt)
        assert m.is_eos_token(500) and not m.is_eos_token(505)
        assert m.is_eos_token(507) is expect, mt
    # the markers really are single tokens under starcoder and are spelled out piece by piece under gpt2
    assert host["starcoder"]["tokenize"]["<|user|>"] == [505] and len(host["gpt2"]["tokenize"]["<|user|>"]) > 1
    assert host["gpt_bigcode"] == dict(host["starcoder"], model_type=host["gpt_bigcode"]["model_type"])
    m = _open(lib, "starcoder")
    assert len(m.logits) == 0
    m.eval(list(g["prompt"]))
    assert np.array_equal(m.logits.to_numpy(), g["logits"][0]) and len(m.embeddings) == 0
    for i, t in enumerate(g["greedy"][:greedy_steps]):
        assert m.sample(top_k=1, repetition_penalty=1.0) == int(t)
        m.eval([int(t)])
        assert np.array_equal(m.logits.to_numpy(), g["logits"][i + 1]), "step %d" % i
    for bs, key in ((8, "long_chunked"),) if light else ((64, "long_one"), (8, "long_chunked")):   # light: the emulator build
        m = _open(lib, "starcoder", batch_size=bs)
        m.eval(list(g["long_prompt"]))
        assert np.array_equal(m.logits.to_numpy(), g[key])


def test_starcoder_on_emulator_build(emu_lib):
    _check(emu_lib, 2, light=True)


def test_starcoder_file_through_the_c_restatement(mirror):
    g = np.load(os.path.join(GOLDEN, "starcoder-tiny-q80.npz"))
    m = mirror.MirrorGpt2(PATH)
    logits = m.eval(g["prompt"], 0)
    assert np.array_equal(logits, g["logits"][0])
    for i, t in enumerate(g["greedy"][:8]):
        assert int(np.argmax(logits)) == int(t)
        logits = m.eval([int(t)], len(g["prompt"]) + i)
        assert np.array_equal(logits, g["logits"][i + 1])


def test_end_marker_is_eos_like_the_reference_build(emu_lib, ref):
    """ctransformers_llm_is_eos_token over the whole vocabulary, against the reference build itself."""
    for mt in ("starcoder", "gpt2"):
        m = _open(emu_lib, mt)
        r = ref.open_llm(PATH, model_type=mt, context_length=96, batch_size=8, threads=1)
        assert [m.is_eos_token(t) for t in range(512)] == [r.is_eos_token(t) for t in range(512)], mt


def test_unknown_legacy_types_are_refused(emu_lib):
    for mt in ("mpt", "gptj", "dollyv2", "replit"):
        with pytest.raises(RuntimeError):
            _open(emu_lib, mt)


@pytest.mark.gpu
def test_starcoder_on_hip_build():
    _check(None, 40)
