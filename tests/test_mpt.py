"""MPT legacy files (reference models/llms/mpt.cc): quantized wte as row lookup and output head, bias-free
LayerNorms, fused Wqkv with the optional clamp, fp16 K / V memory, ALiBi (ggml.c:12193-12254, contracted to one fma by the
reference build) between the score scale and the mask, GELU MLP.  Vectors: tests/golden/make_golden.py `mpt-tiny-q80`
(6 heads of 64: both slope branches, clip_qkv 0.75), `mpt-tiny128-q40` (heads of 128, no clamp) and `tokenizers`, all produced by
the reference build with model_type "mpt"."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from ctransformers_amd.llm import LLM, Config


def _open(lib, name, **kw):
    cfg = dict(context_length=96, batch_size=8, threads=1)
    cfg.update(kw)
    return LLM(os.path.join(GOLDEN, name + ".bin"), "mpt", config=Config(**cfg), lib=lib)


def _check_model(lib, name, greedy_steps, light=False):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = _open(lib, name)
    assert m.model_type == "mpt" and m.vocab_size == 512 and m.context_length == 96 and len(m.logits) == 0
    m.eval(list(g["prompt"]))
    assert np.array_equal(m.logits.to_numpy(), g["logits"][0]) and len(m.embeddings) == 0
    for i, t in enumerate(g["greedy"][:greedy_steps]):
        assert m.sample(top_k=1, repetition_penalty=1.0) == int(t)
        m.eval([int(t)])
        assert np.array_equal(m.logits.to_numpy(), g["logits"][i + 1]), "step %d" % i
    for bs, key in ((8, "long_chunked"),) if light else ((64, "long_one"), (8, "long_chunked")):   # light: the emulator build
        m = _open(lib, name, batch_size=bs)
        m.eval(list(g["long_prompt"]))
        assert np.array_equal(m.logits.to_numpy(), g[key])
    # token by token (the decode kernels) from an empty context gives what the chunk kernels gave
    n = 10 if light else 20
    m = _open(lib, name, batch_size=1)
    for t in g["long_prompt"][:n]:
        m.eval([int(t)])
    a = m.logits.to_numpy().copy()
    m = _open(lib, name, batch_size=64)
    m.eval([int(t) for t in g["long_prompt"][:n]])
    assert np.array_equal(a, m.logits.to_numpy())


def _check_host(lib):
    host = json.load(open(os.path.join(GOLDEN, "mpt_host.json")))
    g = np.load(os.path.join(GOLDEN, "mpt-tiny-q80.npz"))
    m = _open(lib, "mpt-tiny-q80")
    assert m.eos_token_id == host["eos"] == 511 and m.detokenize([300, 233, 10]) == host["detok"]
    for text, ids in host["tokenize"].items():
        assert m.tokenize(text) == ids, repr(text)
    m.eval(list(g["prompt"]))
    for k, p, temp, pen, seed, tok in host["samples"]:
        assert m.sample(top_k=int(k), top_p=p, temperature=temp, repetition_penalty=pen, seed=int(seed)) == int(tok)
    # context length = min(max_seq_len of the file, context_length or 2048) (mpt.cc:15, :80, :605-607)
    assert LLM(os.path.join(GOLDEN, "mpt-tiny128-q40.bin"), "mpt", lib=lib).context_length == host["ctx_default"] == 2048
    assert _open(lib, "mpt-tiny-q80", context_length=4096).context_length == host["ctx_capped"] == 96


@pytest.mark.parametrize("name", ["mpt-tiny-q80", "mpt-tiny128-q40"])
def test_mpt_on_emulator_build(emu_lib, name):
    _check_model(emu_lib, name, 2, light=True)


def test_mpt_host_path_on_emulator_build(emu_lib):
    _check_host(emu_lib)


@pytest.mark.parametrize("name", ["mpt-tiny-q80", "mpt-tiny128-q40"])
def test_mpt_c_restatement_against_reference_vectors(mirror, name):
    """oracle/mirror.c mir_mpt_eval (the CPU restatement of mpt_eval) reproduces the reference build's logits, token by token
    and for the 45-token prompt as one batch."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = mirror.MirrorMpt(os.path.join(GOLDEN, name + ".bin"), 96)
    logits = m.eval(g["prompt"], 0)
    assert np.array_equal(logits, g["logits"][0])
    for i, t in enumerate(g["greedy"][:12]):
        assert int(np.argmax(logits)) == int(t)
        logits = m.eval([int(t)], len(g["prompt"]) + i)
        assert np.array_equal(logits, g["logits"][i + 1]), "step %d" % i
    m = mirror.MirrorMpt(os.path.join(GOLDEN, name + ".bin"), 96)
    assert np.array_equal(m.eval(g["long_prompt"], 0), g["long_one"])


@pytest.mark.parametrize("ftype", [7, 2])
def test_mpt_heads_of_112(emu_lib, mirror, tmp_path, ftype):
    """MPT-30B's head size (112 = three 32-element steps of ggml_vec_dot_f16 + its scalar tail of 16, ggml.c:2392-2425): prompt through the
    chunk kernels and decode steps against the oracle restatement (pinned to the reference build on this shape when the mirror tests were
    written: tests/golden/make_golden.py has no 112 vectors, the GPU suite compares mpt-30b-2l with the reference build itself)."""
    from tools import synth
    p = str(tmp_path / "m.bin")
    hp = synth.write_mpt_ggml(p, "mpt-tiny112", seed=23, ftype=ftype)
    m = LLM(p, "mpt", config=Config(context_length=48, batch_size=16, threads=1), lib=emu_lib)
    o = mirror.MirrorMpt(p, 48)
    toks = synth.prompt_tokens(5, hp["n_vocab"])
    m.eval(toks)
    lg = np.array(o.eval(toks, 0), copy=True)
    assert np.array_equal(m.logits.to_numpy(), lg)
    t = int(lg.argmax())
    m.eval([t])
    assert np.array_equal(m.logits.to_numpy(), o.eval([t], 5))


def test_mpt_kq_scale_is_the_double_sqrt_form(mirror, ref, tmp_path):
    """mpt.cc:460-462 writes `1.0f / sqrt(float(n_embd) / n_head)`: ::sqrt is the double function, the double quotient becomes a float
    once — one ulp away from 1.0f / sqrtf(.) at a head size of 112 (equal at 64 / 128).  This file (32 heads of 112, two layers, seed 21)
    is one where that ulp moves the logits by 1e-2 relative: the restatement must follow the reference build (needs oracle/_ref)."""
    from tools import synth
    p = str(tmp_path / "m.bin")
    hp = synth.write_mpt_ggml(p, dict(n_vocab=512, max_seq_len=2048, n_embd=3584, n_head=32, n_layer=2, alibi_bias_max=8.0, clip_qkv=0.0), seed=21, ftype=2)
    r = ref.open_llm(p, model_type="mpt", context_length=28, batch_size=64, threads=4)
    o = mirror.MirrorMpt(p, 28)
    toks = synth.prompt_tokens(20, hp["n_vocab"])
    r.eval(toks)
    assert np.array_equal(r.logits.to_numpy(), o.eval(toks, 0))


def test_truncated_and_mistyped_mpt_files_are_refused(emu_lib, tmp_path):
    raw = open(os.path.join(GOLDEN, "mpt-tiny-q80.bin"), "rb").read()
    for cut in (3, 20, 40, 700, len(raw) // 2):
        p = tmp_path / ("cut%d.bin" % cut)
        p.write_bytes(raw[:cut])
        with pytest.raises(RuntimeError):
            LLM(str(p), "mpt", config=Config(context_length=32), lib=emu_lib)
    with pytest.raises(RuntimeError):   # a gpt2 container read with the MPT header
        LLM(os.path.join(GOLDEN, "gpt2-tiny-q40.bin"), "mpt", config=Config(context_length=32), lib=emu_lib)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mpt-tiny-q80", "mpt-tiny128-q40"])
def test_mpt_on_hip_build(name):
    _check_model(None, name, 40)


@pytest.mark.gpu
def test_mpt_host_path_on_hip_build():
    _check_host(None)
