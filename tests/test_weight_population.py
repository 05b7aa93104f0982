"""Weights as the reference's file population holds them — not only what this repo's numpy quantizers (tools/synth.py) can emit.

* `quantizer="reference"`: every quantized tensor comes out of the reference's OWN `ggml_quantize_chunk` (models/ggml/ggml.c:19319,
  exported by oracle/_ref): K-quant scale searches (k_quants.c:600-760), Q6_K with `iscale = -128.f / max_scale` (k_quants.c:1048-1115) —
  negative block scales in every block and a negative d in half of them, which the numpy quantizer never produces.
* `quantizer="fuzz"`: ARBITRARY block bytes (tools/synth.py:fuzz_blocks) — every 6-bit scale / min pattern, int8 Q6_K scales over the whole
  range, Q8_0 quants of -128, d / dmin of either sign, zero (dmin = 0), fp16 sub-normal.

Both through the decode mat-vec (token steps) AND the prompt-chunk kernels, compared bit for bit with the reference CPU build on the same
file.  CPU runs use the emulator build of the product sources (tests/emu), `-m gpu` runs the HIP library on the MI355X box."""
import os

import numpy as np
import pytest

from tools import gguf as G, synth
from ctransformers_amd.llm import LLM, Config


def _chunk_tokens(m):
    import ctypes
    f = m._lib.ctamd_chunk_tokens
    f.restype, f.argtypes = ctypes.c_longlong, [ctypes.c_void_p]
    return int(f(m._llm))


def _write(p, shape, ftype, quantizer, seed):
    if shape.startswith("falcon"):
        return synth.write_falcon_gguf(p, shape, ftype, seed=seed, quantizer=quantizer)
    return synth.write_llama_gguf(p, shape, ftype, seed=seed, quantizer=quantizer)


def _q6k_signs(path):
    """(fraction of Q6_K blocks holding a negative scale, fraction with a negative d) over the file's Q6_K tensors."""
    f = G.GGUFFile(path)
    neg_sc, neg_d, n = 0, 0, 0
    for name, (shape, t, data) in f.tensors.items():
        if t != G.Q6_K:
            continue
        b = np.asarray(data, dtype=np.uint8).reshape(-1, 210)[:4096]
        neg_sc += int((b[:, 192:208].view(np.int8) < 0).any(axis=1).sum())
        neg_d += int((b[:, 209] & 0x80 != 0).sum())
        n += b.shape[0]
    return (neg_sc / n, neg_d / n) if n else (0.0, 0.0)


def _compare(ref, lib, p, hp, n_prompt, n_decode, threads=8, token_steps_too=True, monkeypatch=None):
    ctx = n_prompt + n_decode + 8
    cfg = dict(context_length=ctx, batch_size=64, threads=threads)
    r = ref.open_llm(p, **cfg)
    toks = synth.prompt_tokens(n_prompt, hp["n_vocab"])
    r.eval(toks)
    want = [np.array(r.logits.to_numpy(), copy=True)]
    emb = np.array(r.embeddings.to_numpy(), copy=True)
    for _ in range(n_decode):
        t = int(want[-1].argmax())
        r.eval([t])
        want.append(np.array(r.logits.to_numpy(), copy=True))
    assert all(np.isfinite(w).all() for w in want), "the test file must keep the reference's logits finite"
    del r
    modes = ("1", "0") if token_steps_too else ("1",)
    for pf in modes:   # "1": the prompt through the chunk kernels; "0": token steps (the decode mat-vec) for the prompt too
        if monkeypatch is not None:
            monkeypatch.setenv("CT_AMD_PF", pf)
        m = LLM(p, config=Config(**cfg), lib=lib) if lib else LLM(p, config=Config(**cfg))
        m.eval(toks)
        assert _chunk_tokens(m) == (n_prompt if pf == "1" else 0)
        got = m.logits.to_numpy()
        assert np.array_equal(got, want[0]), "prompt (CT_AMD_PF=%s): max rel %.3g" % (pf, np.abs(got - want[0]).max() / np.abs(want[0]).max())
        assert np.array_equal(m.embeddings.to_numpy(), emb)
        for i in range(n_decode):
            t = int(want[i].argmax())
            assert m.sample(top_k=1, repetition_penalty=1.0) == t
            m.eval([t])
            got = m.logits.to_numpy()
            assert np.array_equal(got, want[i + 1]), "step %d (CT_AMD_PF=%s): max rel %.3g" % (i, pf, np.abs(got - want[i + 1]).max() / np.abs(want[i + 1]).max())
        del m


# ---- CPU: the emulator build of the product sources -------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,ftype,quantizer", [
    ("llama-tiny", "Q4_K_M", "reference"), ("llama-small", "Q5_K_M", "reference"), ("llama-tiny", "Q6_K", "reference"),
    ("llama-tiny", "Q8_0", "reference"), ("llama-tiny", "Q4_0", "reference"), ("falcon-tiny", "Q4_K_M", "reference"),
    ("falcon-tiny", "Q5_K_M", "reference"),
    ("llama-tiny", "Q4_K_M", "fuzz"), ("llama-tiny", "Q5_K_M", "fuzz"), ("llama-tiny", "Q6_K", "fuzz"), ("llama-tiny", "Q8_0", "fuzz"),
    ("llama-tiny", "Q4_0", "fuzz"), ("falcon-tiny", "Q4_K_M", "fuzz"),
])
def test_emulator_build_on_reference_quantized_and_arbitrary_blocks(ref, emu_lib, tmp_path, monkeypatch, shape, ftype, quantizer):
    p = str(tmp_path / "m.gguf")
    hp = _write(p, shape, ftype, quantizer, seed=31)
    if quantizer == "reference" and ftype in ("Q4_K_M", "Q5_K_M", "Q6_K", "Q4_0"):   # every llama head and the _K_M mixes hold Q6_K
        sc, d = _q6k_signs(p)
        assert sc > 0.9 and 0.2 < d < 0.8, (sc, d)
    _compare(ref, emu_lib, p, hp, n_prompt=13, n_decode=3, threads=2, monkeypatch=monkeypatch)


def test_fuzz_blocks_cover_the_field_ranges():
    rng = np.random.default_rng(5)
    b = synth.fuzz_blocks(4096, G.Q6_K, 0.05, rng)
    sc = b[:, 192:208].view(np.int8)
    assert sc.min() == -128 and sc.max() == 127
    d = b[:, 208:210].copy().view(np.float16).reshape(-1)
    assert np.isfinite(d).all() and (d < 0).any() and (d == 0).any() and (np.abs(d[d != 0]) < 6.2e-5).any()
    b = synth.fuzz_blocks(4096, G.Q4_K, 0.05, rng)
    sc, mn = synth._unpack_scales_k4(b[:, 4:16])
    assert set(np.unique(sc)) == set(range(64)) and set(np.unique(mn)) == set(range(64))
    dm = b[:, 2:4].copy().view(np.float16).reshape(-1)
    assert (dm == 0).any() and (dm < 0).any()
    assert (synth.fuzz_blocks(4096, G.Q8_0, 0.05, rng)[:, 2:].view(np.int8) == -128).any()


def _write_tie_model(p, shape, seed):
    """Activation blocks in which +amax AND -amax occur: the Q8_K quantizer keeps the sign of the FIRST element of largest magnitude
    (k_quants.c:1198-1204), so the order inside a block decides `iscale`.  Embedding rows of {+-1, +-0.5, +-0.25} in F32 behind norm
    gains of exactly 1: every 256-block of layer 0's normalised input holds both signs of its maximum, the first one at random."""
    hp0 = dict(synth.LLAMA_SHAPES[shape])
    rng = np.random.default_rng(seed)
    E, V = hp0["n_embd"], hp0["n_vocab"]
    emb = rng.choice(np.array([1.0, -1.0, 0.5, -0.5, 0.25, -0.25], dtype=np.float32), size=(V, E))
    emb[:, 0::256] = np.where(rng.random((V, (E + 255) // 256)) < 0.5, 1.0, -1.0)   # the block's first element already is +-amax ...
    emb[:, 7::256] = -emb[:, 0::256]                                               # ... and its negative follows
    ones = np.ones(E, dtype=np.float32)
    return synth.write_llama_gguf(p, shape, "Q4_K_M", seed=seed, type_overrides={"token_embd.weight": G.F32},
                                  tensor_data={"token_embd.weight": emb, "blk.0.attn_norm.weight": ones, "blk.0.ffn_norm.weight": ones})


@pytest.mark.parametrize("shape", ["llama-tiny", "llama-small"])
def test_emulator_build_first_of_two_opposite_maxima(ref, emu_lib, tmp_path, monkeypatch, shape):
    p = str(tmp_path / "m.gguf")
    hp = _write_tie_model(p, shape, seed=3)
    _compare(ref, emu_lib, p, hp, n_prompt=13, n_decode=3, threads=2, monkeypatch=monkeypatch)


# ---- GPU: the HIP library at the real widths ---------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape,ftype,quantizer,n_prompt,n_decode", [
    ("llama-7b-2l", "Q4_K_M", "reference", 33, 8),     # the headline mix: Q4_K + Q6_K (attn_v / ffn_down of a `more bits` layer, the head)
    ("llama-7b-2l", "Q5_K_M", "reference", 33, 8),
    ("llama-70b-2l", "Q5_K_M", "reference", 20, 8),    # config 5 widths
    ("falcon-40b-2l", "Q4_K_M", "reference", 20, 8),   # config 4 widths: Q5_K qkv, Q6_K ffn_down of the first layers, Q8_0 head
    ("llama-7b-2l", "Q4_0", "reference", 33, 8),
    ("llama-7b-2l", "Q8_0", "reference", 33, 8),
    ("llama-small", "Q4_K_M", "reference", 20, 24),    # non-pooled: every block its own quantization
    ("llama-7b-2l", "Q4_K_M", "fuzz", 20, 8),
    ("llama-7b-2l", "Q5_K_M", "fuzz", 20, 8),
    ("llama-7b-2l", "Q6_K", "fuzz", 20, 8),
    ("llama-70b-2l", "Q5_K_M", "fuzz", 12, 6),
    ("falcon-40b-2l", "Q4_K_M", "fuzz", 12, 6),
    ("llama-7b-2l", "Q8_0", "fuzz", 20, 8),
    ("llama-7b-2l", "Q4_0", "fuzz", 20, 8),
    ("llama-small", "Q4_K_M", "fuzz", 20, 24),
    ("llama-small", "Q5_K_M", "fuzz", 20, 24),
])
def test_hip_build_on_reference_quantized_and_arbitrary_blocks(ref, tmp_path, monkeypatch, shape, ftype, quantizer, n_prompt, n_decode):
    p = str(tmp_path / "m.gguf")
    hp = _write(p, shape, ftype, quantizer, seed=41)
    if quantizer == "reference" and ftype in ("Q4_K_M", "Q5_K_M", "Q4_0") and shape.startswith("llama"):
        sc, d = _q6k_signs(p)
        assert sc > 0.9 and 0.2 < d < 0.8, (sc, d)
    _compare(ref, None, p, hp, n_prompt, n_decode, threads=16, monkeypatch=monkeypatch)
    os.remove(p)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["llama-7b-2l", "llama-70b-2l"])
def test_hip_build_first_of_two_opposite_maxima(ref, tmp_path, monkeypatch, shape):
    p = str(tmp_path / "m.gguf")
    hp = _write_tie_model(p, shape, seed=5)
    _compare(ref, None, p, hp, 20, 8, threads=16, monkeypatch=monkeypatch)
    os.remove(p)
