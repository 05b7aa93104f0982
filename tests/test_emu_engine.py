"""The HIP engine's kernel LOGIC, run through the CPU emulation of the HIP subset (tests/emu): same sources as the
product (ctransformers_amd/csrc), compiled with g++ -DCT_EMU.  Checks bit-identity with the golden vectors of the real
reference build, the ABI semantics the reference's Python relies on, and the sampler chain."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from ctransformers_amd.llm import LLM, Config


def open_emu(emu_lib, name, **kw):
    cfg = dict(context_length=96, batch_size=8, threads=1)
    cfg.update(kw)
    if name.startswith("gpt2"):   # legacy GGML container: model_type is required, as with the reference
        return LLM(os.path.join(GOLDEN, name + ".bin"), "gpt2", config=Config(**cfg), lib=emu_lib)
    return LLM(os.path.join(GOLDEN, name + ".gguf"), config=Config(**cfg), lib=emu_lib)


def chunk_tokens(m):
    """Tokens the handle evaluated through the prompt-chunk kernels (include/ctransformers_amd_ext.h)."""
    import ctypes
    f = m._lib.ctamd_chunk_tokens
    f.restype, f.argtypes = ctypes.c_longlong, [ctypes.c_void_p]
    return int(f(m._llm))


@pytest.mark.parametrize("name,steps", [("tiny-q4km", 4), ("tiny-q4km-refq", 4), ("tiny-q5km", 2), ("tiny-q80", 2), ("tiny-q40", 2),
                                        ("falcon-tiny-q4km", 2), ("falcon-tiny7-q4km", 2), ("gpt2-tiny-q40", 2)])
def test_logits_bit_identical_to_reference(emu_lib, name, steps):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = open_emu(emu_lib, name)
    assert m.model_type == name.split("-")[0].replace("tiny", "llama") and m.vocab_size == 512 and m.context_length == 96
    assert len(m.logits) == 0  # nothing evaluated yet
    m.eval(list(g["prompt"]))
    assert np.array_equal(m.logits.to_numpy(), g["logits"][0])
    # every golden model takes the prompt-chunk kernels (the 11 tokens in one pass; batches of 8 + 3 for the arithmetic)
    assert chunk_tokens(m) == len(g["prompt"])
    if name.startswith("gpt2"):
        assert len(m.embeddings) == 0   # legacy models expose no embeddings (reference models/llm.h:73)
    else:
        assert np.array_equal(m.embeddings.to_numpy(), g["embeddings"][0])
    for i in range(steps):
        t = m.sample(top_k=1, repetition_penalty=1.0)
        assert t == int(g["greedy"][i])
        m.eval([t])
        assert np.array_equal(m.logits.to_numpy(), g["logits"][i + 1]), "step %d" % i


def falcon_fold(m):
    import ctypes
    f = m._lib.ctamd_falcon_fold
    f.restype, f.argtypes = ctypes.c_int, [ctypes.c_void_p]
    return int(f(m._llm))


@pytest.mark.parametrize("name", ["falcon-tiny-q4km", "falcon-tiny7-q4km"])
def test_falcon_rope_and_kv_append_in_the_qkv_launch(emu_lib, name, monkeypatch):
    """Round 6: the rows of attn_qkv are reordered at load (NEOX pair (i, i + head_dim / 2) -> rows (2i, 2i + 1)), so the token step's QKV launch rotates, rounds
    to fp16 and appends K / V in its own epilogue (MatvecArgs::rope_neox) — one launch per layer less; prompt chunks read the same reordered rows through
    falcon_rope_store_kernel's `perm` form.  On by default; CT_AMD_FALCON_FOLD=0 and CT_AMD_GPU_REPACK=0 keep the separate launch: the same bits
    (the golden logits of the reference build, llama.cpp:2652-2700)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    for env, want in ((None, 1), ("CT_AMD_FALCON_FOLD", 0), ("CT_AMD_GPU_REPACK", 0)):
        if env:
            monkeypatch.setenv(env, "0")
        m = open_emu(emu_lib, name)
        assert falcon_fold(m) == want
        if want:                                   # three tokens through a chunk (the `perm` form), the rest one by one through token steps (the epilogue)
            m.eval(list(g["prompt"])[:3])
            for t in list(g["prompt"])[3:]:
                m.eval([int(t)])
        else:                                      # the separate launch: the whole prompt as a chunk, then token steps
            m.eval(list(g["prompt"]))
        assert np.array_equal(m.logits.to_numpy(), g["logits"][0])
        for i in range(2):
            t = m.sample(top_k=1, repetition_penalty=1.0)
            assert t == int(g["greedy"][i])
            m.eval([t])
            assert np.array_equal(m.logits.to_numpy(), g["logits"][i + 1])
        # ... and with the rows reordered the token steps take the fused QKV + attention launch (kernels_qa9.h, LayerNorm form): four launches per layer
        assert (qa_launches(m) > 0) == bool(want)
        if env:
            monkeypatch.delenv(env)


def qa_launches(m):
    """Fused QKV + attention launches this handle has issued (kernels_qa9.h)."""
    import ctypes
    f = m._lib.ctamd_qa_launches
    f.restype, f.argtypes = ctypes.c_longlong, [ctypes.c_void_p]
    return int(f(m._llm))


@pytest.mark.parametrize("name", ["tiny-q4km", "tiny-q4km-refq", "tiny-q5km"])
def test_fused_qkv_attention_launch(emu_lib, name, monkeypatch):
    """Token steps of the llama graph with K-quant q / k / v take the fused QKV + attention launch (the test build runs its two phases as
    two passes); with CT_AMD_FUSE_QA=0 the two launches of round 4.  Both: the golden logits, bit for bit, beyond the first 32-step of the
    V*P dot (positions 11 .. 40: the new position in the leftover part and, at 32, as the last element of a 32-step)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    outs = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("CT_AMD_FUSE_QA", fuse)
        m = open_emu(emu_lib, name)
        m.eval(list(g["prompt"]))
        seq = []
        for i in range(30):
            t = m.sample(top_k=1, repetition_penalty=1.0)
            if i < len(g["greedy"]):
                assert t == int(g["greedy"][i])
            m.eval([t])
            if i + 1 < len(g["logits"]):
                assert np.array_equal(m.logits.to_numpy(), g["logits"][i + 1]), "step %d" % i
            seq.append(t)
        n = qa_launches(m)
        assert (n > 0) == (fuse == "1"), n
        outs.append((seq, m.logits.to_numpy().copy()))
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])


def test_batch_structure(emu_lib):
    """One 45-token batch: bit-identical to the reference's one-batch result (the chunks-of-8 form of the same prompt is
    covered through the pipeline in tests/test_pipeline.py::test_gloo_pipeline_world2)."""
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    m = open_emu(emu_lib, "tiny-q4km", batch_size=64)
    m.eval(list(g["long_prompt"]))
    assert np.array_equal(m.logits.to_numpy(), g["long_one"])
    assert chunk_tokens(m) == 45   # 32 + 13 tokens through the chunk kernels


@pytest.mark.parametrize("name", ["tiny-q4km", "tiny-q5km"])
def test_batches_coalesced_into_one_pass(emu_lib, name):
    """batch_eval with the reference's default batch_size 8: the 45-token prompt goes down as ONE eval (chunk kernels over all
    45 tokens) and still equals the reference's batch-by-batch result — which differs from its one-batch result."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = open_emu(emu_lib, name, batch_size=8)
    m.eval(list(g["long_prompt"]))
    assert np.array_equal(m.logits.to_numpy(), g["long_chunked"])
    assert chunk_tokens(m) == 45


@pytest.mark.parametrize("name", ["tiny-q4km"])   # tiny-q5km: tests/test_gpu_parity.py
def test_prompt_chunk_kernels_equal_token_by_token(emu_lib, name, monkeypatch):
    """The chunk kernels (kernels_pg.h) against the decode kernels on the same batches: a full pass of
    CT_AMD_PF_CHUNK tokens plus one on the decode path, a 2-token chunk at a non-zero n_past; logits and embeddings."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    toks = list(g["long_prompt"])[:19]
    monkeypatch.setenv("CT_AMD_PF_CHUNK", "16")
    out = {}
    for pf in ("1", "0"):
        monkeypatch.setenv("CT_AMD_PF", pf)
        m = open_emu(emu_lib, name, batch_size=64)
        res = []
        for lo, hi in ((0, 17), (17, 19)):
            m.eval(toks[lo:hi])
            res.append((m.logits.to_numpy().copy(), m.embeddings.to_numpy().copy()))
        out[pf] = (res, chunk_tokens(m))
    assert out["1"][1] == 16 + 2 and out["0"][1] == 0   # 17 = one 16-token pass + one token on the decode path
    for (la, ea), (lb, eb) in zip(out["1"][0], out["0"][0]):
        assert np.array_equal(la, lb) and np.array_equal(ea, eb)


def test_abi_semantics(emu_lib):
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    m = open_emu(emu_lib, "tiny-q4km")
    prompt = list(g["prompt"])
    # batch_size chunking (reference models/llm.h:40-54): 11 tokens in chunks of 4 == one shot
    m.eval(prompt, batch_size=4)
    assert np.array_equal(m.logits.to_numpy(), g["logits"][0])
    # in-place logits mutation persists across property reads (reference tests/test_model.py:10-16)
    m.logits[3] = 123.5
    assert m.logits[3] == 123.5
    # KV overwrite: roll the Python-side context back by 3 tokens and re-evaluate them -> identical logits
    keep = m._context[:-3]
    redo = m._context[-3:]
    m._context = list(keep)
    m.eval(redo)
    assert np.array_equal(m.logits.to_numpy(), g["logits"][0])
    # prefix reuse path of generate(): only the non-shared suffix is evaluated
    rest = m.prepare_inputs_for_generation(prompt + [int(g["greedy"][0])])
    assert rest == [int(g["greedy"][0])]
    # tokenizer plumbing: byte fallback + BOS (synthetic vocab has only byte tokens for ASCII)
    toks = m.tokenize("hi")
    assert toks[0] == m.bos_token_id == 1 and m.eos_token_id == 2
    assert m.detokenize(toks[1:]) != "" and m.is_eos_token(2) and not m.is_eos_token(5)


def test_sampler_chain_matches_reference(emu_lib):
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    m = open_emu(emu_lib, "tiny-q4km")
    # put the golden final logits into the library's buffer and replay the reference's sampling calls
    ctx = list(g["context"])
    m.eval(ctx[:1])
    final = g["logits"][-1]
    buf = m.logits
    for i in range(len(buf)):
        buf[i] = float(final[i])
    m._context = ctx
    for k, p, temp, pen, seed, expect in g["samples"]:
        got = m.sample(top_k=int(k), top_p=float(p), temperature=float(temp), repetition_penalty=float(pen),
                       last_n_tokens=64, seed=int(seed))
        assert got == int(expect)


def test_bpe_tokenizer_matches_reference(emu_lib):
    """Falcon's byte-level BPE (reference llm_tokenizer_bpe, llama.cpp:3228-3388): ids produced by the reference build for
    these strings are committed in tests/golden/falcon_bpe.json (generated with oracle/_ref on the falcon-tiny vocabulary)."""
    import json
    exp = json.load(open(os.path.join(GOLDEN, "falcon_bpe.json")))
    m = open_emu(emu_lib, "falcon-tiny-q4km")
    for text, ids in exp.items():
        assert m.tokenize(text) == ids, repr(text)
    assert m.tokenize("") == [] and m.eos_token_id == 11 and m.bos_token_id == 11
    assert m.detokenize(m.tokenize("ab cd\n")) == "ab cd\n"


def test_wide_k_rows(emu_lib, mirror, tmp_path):
    """ffn_down with K > 12288 (70B / Falcon-40B class rows; the MAXK = 32768 instantiation of kernels_v9.h): here a
    Q6_K matrix (layer 0 is a use_more_bits layer) with 52 blocks (13 records of four, an uneven split over the waves), checked
    against the oracle.  (Q4_K / Q5_K at K = 28672 / 32768 are covered on hardware: tests/test_gpu_parity.py, 70b-2l / 40b-2l.)"""
    from tools import synth
    p = str(tmp_path / "wide.gguf")
    hp = synth.write_llama_gguf(p, "llama-tiny", "Q5_K_M", seed=31, overrides=dict(n_ff=13312, n_layer=1))
    m = LLM(p, config=Config(context_length=32, batch_size=8, threads=1), lib=emu_lib)
    o = mirror.MirrorLlama(p, 32)
    toks = synth.prompt_tokens(2, hp["n_vocab"])
    m.eval(toks)
    assert np.array_equal(m.logits.to_numpy(), o.eval(toks, 0))


def test_gpt2_tokenizer_and_sampler(emu_lib):
    """Legacy-model host path (reference models/common.h): regex split + longest-piece tokenizer, raw detokenizer and the
    double-precision top-k/top-p sampler; expectations generated with the reference build (tests/golden/make_golden.py)."""
    import json
    exp = json.load(open(os.path.join(GOLDEN, "gpt2_host.json")))
    m = open_emu(emu_lib, "gpt2-tiny-q40")
    for text, ids in exp["tokenize"].items():
        assert m.tokenize(text) == ids, repr(text)
    g = np.load(os.path.join(GOLDEN, "gpt2-tiny-q40.npz"))
    m.eval(list(g["prompt"]))
    for k, p, temp, pen, seed, tok in exp["samples"]:
        assert m.sample(top_k=int(k), top_p=p, temperature=temp, repetition_penalty=pen, seed=int(seed)) == int(tok)
    assert m.eos_token_id == 0 and m.bos_token_id == 0 and m.detokenize([300, 10]) == exp["detok_300_10"]


def test_greedy_fast_path_equals_reference_chain(emu_lib, ref):
    """sample(top_k=1) takes an argmax fast path; it must pick what the reference's chain picks, including exact ties
    (partial_sort keeps the first strict maximum) and repetition penalties.  Needs the reference build (skips without)."""
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    m = open_emu(emu_lib, "tiny-q4km")
    r = ref.open_llm(os.path.join(GOLDEN, "tiny-q4km.gguf"), context_length=96, batch_size=8, threads=1)
    p = list(g["prompt"])
    m.eval(p)
    r.eval(p)
    rng = np.random.default_rng(0)
    for trial in range(12):
        noise = rng.standard_normal(512).astype(np.float32)
        if trial % 3 == 0:
            noise = np.round(noise)   # many exact ties
        for i in range(512):
            m.logits[i] = float(noise[i])   # in-place mutation through the ABI's logits buffer
            r.logits[i] = float(noise[i])
        for pen in (1.0, 1.3, 0.7):
            assert m.sample(top_k=1, repetition_penalty=pen, last_n_tokens=8) == r.sample(top_k=1, repetition_penalty=pen, last_n_tokens=8)
