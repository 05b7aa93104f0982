"""Generation-9 decode mat-vec (kernels_v9.h: row-pair units, lane-major LAYOUT_L9 records, quad-DPP chain) through the CPU
emulation of the HIP sources: token-by-token evaluation (CT_AMD_PF=0 keeps prompts off the chunk kernels) against the
golden vectors of the reference build and, for shapes the goldens do not have, against the oracle restatement."""
import ctypes
import os

import numpy as np
import pytest

from conftest import GOLDEN
from tools import synth
from ctransformers_amd.llm import LLM, Config


def kq_launches(lib_handle):
    f = lib_handle.ctamd_kq_launches
    f.restype, f.argtypes = ctypes.c_longlong, []
    return int(f())


@pytest.mark.parametrize("name", ["tiny-q4km", "tiny-q5km", "falcon-tiny-q4km", "falcon-tiny7-q4km"])
def test_matvec9_token_by_token_equals_reference(emu_lib, monkeypatch, name):
    """Every mat-vec of the prompt and of the greedy steps runs on the K-quant decode kernel (two-type launches: attn_v / ffn_down are
    Q6_K in these files; falcon: LayerNorm prologue, GELU and double-residual epilogues)."""
    monkeypatch.setenv("CT_AMD_PF", "0")
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = LLM(os.path.join(GOLDEN, name + ".gguf"), config=Config(context_length=96, batch_size=8, threads=1), lib=emu_lib)
    n0 = kq_launches(m._lib)
    m.eval(list(g["prompt"]))
    assert np.array_equal(m.logits.to_numpy(), g["logits"][0])
    assert np.array_equal(m.embeddings.to_numpy(), g["embeddings"][0])
    per_layer = 4
    heads = 2                                              # one lm_head launch per reference batch (8 + 3 tokens); falcon's Q8_0 head takes the same kernel
    assert kq_launches(m._lib) - n0 == len(g["prompt"]) * 2 * per_layer + heads
    t = m.sample(top_k=1, repetition_penalty=1.0)
    assert t == int(g["greedy"][0])
    m.eval([t])
    assert np.array_equal(m.logits.to_numpy(), g["logits"][1])


@pytest.mark.parametrize("ftype,n_ff,n_embd", [("Q4_K_M", 2816, 256), ("Q5_K_M", 13312, 256), ("Q4_K_M", 1280, 768)])
def test_matvec9_ragged_and_wide_rows(emu_lib, mirror, monkeypatch, tmp_path, ftype, n_ff, n_embd):
    """Rows whose block count is not a multiple of the 4 blocks of a record (11, 52 = 13 records, 3 blocks), rows wider than
    one prologue round, odd unit counts per wave: against the oracle restatement, token by token and over a decode step."""
    monkeypatch.setenv("CT_AMD_PF", "0")
    p = str(tmp_path / "m.gguf")
    hp = synth.write_llama_gguf(p, "llama-tiny", ftype, seed=7, overrides=dict(n_ff=n_ff, n_embd=n_embd, n_layer=2, n_head=4, n_head_kv=2))
    m = LLM(p, config=Config(context_length=32, batch_size=8, threads=1), lib=emu_lib)
    o = mirror.MirrorLlama(p, 32)
    toks = synth.prompt_tokens(3, hp["n_vocab"])
    n0 = kq_launches(m._lib)
    m.eval(toks)
    assert kq_launches(m._lib) > n0
    lg = o.eval(toks, 0)
    assert np.array_equal(m.logits.to_numpy(), lg)
    t = int(lg.argmax())
    m.eval([t])
    assert np.array_equal(m.logits.to_numpy(), o.eval([t], 3))


@pytest.mark.parametrize("case", ["Q4_K_S", "v_q8_0", "k_q4_0_v_q5_k"])
def test_matvec9_type_mixes_at_one_site(emu_lib, mirror, monkeypatch, tmp_path, case):
    """Weight-type mixes one kernel launch cannot take together are issued as one launch per group (engine.cc:launch_matvec; reference
    files mix freely, llama.cpp:4785-4850): a Q4_K_S file (attn_v and ffn_down in Q5_K beside Q4_K), a Q8_0 attn_v beside K-quant
    q / k, a Q4_0 attn_k and a Q5_K attn_v beside a Q4_K attn_q.  Token by token and through the prompt-chunk kernels, against the
    oracle restatement."""
    from tools import gguf as G
    p = str(tmp_path / "m.gguf")
    kw = dict(overrides=dict(n_layer=2))
    if case == "Q4_K_S":
        hp = synth.write_llama_gguf(p, "llama-tiny", "Q4_K_S", seed=11, **kw)
    elif case == "v_q8_0":
        hp = synth.write_llama_gguf(p, "llama-tiny", "Q4_K_M", seed=12, type_overrides={"attn_v.weight": G.Q8_0}, **kw)
    else:
        hp = synth.write_llama_gguf(p, "llama-tiny", "Q4_K_M", seed=13, type_overrides={"attn_k.weight": G.Q4_0, "attn_v.weight": G.Q5_K}, **kw)
    o = mirror.MirrorLlama(p, 32)
    toks = synth.prompt_tokens(9, hp["n_vocab"])
    o.eval(toks[:8], 0)
    want = np.array(o.eval(toks[8:], 8), copy=True)   # the restatement returns a view of its live logits buffer
    for pf in ("0", "1"):
        monkeypatch.setenv("CT_AMD_PF", pf)
        m = LLM(p, config=Config(context_length=32, batch_size=8, threads=1), lib=emu_lib)
        m.eval(toks)
        assert np.array_equal(m.logits.to_numpy(), want), (case, pf)
        t = int(want.argmax())
        m.eval([t])
        if pf == "0":
            nxt = np.array(o.eval([t], 9), copy=True)
        assert np.array_equal(m.logits.to_numpy(), nxt), (case, pf)


@pytest.mark.parametrize("arch,ftype", [("falcon", "Q8_0"), ("falcon", "Q4_0"), ("llama", "Q8_0"), ("llama", "Q4_0")])
def test_block32_rows_not_a_multiple_of_128(emu_lib, mirror, tmp_path, arch, ftype):
    """Q8_0 / Q4_0 rows that are whole 32-blocks but not whole groups of four (real Falcon-7B: n_embd 4544 = 142 blocks; here 192 =
    6 blocks, llama ffn_down 480 = 15 blocks): the decode arena's last record and the chunk kernels' last group of four blocks are
    padded with zero blocks, the activation images likewise — prompt through the chunk kernels, decode steps, against the oracle
    restatement, an odd head count (3) with one KV head included.  Reference: every type has a vec_dot and no row-length rule
    beyond the block size (ggml.c:1676-1795)."""
    p = str(tmp_path / "m.gguf")
    if arch == "falcon":
        hp = synth.write_falcon_gguf(p, "falcon-tiny7", ftype, seed=17, overrides=dict(n_embd=192, n_head=3, n_head_kv=1, n_ff=768, n_layer=2))
        o = mirror.MirrorFalcon(p, 32)
    else:
        from tools import gguf as G
        hp = synth.write_llama_gguf(p, "llama-tiny", ftype, seed=18, overrides=dict(n_embd=192, n_head=3, n_head_kv=1, n_ff=480, n_layer=2),
                                    type_overrides={"output.weight": G.Q8_0})   # llama.cpp:4787: rows that are not whole 256-blocks -> Q8_0 head
        o = mirror.MirrorLlama(p, 32)
    m = LLM(p, config=Config(context_length=32, batch_size=8, threads=1), lib=emu_lib)
    toks = synth.prompt_tokens(5, hp["n_vocab"])
    m.eval(toks)
    lg = np.array(o.eval(toks, 0), copy=True)
    assert np.array_equal(m.logits.to_numpy(), lg)
    assert np.array_equal(m.embeddings.to_numpy(), o.embeddings)
    f = m._lib.ctamd_chunk_tokens
    f.restype, f.argtypes = ctypes.c_longlong, [ctypes.c_void_p]
    assert int(f(m._llm)) == 5   # the prompt went through the chunk kernels
    t = int(lg.argmax())
    m.eval([t])
    assert np.array_equal(m.logits.to_numpy(), o.eval([t], 5))


@pytest.mark.parametrize("heads,n_embd,cus", [((4, 2), 256, 4), ((4, 2), 512, 16)])
def test_decode_attention_long_context_form(emu_lib, mirror, monkeypatch, tmp_path, heads, n_embd, cus):
    """Contexts above 1024 take the deep-ring form of the decode attention (kernels_attn9.h: four K-row slots, sixteen V chunks, every
    request unconditional, peeled last rounds, four score waves): head sizes 64 and 128, four / two V*P waves per workgroup (the
    emulated chip's CU count decides, CT_EMU_CUS), decode steps at positions 70.. (two 32-position fma steps + leftovers) against the
    oracle restatement."""
    monkeypatch.setenv("CT_EMU_CUS", str(cus))
    p = str(tmp_path / "m.gguf")
    hp = synth.write_llama_gguf(p, "llama-tiny", "Q4_K_M", seed=29, overrides=dict(n_embd=n_embd, n_head=heads[0], n_head_kv=heads[1], n_layer=1))
    m = LLM(p, config=Config(context_length=1088, batch_size=128, threads=1), lib=emu_lib)
    o = mirror.MirrorLlama(p, 1088)
    toks = synth.prompt_tokens(70, hp["n_vocab"])
    m.eval(toks)
    lg = np.array(o.eval(toks, 0), copy=True)
    assert np.array_equal(m.logits.to_numpy(), lg)
    for i in range(2):
        t = int(lg.argmax())
        m.eval([t])
        lg = np.array(o.eval([t], 70 + i), copy=True)
        assert np.array_equal(m.logits.to_numpy(), lg), "position %d" % (70 + i)


def test_context_above_32768(emu_lib, mirror, tmp_path):
    """The reference takes any context_length (models/llms/llama.cc:90-92).  Above 32768 positions a token's probability row no longer
    fits LDS: it lives in global memory (kernels_exact.h GPROB, one workgroup per head), prompts run token by token — against the
    oracle restatement at context 40000 (positions 0..36: the fma steps and the leftovers of the value dot product)."""
    p = str(tmp_path / "m.gguf")
    hp = synth.write_llama_gguf(p, "llama-tiny", "Q4_K_M", seed=37, overrides=dict(n_layer=1))
    m = LLM(p, config=Config(context_length=40000, batch_size=64, threads=1), lib=emu_lib)
    assert m.context_length == 40000
    o = mirror.MirrorLlama(p, 40000)
    toks = synth.prompt_tokens(35, hp["n_vocab"])
    m.eval(toks)
    lg = np.array(o.eval(toks, 0), copy=True)
    assert np.array_equal(m.logits.to_numpy(), lg)
    for i in range(2):
        t = int(lg.argmax())
        m.eval([t])
        lg = np.array(o.eval([t], 35 + i), copy=True)
        assert np.array_equal(m.logits.to_numpy(), lg), "position %d" % (35 + i)


@pytest.mark.parametrize("ftype", ["F16", "F32"])
def test_f16_weight_matrices(emu_lib, mirror, tmp_path, ftype):
    """A llama GGUF of ftype F32 (vec_dot_type F32: ggml_vec_dot_f32 of the weight row and the f32 activation row, ggml.c:2355-2389) or F16 (every 2-D tensor fp16; reference: vec_dot_type F16 — the activation row through ggml_fp32_to_fp16_row,
    ggml_vec_dot_f16 per output row, ggml.c:11031-11245 / :2392-2425): kernels_f16.h, token steps for the prompt too — against the oracle
    restatement (which the GPU suite and tests/test_oracle.py compare with the reference build on such a file)."""
    p = str(tmp_path / "m.gguf")
    hp = synth.write_llama_gguf(p, "llama-tiny", ftype, seed=41)
    m = LLM(p, config=Config(context_length=64, batch_size=8, threads=1), lib=emu_lib)
    o = mirror.MirrorLlama(p, 64)
    toks = synth.prompt_tokens(9, hp["n_vocab"])
    o.eval(toks[:8], 0)
    lg = np.array(o.eval(toks[8:], 8), copy=True)   # the reference's batches of 8: 8 + 1
    m.eval(toks)
    assert np.array_equal(m.logits.to_numpy(), lg)
    assert np.array_equal(m.embeddings.to_numpy(), o.embeddings)
    for i in range(2):
        t = int(lg.argmax())
        m.eval([t])
        lg = np.array(o.eval([t], 9 + i), copy=True)
        assert np.array_equal(m.logits.to_numpy(), lg), "position %d" % (9 + i)


@pytest.mark.parametrize("ftype", ["Q4_1", "Q5_0", "Q5_1"])
def test_q4_1_q5_0_q5_1_weight_matrices(emu_lib, mirror, tmp_path, ftype):
    """A llama GGUF of ftype Q4_1 / Q5_0 / Q5_1 (every 2-D tensor of the base type, output.weight Q6_K, token_embd rows dequantized by
    get_rows; reference: vec_dot_type Q8_1 / Q8_0 / Q8_1, ggml.c:2699 / :2825 / :3065 AVX2 forms): kernels_raw32.h on the file layout,
    token steps for the prompt too — against the oracle restatement (tests/test_oracle.py compares it with the reference build on the
    same kind of file and op by op)."""
    p = str(tmp_path / "m.gguf")
    hp = synth.write_llama_gguf(p, "llama-tiny", ftype, seed=43, overrides=dict(n_ff=608))   # ffn_down rows of 19 blocks: not whole rounds of four
    m = LLM(p, config=Config(context_length=64, batch_size=8, threads=1), lib=emu_lib)
    o = mirror.MirrorLlama(p, 64)
    toks = synth.prompt_tokens(9, hp["n_vocab"])
    o.eval(toks[:8], 0)
    lg = np.array(o.eval(toks[8:], 8), copy=True)   # the reference's batches of 8: 8 + 1
    m.eval(toks)
    assert np.array_equal(m.logits.to_numpy(), lg)
    assert np.array_equal(m.embeddings.to_numpy(), o.embeddings)
    for i in range(2):
        t = int(lg.argmax())
        m.eval([t])
        lg = np.array(o.eval([t], 9 + i), copy=True)
        assert np.array_equal(m.logits.to_numpy(), lg), "position %d" % (9 + i)


@pytest.mark.parametrize("arch", ["gpt2", "mpt"])
def test_legacy_graphs_at_widths_that_are_not_multiples_of_128(emu_lib, mirror, tmp_path, arch):
    """GPT-2 XL has n_embd 1600 (50 blocks of 32 per row); here 192 with three heads, for the gpt2 and the mpt graph: prompt through the
    chunk kernels (zero blocks behind a row's end), decode steps (the F32 attention's dot tails of 4 and more leftovers: dot_f32_tail)
    — against the oracle restatement."""
    p = str(tmp_path / "m.bin")
    if arch == "gpt2":
        synth.write_gpt2_ggml(p, dict(n_vocab=512, n_ctx=96, n_embd=192, n_head=3, n_layer=2), seed=5)
        o = mirror.MirrorGpt2(p)
    else:
        synth.write_mpt_ggml(p, dict(n_vocab=512, max_seq_len=96, n_embd=192, n_head=3, n_layer=2, alibi_bias_max=8.0, clip_qkv=0.0), seed=6, ftype=7)
        o = mirror.MirrorMpt(p, 96)
    m = LLM(p, arch, config=Config(context_length=96, batch_size=64, threads=1), lib=emu_lib)
    toks = synth.prompt_tokens(13, 512)
    m.eval(toks)
    lg = np.array(o.eval(toks, 0), copy=True)
    assert np.array_equal(m.logits.to_numpy(), lg)
    for i in range(2):
        t = int(lg.argmax())
        m.eval([t])
        lg = np.array(o.eval([t], 13 + i), copy=True)
        assert np.array_equal(m.logits.to_numpy(), lg)


@pytest.mark.parametrize("arch,ftype", [("gpt2", "Q5_1"), ("gpt2", "Q5_0"), ("mpt", "Q4_1"), ("mpt", "Q5_0"), ("falcon", "Q5_1"), ("falcon", "Q4_1"),
                                        ("gpt2", "F16"), ("mpt", "F16"), ("falcon", "F16"), ("gpt2", "F32"), ("falcon", "F32")])
def test_q4_1_q5_0_q5_1_in_the_layernorm_graphs(emu_lib, mirror, tmp_path, arch, ftype):
    """Legacy GGML files of ftype 3 / 8 / 9 (gpt2, mpt) and falcon GGUF files of ftype Q4_1 / Q5_0 / Q5_1: kernels_raw32.h — and of ftype 1 / F16
    (what the reference's convert scripts write): kernels_f16.h — behind a LayerNorm
    (with and without bias), the row-bias / GELU / two-residual epilogues, the tied Q5_x lm_head of gpt2 — against the oracle restatement
    (tests/test_oracle.py compares it with the reference build on the same kinds of file)."""
    ft = {"F32": 0, "F16": 1, "Q4_1": 3, "Q5_0": 8, "Q5_1": 9}[ftype]
    p = str(tmp_path / ("m.gguf" if arch == "falcon" else "m.bin"))
    if arch == "gpt2":
        synth.write_gpt2_ggml(p, dict(n_vocab=512, n_ctx=96, n_embd=192, n_head=3, n_layer=2), seed=5, ftype=ft)
        o = mirror.MirrorGpt2(p)
    elif arch == "mpt":
        synth.write_mpt_ggml(p, dict(n_vocab=512, max_seq_len=96, n_embd=192, n_head=3, n_layer=2, alibi_bias_max=8.0, clip_qkv=0.0), seed=6, ftype=ft)
        o = mirror.MirrorMpt(p, 96)
    else:
        synth.write_falcon_gguf(p, "falcon-tiny7", ftype, seed=17)
        o = mirror.MirrorFalcon(p, 96)
    m = LLM(p, None if arch == "falcon" else arch, config=Config(context_length=96, batch_size=64, threads=1), lib=emu_lib)
    toks = synth.prompt_tokens(7, 512)
    m.eval(toks)
    lg = np.array(o.eval(toks, 0), copy=True)
    assert np.array_equal(m.logits.to_numpy(), lg)
    for i in range(2):
        t = int(lg.argmax())
        m.eval([t])
        lg = np.array(o.eval([t], 7 + i), copy=True)
        assert np.array_equal(m.logits.to_numpy(), lg)


@pytest.mark.parametrize("head_type", ["F16", "Q5_0"])
def test_file_layout_lm_head_behind_prompt_chunks(emu_lib, mirror, tmp_path, head_type):
    """A K-quant file whose output.weight stays in a file-layout type (the F16 fallback of the reference's quantizer for rows that are not whole
    256-blocks, llama.cpp:4866-4869; a requantized head): the layers keep their prompt-chunk kernels, the chunk's last token goes through the
    file-layout mat-vec for the logits."""
    from tools import gguf as G
    p = str(tmp_path / "m.gguf")
    hp = synth.write_llama_gguf(p, "llama-tiny", "Q4_K_M", seed=47, type_overrides={"output.weight": {"F16": G.F16, "Q5_0": G.Q5_0}[head_type]})
    m = LLM(p, config=Config(context_length=64, batch_size=8, threads=1), lib=emu_lib)
    o = mirror.MirrorLlama(p, 64)
    toks = synth.prompt_tokens(13, hp["n_vocab"])
    o.eval(toks[:8], 0)
    lg = np.array(o.eval(toks[8:], 8), copy=True)
    m.eval(toks)
    assert np.array_equal(m.logits.to_numpy(), lg)
    f = m._lib.ctamd_chunk_tokens
    f.restype, f.argtypes = ctypes.c_longlong, [ctypes.c_void_p]
    assert int(f(m._llm)) == len(toks)   # the layers really took the chunk kernels (only the head is in a file-layout type)
    t = int(lg.argmax())
    m.eval([t])
    assert np.array_equal(m.logits.to_numpy(), o.eval([t], 13))

