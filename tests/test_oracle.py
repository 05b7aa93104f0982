"""Pins the oracle: the C restatement (oracle/mirror.c) must be BIT-IDENTICAL to the committed golden vectors, which
were produced by the real reference build (tests/golden/make_golden.py); where the reference build itself is present
it is checked against the same vectors (guards against a drifting toolchain)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from tools import gguf as G, synth

OPS = np.load(os.path.join(GOLDEN, "ops.npz"))
TYPES = [G.Q4_K, G.Q5_K, G.Q6_K, G.Q8_0, G.Q4_0]


def test_activation_quantizers_bit_exact(mirror):
    x = OPS["act_x"]
    assert np.array_equal(mirror.quantize_q8_K(x), OPS["act_q8_K"])  # includes a sign tie and an all-zero block
    assert np.array_equal(mirror.quantize_q8_0(x), OPS["act_q8_0"])


@pytest.mark.parametrize("t", TYPES)
def test_weight_dot_products_bit_exact(mirror, t):
    name = G.TYPE_NAMES[t]
    y = mirror.matvec(t, OPS["w_" + name], OPS["act_x"], OPS["act_x"].size)
    assert np.array_equal(y, OPS["y_" + name])


@pytest.mark.parametrize("t", TYPES)
def test_dequantize_bit_exact(mirror, t):
    name = G.TYPE_NAMES[t]
    K = OPS["act_x"].size
    assert np.array_equal(mirror.dequantize(OPS["w_" + name][:2], t, K), OPS["deq_" + name])
    assert np.array_equal(synth.dequantize(OPS["w_" + name][:2], t, K), OPS["deq_" + name])  # numpy restatement


def test_rope_and_rmsnorm_bit_exact(mirror):
    for i, p in enumerate(OPS["rope_pos"]):
        assert np.array_equal(mirror.rope(OPS["rope_x"][i], int(p)), OPS["rope_y"][i])
    assert np.array_equal(mirror.rms_norm_mul(OPS["norm_x"], OPS["norm_w"], 1e-5), OPS["norm_y"])


def test_falcon_ops_bit_exact(mirror):
    """LayerNorm*w+b, neox RoPE (the contraction form the reference build uses) and GELU through the fp16 table."""
    f = np.load(os.path.join(GOLDEN, "falcon_ops.npz"))
    assert np.array_equal(mirror.norm_mul_add(f["ln_x"], f["ln_w"], f["ln_b"], 1e-5), f["ln_y"])
    for i, p in enumerate(f["neox_pos"]):
        assert np.array_equal(mirror.rope_neox(f["neox_x"][i], int(p)), f["neox_y"][i])
    assert np.array_equal(mirror.gelu(f["gelu_x"]), f["gelu_y"])


@pytest.mark.parametrize("name", ["tiny-q4km", "tiny-q5km", "tiny-q80", "tiny-q40", "falcon-tiny-q4km", "falcon-tiny7-q4km",
                                  "gpt2-tiny-q40"])
def test_whole_model_bit_exact(mirror, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    if name.startswith("gpt2"):
        m = mirror.MirrorGpt2(os.path.join(GOLDEN, name + ".bin"))
    else:
        cls = mirror.MirrorFalcon if name.startswith("falcon") else mirror.MirrorLlama
        m = cls(os.path.join(GOLDEN, name + ".gguf"), 96)
    logits = m.eval(g["prompt"], 0)
    assert np.array_equal(logits, g["logits"][0])
    if not name.startswith("gpt2"):
        assert np.array_equal(m.embeddings, g["embeddings"][0])
    pos = len(g["prompt"])
    for i, t in enumerate(g["greedy"][:24]):
        assert int(np.argmax(logits)) == int(t)
        logits = m.eval([int(t)], pos + i)
        assert np.array_equal(logits, g["logits"][i + 1]), "step %d" % i


@pytest.mark.parametrize("name", ["tiny-q4km", "tiny-q5km"])
def test_batch_structure_bit_exact(mirror, name):
    """45-token prompt as one batch and in chunks of 8: the fma/leftover split of vec_dot_f16 follows the batch."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    lp = g["long_prompt"]
    m = mirror.MirrorLlama(os.path.join(GOLDEN, name + ".gguf"), 96)
    assert np.array_equal(m.eval(lp, 0), g["long_one"])
    m = mirror.MirrorLlama(os.path.join(GOLDEN, name + ".gguf"), 96)
    for s in range(0, len(lp), 8):
        out = m.eval(lp[s:s + 8], s)
    assert np.array_equal(out, g["long_chunked"])


def test_reference_build_reproduces_golden(ref):
    """The oracle/_ref binary in this checkout still produces the committed vectors."""
    q, _ = ref.quantize_activation(OPS["act_x"], G.Q4_K)
    assert np.array_equal(q, OPS["act_q8_K"])
    for t in TYPES:
        name = G.TYPE_NAMES[t]
        assert np.array_equal(ref.matvec(t, OPS["w_" + name], OPS["act_x"], OPS["act_x"].size), OPS["y_" + name])
    ops = ref.GgmlOps()
    assert np.array_equal(ops.rope(OPS["rope_x"][1][None], int(OPS["rope_pos"][1]))[0], OPS["rope_y"][1])
    assert np.array_equal(ops.scale_softmax(OPS["sm_x"], float(OPS["sm_scale"])), OPS["sm_y"])
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    r = ref.open_llm(os.path.join(GOLDEN, "tiny-q4km.gguf"), context_length=96, batch_size=8, threads=2)
    r.eval(list(g["prompt"]))
    assert np.array_equal(r.logits.to_numpy(), g["logits"][0])


def test_restatement_follows_reference_build_on_shapes_without_goldens(mirror, ref, tmp_path):
    """Shapes and types added after the golden vectors were generated, pinned on the reference build itself (needs oracle/_ref): fp16
    weight matrices (ftype F16: vec_dot_type F16), ftypes Q4_1 / Q5_0 / Q5_1, Q8_0 rows that are not whole groups of four blocks with an odd head count on one KV
    head (Falcon-7B's geometry), MPT heads of 112 (the f16 dot's scalar tail; the double-sqrt scale of the legacy graphs is pinned
    by tests/test_mpt.py)."""
    cases = []
    p = str(tmp_path / "f16.gguf")
    hp = synth.write_llama_gguf(p, "llama-tiny", "F16", seed=41)
    cases.append((p, None, mirror.MirrorLlama(p, 64), hp))
    for ft in ("Q4_1", "Q5_0", "Q5_1"):   # vec_dot_type Q8_1 / Q8_0 / Q8_1; token_embd rows through dequantize_row_q4_1 / q5_0 / q5_1
        p = str(tmp_path / (ft + ".gguf"))
        hp = synth.write_llama_gguf(p, "llama-tiny", ft, seed=43, overrides=dict(n_ff=608))
        cases.append((p, None, mirror.MirrorLlama(p, 64), hp))
    p = str(tmp_path / "f7.gguf")
    hp = synth.write_falcon_gguf(p, "falcon-tiny7", "Q8_0", seed=17, overrides=dict(n_embd=192, n_head=3, n_head_kv=1, n_ff=768, n_layer=2))
    cases.append((p, None, mirror.MirrorFalcon(p, 64), hp))
    p = str(tmp_path / "m112.bin")
    hp = synth.write_mpt_ggml(p, "mpt-tiny112", seed=23, ftype=2)
    cases.append((p, "mpt", mirror.MirrorMpt(p, 64), hp))
    # the LayerNorm graphs on Q4_1 / Q5_0 / Q5_1: legacy ftypes 3 / 8 / 9 (enum ggml_ftype), falcon GGUF
    p = str(tmp_path / "g51.bin")
    hp = synth.write_gpt2_ggml(p, dict(n_vocab=512, n_ctx=64, n_embd=192, n_head=3, n_layer=2), seed=5, ftype=9)
    cases.append((p, "gpt2", mirror.MirrorGpt2(p), hp))
    p = str(tmp_path / "m41.bin")
    hp = synth.write_mpt_ggml(p, dict(n_vocab=512, max_seq_len=64, n_embd=192, n_head=3, n_layer=2, alibi_bias_max=8.0, clip_qkv=0.0), seed=6, ftype=3)
    cases.append((p, "mpt", mirror.MirrorMpt(p, 64), hp))
    p = str(tmp_path / "f50.gguf")
    hp = synth.write_falcon_gguf(p, "falcon-tiny7", "Q5_0", seed=17)
    cases.append((p, None, mirror.MirrorFalcon(p, 64), hp))
    # fp16 matrices behind LayerNorms: legacy ftype 1 (what the reference's convert scripts write), falcon GGUF of ftype F16
    p = str(tmp_path / "g16.bin")
    hp = synth.write_gpt2_ggml(p, dict(n_vocab=512, n_ctx=64, n_embd=192, n_head=3, n_layer=2), seed=5, ftype=1)
    cases.append((p, "gpt2", mirror.MirrorGpt2(p), hp))
    p = str(tmp_path / "m16.bin")
    hp = synth.write_mpt_ggml(p, dict(n_vocab=512, max_seq_len=64, n_embd=192, n_head=3, n_layer=2, alibi_bias_max=8.0, clip_qkv=0.0), seed=6, ftype=1)
    cases.append((p, "mpt", mirror.MirrorMpt(p, 64), hp))
    p = str(tmp_path / "f16f.gguf")
    hp = synth.write_falcon_gguf(p, "falcon-tiny7", "F16", seed=17)
    cases.append((p, None, mirror.MirrorFalcon(p, 64), hp))
    # F32 matrices (vec_dot_type F32: ggml_vec_dot_f32 on the f32 activation row)
    p = str(tmp_path / "l32.gguf")
    hp = synth.write_llama_gguf(p, "llama-tiny", "F32", seed=41)
    cases.append((p, None, mirror.MirrorLlama(p, 64), hp))
    p = str(tmp_path / "g32.bin")
    hp = synth.write_gpt2_ggml(p, dict(n_vocab=512, n_ctx=64, n_embd=192, n_head=3, n_layer=2), seed=5, ftype=0)
    cases.append((p, "gpt2", mirror.MirrorGpt2(p), hp))
    for path, mt, o, hp in cases:
        r = ref.open_llm(path, model_type=mt, context_length=64, batch_size=64, threads=2)
        toks = synth.prompt_tokens(37, hp["n_vocab"])
        r.eval(toks)
        a = np.array(r.logits.to_numpy(), copy=True)
        assert np.array_equal(a, o.eval(toks, 0)), path
        t = int(a.argmax())
        r.eval([t])
        assert np.array_equal(r.logits.to_numpy(), o.eval([t], 37)), path


def test_f32_dot_tail_follows_the_reference_build(mirror, ref):
    """ggml_vec_dot_f32 (GPT-2 / StarCoder attention over the F32 KV cache): the reference BUILD runs the scalar tail `sumf += x[i]*y[i]`
    as groups of 8 and 4 UNFUSED products added in order and at most 3 fused steps (gcc's in-order vectorised reduction).  Every length
    1..70 against the exported function of oracle/_ref (a pure-fma tail differs in ~40 % of the dots with 4 or more leftovers)."""
    import ctypes
    from oracle import mirror as mm
    L = mm.lib()
    L.mir_vec_dot_f32.restype = ctypes.c_float
    L.mir_vec_dot_f32.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    vd = ref._VEC_DOT(ref.traits(G.F32).vec_dot)
    rng = np.random.default_rng(5)
    for n in range(1, 71):
        for _ in range(20):
            x = rng.standard_normal(n).astype(np.float32)
            y = rng.random(n).astype(np.float32)
            s = ctypes.c_float(0)
            vd(n, ctypes.byref(s), x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p))
            assert np.float32(s.value) == np.float32(L.mir_vec_dot_f32(n, x.ctypes.data, y.ctypes.data)), n


def test_f16_dot_follows_the_reference_build_at_every_length(mirror, ref):
    """ggml_vec_dot_f16 (llama / falcon / mpt attention; F16 weight matrices): 32-element steps, the AVX reduce tree, and the scalar tail
    `sumf += (double)(x[i]*y[i])` — every length 1..70 against the exported function of oracle/_ref."""
    import ctypes
    from oracle import mirror as mm
    L = mm.lib()
    L.mir_vec_dot_f16.restype = ctypes.c_float
    L.mir_vec_dot_f16.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    vd = ref._VEC_DOT(ref.traits(G.F16).vec_dot)
    rng = np.random.default_rng(6)
    for n in range(1, 71):
        for _ in range(20):
            x = (rng.standard_normal(n) * rng.choice([0.05, 1.0, 30.0])).astype(np.float16)
            y = rng.random(n).astype(np.float16)
            s = ctypes.c_float(0)
            vd(n, ctypes.byref(s), x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p))
            assert np.float32(s.value) == np.float32(L.mir_vec_dot_f16(n, x.ctypes.data, y.ctypes.data)), n


@pytest.mark.parametrize("t", [G.Q4_1, G.Q5_0, G.Q5_1])
def test_q4_1_q5_0_q5_1_follow_the_reference_build(mirror, ref, t):
    """Q8_1 activation blocks (d f32, s = d * sum), the three AVX2 dot products — `summs += m * s` is ONE fused multiply-add per block in
    the reference build (an unfused sum differs in two rows of three) — and the dequantize rows (numpy and C restatement), op by op
    against oracle/_ref: scales 1e-3 .. 300, all-zero blocks, rows of 1 .. 142 blocks."""
    import ctypes
    from oracle import mirror as mm
    L = mm.lib()
    rng = np.random.default_rng(3 + t)
    for trial in range(40):
        sc = float(rng.choice([1e-3, 0.1, 1, 10, 300]))
        K = int(rng.choice([32, 96, 512, 2816, 4544]))
        w = (rng.standard_normal((5, K)) * 0.1 + rng.choice([0, 0.05])).astype(np.float32)
        wraw = synth.quantize(w, t).reshape(5, -1)
        x = (rng.standard_normal(K) * sc).astype(np.float32)
        if trial % 7 == 0:
            x[:32] = 0
        a, _ = ref.quantize_activation(x, t)
        if t != G.Q5_0:
            out = np.zeros(K // 32 * 40, dtype=np.uint8)
            L.mir_quantize_row_q8_1(x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), K)
            assert np.array_equal(out, np.asarray(a).view(np.uint8).ravel())
        assert np.array_equal(mirror.matvec(t, wraw, x, K), ref.matvec(t, wraw, x, K))
        d = ref.dequantize(wraw, t, K)
        assert np.array_equal(mirror.dequantize(wraw, t, K), d)
        assert np.array_equal(synth.dequantize(wraw, t, K), d)
        assert np.abs(d - w).max() < 0.05   # the numpy quantizers (synthetic files) produce legal, close blocks
