"""GPU parity tests proper (run with -m gpu on the MI355X box).  Everything goes through the C ABI of the HIP library
(ctransformers_amd/lib/libctransformers.so via the Python host mirror).  Bar: BIT-IDENTICAL logits to the reference CPU
build — checked (a) against the committed golden vectors and (b) against the reference .so itself (oracle/_ref, which
travels with the snapshot) on freshly generated models, plus size-independent properties at the full 7B size."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from tools import synth
from ctransformers_amd.llm import LLM, Config

pytestmark = pytest.mark.gpu


def open_hip(path, model_type=None, **kw):
    cfg = dict(context_length=96, batch_size=8)
    cfg.update(kw)
    return LLM(path, model_type, config=Config(**cfg))  # default lib = the HIP build; raises if missing / no GPU


def chunk_tokens(m):
    """Tokens the handle evaluated through the prompt-chunk kernels (include/ctransformers_amd_ext.h)."""
    import ctypes
    f = m._lib.ctamd_chunk_tokens
    f.restype, f.argtypes = ctypes.c_longlong, [ctypes.c_void_p]
    return int(f(m._llm))


@pytest.mark.parametrize("name", ["tiny-q4km", "tiny-q4km-refq", "tiny-q5km", "tiny-q80", "tiny-q40", "falcon-tiny-q4km", "falcon-tiny7-q4km",
                                  "gpt2-tiny-q40"])
@pytest.mark.parametrize("graph", ["1", "0"])
def test_golden_logits_bit_identical(name, graph, monkeypatch):
    monkeypatch.setenv("CT_AMD_GRAPH", graph)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    if name.startswith("gpt2"):
        m = open_hip(os.path.join(GOLDEN, name + ".bin"), "gpt2")
    else:
        m = open_hip(os.path.join(GOLDEN, name + ".gguf"))
    m.eval(list(g["prompt"]))
    assert np.array_equal(m.logits.to_numpy(), g["logits"][0])
    # every golden model: the prompt went through the chunk kernels (one pass; batches of 8 + 3 for the arithmetic)
    assert chunk_tokens(m) == len(g["prompt"])
    if not name.startswith("gpt2"):
        assert np.array_equal(m.embeddings.to_numpy(), g["embeddings"][0])
    for i, t in enumerate(g["greedy"]):
        assert m.sample(top_k=1, repetition_penalty=1.0) == int(t)
        m.eval([int(t)])
        assert np.array_equal(m.logits.to_numpy(), g["logits"][i + 1]), "step %d" % i


@pytest.mark.parametrize("name", ["tiny-q4km", "tiny-q5km"])
def test_batch_structure_on_gpu(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = open_hip(os.path.join(GOLDEN, name + ".gguf"), batch_size=64)
    m.eval(list(g["long_prompt"]))
    assert np.array_equal(m.logits.to_numpy(), g["long_one"])
    m = open_hip(os.path.join(GOLDEN, name + ".gguf"), batch_size=8)
    m.eval(list(g["long_prompt"]))
    assert np.array_equal(m.logits.to_numpy(), g["long_chunked"])


def test_abi_semantics_on_gpu():
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    m = open_hip(os.path.join(GOLDEN, "tiny-q4km.gguf"))
    prompt = list(g["prompt"])
    m.eval(prompt, batch_size=4)  # chunked == one shot
    assert np.array_equal(m.logits.to_numpy(), g["logits"][0])
    m.logits[7] = -55.25  # in-place mutation persists (reference tests/test_model.py:10-16)
    assert m.logits[7] == -55.25
    m._context = m._context[:-3]  # KV overwrite at an earlier n_past
    m.eval(prompt[-3:])
    assert np.array_equal(m.logits.to_numpy(), g["logits"][0])
    # sampler / tokenizer parity of the HIP binary: tests/test_tokenizers.py::test_tokenizers_and_samplers_match_reference_hip_build


LEGACY_FTYPE = {"F32": 0, "F16": 1, "Q4_0": 2, "Q4_1": 3, "Q8_0": 7, "Q5_0": 8, "Q5_1": 9}   # enum ggml_ftype (reference ggml.h:322-336)


@pytest.mark.parametrize("shape,ftype,n_prompt,n_decode", [
    ("llama-small", "Q4_K_M", 20, 60),   # MHA 8/8, head_dim 64, K = 512 / 1280 (odd block counts)
    ("llama-tiny", "Q5_K_M", 5, 80),     # GQA 4/2, runs past 64 positions (fp16 dot leftovers + full 32-steps)
    ("llama-7b-2l", "Q4_K_M", 33, 12),   # two layers at the real 7B shapes incl. the 32000x4096 Q6_K head
    ("llama-small", "Q8_0", 20, 40),     # config 3: 32-element blocks, Q8_0 activations (kernels_q32.h)
    ("llama-small", "Q4_0", 20, 40),     # Q4_0 layers + Q6_K head
    ("llama-7b-2l", "Q8_0", 33, 8),      # real 7B shapes: K = 4096 / 11008 (86 block groups), 32000-row Q8_0 head
    ("llama-7b-2l", "Q4_K_S", 33, 8),    # Q4_K_S mix at 7B widths: attn_v / ffn_down in Q5_K beside Q4_K (one launch per type group at the QKV site)
    ("llama-7b-2l", "Q4_K_M+v_q8_0", 20, 6),   # a Q8_0 attn_v beside K-quant q / k: K-quant and 32-block kernels at one site
    ("llama-7b-2l", "F16", 12, 6),       # fp16 weight matrices at the 7B widths (kernels_f16.h: token steps, the prompt too)
    ("llama-7b-2l", "Q4_1", 12, 6),      # ftypes Q4_1 / Q5_0 / Q5_1 at the 7B widths (kernels_raw32.h on the file layout: Q8_1 / Q8_0 activation
    ("llama-7b-2l", "Q5_0", 12, 6),      # blocks, token steps for the prompt too; rows of 11008 = 344 blocks)
    ("llama-7b-2l", "Q5_1", 12, 6),
    ("falcon-small", "Q4_K_M", 40, 30),  # config 4 graph: LayerNorm x2, fused QKV (Q5_K), neox RoPE, GQA 16/2, GELU, Q8_0 head
    ("falcon-tiny7", "Q8_0", 20, 50),    # 7B-style block (one norm, MQA) on the 32-element-block kernels
    ("falcon-7b-2l", "Q8_0", 9, 6),      # real Falcon-7B widths: n_embd 4544 = 142 blocks of 32 (not whole groups of four), 71 heads on ONE KV head
    ("falcon-7b-2l", "Q4_0", 9, 4),
    ("falcon-40b-2l", "Q4_K_M", 9, 4),   # config 4 widths: K = 8192 (12288 instantiation) and K = 32768 (wide-K path), Q8_0 head
    ("llama-70b-2l", "Q5_K_M", 9, 4),    # config 5 widths: GQA 64/8, K = 8192 / 28672, Q5_K + Q6_K
    ("gpt2-117m", "Q4_0", 40, 24),       # config 1: GPT-2 117M shapes, legacy GGML container, F32 KV cache, tied lm_head
    ("gpt2-xl-2l", "Q4_0", 20, 24),       # GPT-2 XL widths: n_embd 1600 (50 blocks per row), 25 heads; the F32 dot tails of 4+ leftovers
    ("starcoder-1b-4l", "Q8_0", 40, 16),  # StarCoderBase-1B widths through the reference's starcoder loader: heads of 128, 49152 rows
    ("starcoder-7b-2l", "Q4_0", 33, 8),   # StarCoderBase-7B widths: rows of 16384 (wide-row kernels, bias epilogues)
    ("mpt-7b-2l", "Q4_0", 33, 12),        # MPT-7B widths: ALiBi over 32 heads of 128, 50432-row tied head, rows of 16384
    ("mpt-7b-2l", "Q8_0", 12, 6),
    ("gpt2-xl-2l", "Q5_1", 9, 6),         # legacy ftypes 3 / 8 / 9 and falcon GGUF in Q4_1 / Q5_0 / Q5_1: kernels_raw32.h behind LayerNorms, row-bias /
    ("starcoder-1b-4l", "Q4_1", 9, 4),    # GELU / two-residual epilogues, tied lm_head in the same type
    ("mpt-7b-2l", "Q5_0", 9, 4),
    ("falcon-7b-2l", "Q5_1", 9, 4),
    ("gpt2-xl-2l", "F16", 9, 4),          # fp16 matrices behind LayerNorms (legacy ftype 1: what the reference's convert scripts write; falcon GGUF F16)
    ("mpt-7b-2l", "F16", 9, 4),
    ("falcon-7b-2l", "F16", 9, 4),
    ("llama-7b-2l", "F32", 9, 4),         # F32 matrices (vec_dot_type F32): the fp16 kernels' structure on f32 operands
    ("gpt2-xl-2l", "F32", 9, 4),
    ("mpt-30b-2l", "Q4_0", 40, 8),        # MPT-30B widths: 64 heads of 112 (the f16 dot's scalar tail), d_model 7168, rows of 28672
])
def test_bit_identical_to_reference_build(ref, tmp_path, shape, ftype, n_prompt, n_decode):
    p = str(tmp_path / "m.gguf")
    mt = None
    if shape.startswith("gpt2"):
        hp, mt = synth.write_gpt2_ggml(p, shape, seed=21, ftype=LEGACY_FTYPE[ftype]), "gpt2"
    elif shape.startswith("starcoder"):
        hp, mt = synth.write_gpt2_ggml(p, shape, seed=21, ftype=LEGACY_FTYPE[ftype], pieces=synth.STARCODER_PIECES), "starcoder"
    elif shape.startswith("mpt"):
        hp, mt = synth.write_mpt_ggml(p, shape, seed=21, ftype=LEGACY_FTYPE[ftype]), "mpt"
    elif "+v_q8_0" in ftype:
        from tools import gguf as G
        hp = synth.write_llama_gguf(p, shape, ftype.split("+")[0], seed=21, type_overrides={"attn_v.weight": G.Q8_0})
    else:
        hp = (synth.write_falcon_gguf if shape.startswith("falcon") else synth.write_llama_gguf)(p, shape, ftype, seed=21)
    ctx = n_prompt + n_decode + 8
    r = ref.open_llm(p, model_type=mt, context_length=ctx, batch_size=64, threads=8)
    m = open_hip(p, mt, context_length=ctx, batch_size=64)
    toks = synth.prompt_tokens(n_prompt, hp["n_vocab"])
    r.eval(toks)
    m.eval(toks)
    # every model family / weight type evaluates prompts through the chunk kernels (32-block rows that are not whole groups of four
    # blocks — Falcon-7B's 4544 — with zero blocks behind the row's end) — except file-layout matrices (F32, F16, Q4_1, Q5_0, Q5_1): token by token
    assert chunk_tokens(m) == (0 if ftype in ("F32", "F16", "Q4_1", "Q5_0", "Q5_1") else n_prompt)
    for i in range(n_decode):
        a, b = r.logits.to_numpy(), m.logits.to_numpy()
        assert np.array_equal(a, b), "step %d: max rel %.3g" % (i, np.abs(a - b).max() / np.abs(a).max())
        t = int(a.argmax())
        r.eval([t])
        m.eval([t])
    assert len(r.embeddings) == len(m.embeddings)
    if len(r.embeddings):
        assert np.array_equal(r.embeddings.to_numpy(), m.embeddings.to_numpy())


def bench_file(config):
    """The file `bench.py --config N` times (same path, same generator, same seed): block pools out of the reference's own ggml_quantize_chunk where
    oracle/_ref travelled with the snapshot ("refq"), this repo's numpy quantizers otherwise ("r2") — the timed bytes are the checked bytes."""
    import bench
    shape, ftype, p = bench.CONFIGS[config]
    if not os.path.exists(p):
        (synth.write_falcon_gguf if shape.startswith("falcon") else synth.write_llama_gguf)(p + ".tmp", shape, ftype, seed=1234, quantizer=bench.QUANTIZER)
        os.replace(p + ".tmp", p)
    return shape, ftype, p


@pytest.fixture(scope="module")
def model_7b(tmp_path_factory):
    if os.environ.get("CTAMD_BENCH_MODEL") and os.path.exists(os.environ["CTAMD_BENCH_MODEL"]):
        return os.environ["CTAMD_BENCH_MODEL"]
    return bench_file(2)[2]


def test_full_7b_properties_and_reference(ref, model_7b):
    """BASELINE.json's full-size config: parity vs the reference CPU build on a short prompt + greedy steps, and
    size-independent properties (determinism across instances, chunking invariance, KV rollback)."""
    toks = synth.prompt_tokens(24, 32000)
    m = open_hip(model_7b, context_length=512, batch_size=128)
    m.eval(toks)
    first = m.logits.to_numpy().copy()
    r = ref.open_llm(model_7b, context_length=512, batch_size=128, threads=16)
    r.eval(toks)
    assert np.array_equal(r.logits.to_numpy(), first)
    seq = []
    for i in range(8):
        a, b = r.logits.to_numpy(), m.logits.to_numpy()
        assert np.array_equal(a, b), "7B step %d" % i
        t = int(a.argmax())
        seq.append(t)
        r.eval([t])
        m.eval([t])
    del r
    last = m.logits.to_numpy().copy()
    # rollback: drop the last 5 evaluated tokens and replay them -> same logits (KV overwrite semantics)
    replay = m._context[-5:]
    m._context = m._context[:-5]
    m.eval(replay, batch_size=2)
    assert np.array_equal(m.logits.to_numpy(), last)
    # a second, independent instance is bit-deterministic
    m2 = open_hip(model_7b, context_length=512, batch_size=8)
    m2.eval(toks)  # chunks of 8
    assert np.array_equal(m2.logits.to_numpy(), first)


def test_config2_full_size(ref, model_7b):
    """BASELINE.json configs[1] as written: Llama-2-7B Q4_K_M, 128-token prompt (batch_size 128, and again in reference batches of 8),
    then 256 greedy tokens — EVERY logits vector bit-identical to the reference CPU build on the same file."""
    toks = synth.prompt_tokens(128, 32000)
    m = open_hip(model_7b, context_length=512, batch_size=128)
    r = ref.open_llm(model_7b, context_length=512, batch_size=128, threads=16)
    m.eval(toks)
    r.eval(toks)
    assert chunk_tokens(m) == 128
    for i in range(256):
        a, b = r.logits.to_numpy(), m.logits.to_numpy()
        assert np.array_equal(a, b), "config 2, decode step %d: max rel %.3g" % (i, np.abs(a - b).max() / np.abs(a).max())
        t = int(a.argmax())
        r.eval([t])
        m.eval([t])
    assert np.array_equal(r.logits.to_numpy(), m.logits.to_numpy())
    assert np.array_equal(r.embeddings.to_numpy(), m.embeddings.to_numpy())
    # the same prompt in the reference's default batches of 8 (16 batches; coalesced into one pass here)
    m8 = open_hip(model_7b, context_length=512, batch_size=8)
    r8 = ref.open_llm(model_7b, context_length=512, batch_size=8, threads=16)
    m8.eval(toks)
    r8.eval(toks)
    assert np.array_equal(r8.logits.to_numpy(), m8.logits.to_numpy())


def test_chunk_path_repeatable_on_full_7b(ref, model_7b):
    """The LDS-DMA staging of kernels_pg.h once produced RARE wrong stage data on the 32-layer model that single runs of the parity
    tests did not catch: fresh handles, several prompt lengths (one group, ragged groups, a full chunk, two chunks),
    four repetitions, every logits vector equal to the reference CPU build's (tools/stress_chunks.py is the longer form)."""
    lens = (24, 33, 128, 200)
    r = ref.open_llm(model_7b, context_length=512, batch_size=128, threads=16)
    want = {}
    for n in lens:
        r._context = []
        r.eval(synth.prompt_tokens(n, 32000))
        want[n] = r.logits.to_numpy().copy()
    del r
    for rep in range(4):
        m = open_hip(model_7b, context_length=512, batch_size=128)
        for n in lens:
            m._context = []
            m.eval(synth.prompt_tokens(n, 32000))
            assert np.array_equal(m.logits.to_numpy(), want[n]), "repetition %d, %d-token prompt" % (rep, n)
        del m


def test_config3_full_size_q8_0(ref, tmp_path_factory):
    """BASELINE.json configs[2]: the full 32-layer Llama-2-7B Q8_0 file — 128-token prompt + 32 greedy tokens against the reference
    build (kernels_q32.h decode, the dot4 chunk kernels for the prompt)."""
    p = bench_file(3)[2]
    toks = synth.prompt_tokens(128, 32000)
    m = open_hip(p, context_length=256, batch_size=128)
    r = ref.open_llm(p, context_length=256, batch_size=128, threads=16)
    m.eval(toks)
    r.eval(toks)
    for i in range(32):
        a, b = r.logits.to_numpy(), m.logits.to_numpy()
        assert np.array_equal(a, b), "config 3, decode step %d" % i
        t = int(a.argmax())
        r.eval([t])
        m.eval([t])
    os.remove(p)


def test_full_7b_numpy_quantized_blocks(ref):
    """The one full-size case on this repo's own numpy quantizers (tools/synth.py) — the other full-size tests read the files bench.py times, whose blocks
    come out of the reference's ggml_quantize_chunk: an 8-token prompt + 4 greedy steps of the 32-layer Q4_K_M file against the reference build."""
    p = "/tmp/ctamd_llama2_7b_q4km_r2.gguf"
    if not os.path.exists(p):
        synth.write_llama_gguf(p + ".tmp", "llama-2-7b", "Q4_K_M", seed=1234, quantizer=None)
        os.replace(p + ".tmp", p)
    m = open_hip(p, context_length=64, batch_size=8)
    r = ref.open_llm(p, context_length=64, batch_size=8, threads=16)
    toks = synth.prompt_tokens(8, 32000)
    m.eval(toks)
    r.eval(toks)
    for i in range(4):
        a, b = r.logits.to_numpy(), m.logits.to_numpy()
        assert np.array_equal(a, b), "step %d" % i
        t = int(a.argmax())
        r.eval([t])
        m.eval([t])
    del m, r
    os.remove(p)


def _big_configs_on():
    """Configs 4 / 5 at full size need 25 GB / 49 GB of scratch disk (pooled synthesis: tens of seconds): on by default where /tmp has the room,
    CTAMD_BENCH_BIG=0 / 1 forces."""
    import shutil
    v = os.environ.get("CTAMD_BENCH_BIG")
    return v == "1" if v is not None else shutil.disk_usage("/tmp").free > 60e9


@pytest.mark.skipif(not _big_configs_on(), reason="needs 60 GB free in /tmp (or CTAMD_BENCH_BIG=1): synthesises the 25 GB / 49 GB files of BASELINE configs 4 / 5")
@pytest.mark.parametrize("config", [4, 5])
def test_big_config_full_size(ref, config):
    """BASELINE.json configs[3] / configs[4] at FULL size on one GPU: the 60-layer Falcon-40B Q4_K_M / 80-layer Llama-2-70B Q5_K_M file
    (the ones `bench.py --config 4|5` times), an 8-token prompt + 4 greedy steps, every logits vector bit-identical to the reference CPU
    build on the same file.  (The 2-layer models of test_reference_build_parity cover the same widths on every run.)"""
    shape, ftype, p = bench_file(config)
    m = open_hip(p, context_length=64, batch_size=8)
    r = ref.open_llm(p, context_length=64, batch_size=8, threads=32)
    toks = synth.prompt_tokens(8, m.vocab_size)
    m.eval(toks)
    r.eval(toks)
    for i in range(4):
        a, b = r.logits.to_numpy(), m.logits.to_numpy()
        assert np.array_equal(a, b), "config %d, step %d: max rel %.3g" % (config, i, np.abs(a - b).max() / np.abs(a).max())
        t = int(a.argmax())
        assert m.sample(top_k=1, repetition_penalty=1.0) == t
        r.eval([t])
        m.eval([t])
    assert np.array_equal(r.logits.to_numpy(), m.logits.to_numpy())
    del m, r
    if os.environ.get("CTAMD_BENCH_KEEP_BIG") != "1":
        os.remove(p)   # one at a time on the scratch disk


def test_pipeline_stages_on_gpu():
    """Row (e) on hardware: two layer stages of the HIP library chained through device-resident hand-off rows are
    bit-identical to the reference goldens (prefill in chunks of 8 + greedy decode)."""
    import torch
    from tools import rccl_pipeline as pipeline
    path = os.path.join(GOLDEN, "tiny-q4km.gguf")
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    s0 = pipeline.HipStage(path, 0, 1, context_length=96, device="cuda:0")
    s1 = pipeline.HipStage(path, 1, 2, context_length=96, device="cuda:0")
    prompt = [int(t) for t in g["long_prompt"]]
    for s in range(0, len(prompt), 8):
        c = prompt[s:s + 8]
        x = s0.forward(c, s)
        assert x.is_cuda and tuple(x.shape) == (len(c), s0.n_embd)
        logits = s1.forward([0] * len(c), s, x)
    torch.cuda.synchronize()
    assert np.array_equal(logits.numpy(), g["long_chunked"])
    s0b = pipeline.HipStage(path, 0, 1, context_length=96, device="cuda:0")
    s1b = pipeline.HipStage(path, 1, 2, context_length=96, device="cuda:0")
    p = [int(t) for t in g["prompt"]]
    logits = s1b.forward([0] * len(p), 0, s0b.forward(p, 0))
    assert np.array_equal(logits.numpy(), g["logits"][0])
    for i, t in enumerate(g["greedy"][:4]):
        logits = s1b.forward([0], len(p) + i, s0b.forward([int(t)], len(p) + i))
        assert np.array_equal(logits.numpy(), g["logits"][i + 1])


def _visible_gpus():
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


@pytest.mark.parametrize("shape,ftype", [("llama-7b-2l", "Q4_K_M"), ("llama-70b-2l", "Q5_K_M")])
def test_inprocess_pipeline_two_real_devices(ref, tmp_path, monkeypatch, shape, ftype):
    """Row (e) across a REAL device boundary (skipped on a 1-GPU box): two stages on devices 0 and 1 behind ctransformers_llm_create —
    peer access, per-device LDS opt-ins and CU counts, cross-device event waits, hipMemcpyPeerAsync over xGMI, per-stage chunk graphs —
    against the reference build: a 40-token prompt (micro-batches of 16) and greedy steps, every logits vector.  Both the
    gpu_layers-driven default partition (no CT_AMD_DEVICES) and the explicit one."""
    if _visible_gpus() < 2:
        pytest.skip("needs two visible MI355X devices")
    from tools import synth
    from ctransformers_amd.llm import LLM, Config
    p = str(tmp_path / "m.gguf")
    hp = synth.write_llama_gguf(p, shape, ftype, seed=21)
    cfg = dict(context_length=128, batch_size=64, threads=8)
    r = ref.open_llm(p, **cfg)
    toks = synth.prompt_tokens(40, hp["n_vocab"])
    r.eval(toks)
    want = [np.array(r.logits.to_numpy(), copy=True)]
    for _ in range(3):
        t = int(want[-1].argmax())
        r.eval([t])
        want.append(np.array(r.logits.to_numpy(), copy=True))
    monkeypatch.setenv("CT_AMD_PP_MB", "16")
    for env, gpu_layers in (("0,1", 1000), (None, 1)):
        if env is None:
            monkeypatch.delenv("CT_AMD_DEVICES", raising=False)
        else:
            monkeypatch.setenv("CT_AMD_DEVICES", env)
        m = LLM(p, config=Config(gpu_layers=gpu_layers, **cfg))
        import ctypes
        m._lib.ctamd_n_stages.restype, m._lib.ctamd_n_stages.argtypes = ctypes.c_int, [ctypes.c_void_p]
        assert m._lib.ctamd_n_stages(m._llm) == 2
        for rep in range(3):            # the third pass replays the stages' chunk graphs
            m._context = []
            m.eval(toks)
            assert np.array_equal(m.logits.to_numpy(), want[0]), (env, rep)
        for i in range(3):
            t = m.sample(top_k=1, repetition_penalty=1.0)
            assert t == int(want[i].argmax())
            m.eval([t])
            assert np.array_equal(m.logits.to_numpy(), want[i + 1])


def test_pipeline_two_processes_one_gpu(tmp_path):
    """The N > 1 driver on the 1-GPU box: two ranks (gloo rendezvous on 127.0.0.1, hand-off staged through the host), each
    with its HIP stage on cuda:0.  Same assertions as the CPU gloo test, against the reference goldens."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    from test_pipeline import _free_port
    path = os.path.join(GOLDEN, "tiny-q4km.gguf")
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "pipeline_worker.py"), path, "hip",
                                       str(tmp_path), "8", "3"], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    recs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert np.array_equal(np.load(tmp_path / "prefill_logits.npy"), g["long_chunked"])
    assert recs[0]["tokens"] == recs[1]["tokens"] and len(recs[0]["tokens"]) == 3


@pytest.mark.parametrize("ctx,overrides", [(3072, dict(n_embd=512, n_head=4, n_head_kv=2, n_layer=2, n_ff=1024)),   # heads of 128, GQA 4/2
                                            (3072, None)])                                                             # llama-small: heads of 64
def test_chunk_attention_8_tokens_per_workgroup(ref, tmp_path, ctx, overrides):
    """Contexts between 2048 and 4096: attn_chunk_long_kernel keeps 8 (not 16) probability rows in LDS.  A 300-token prompt in
    reference batches of 24 (ragged batch ends), then decode, against the reference build."""
    p = str(tmp_path / "m.gguf")
    hp = synth.write_llama_gguf(p, "llama-small", "Q4_K_M", seed=35, overrides=overrides)
    r = ref.open_llm(p, context_length=ctx, batch_size=24, threads=8)
    m = open_hip(p, context_length=ctx, batch_size=24)
    toks = synth.prompt_tokens(300, hp["n_vocab"])
    r.eval(toks)
    m.eval(toks)
    for i in range(6):
        a, b = r.logits.to_numpy(), m.logits.to_numpy()
        assert np.array_equal(a, b), "position %d" % (300 + i)
        t = int(a.argmax())
        r.eval([t])
        m.eval([t])


def test_long_context(ref, tmp_path):
    """A 630-token prompt (chunks of 64) and decode beyond it: several passes of the attention kernel's position loops,
    logits bit-identical to the reference build."""
    p = str(tmp_path / "m.gguf")
    hp = synth.write_llama_gguf(p, "llama-small", "Q4_K_M", seed=33)
    r = ref.open_llm(p, context_length=768, batch_size=64, threads=8)
    m = open_hip(p, context_length=768, batch_size=64)
    toks = synth.prompt_tokens(630, hp["n_vocab"])
    r.eval(toks)
    m.eval(toks)
    for i in range(24):
        a, b = r.logits.to_numpy(), m.logits.to_numpy()
        assert np.array_equal(a, b), "position %d" % (630 + i)
        t = int(a.argmax())
        r.eval([t])
        m.eval([t])


def test_context_above_8192(ref, tmp_path):
    """context_length 10240 (the probability row of the attention kernel lives in dynamic LDS, up to 32768 positions): a 9000-token
    prompt, then greedy steps past position 9000, bit-identical to the reference CPU build; then a handle at context 40000 (the row
    in global memory)."""
    p = str(tmp_path / "c.gguf")
    hp = synth.write_llama_gguf(p, "llama-tiny", "Q4_K_M", seed=43)
    toks = synth.prompt_tokens(9000, hp["n_vocab"])
    r = ref.open_llm(p, context_length=10240, batch_size=128, threads=16)
    m = open_hip(p, context_length=10240, batch_size=128)
    r.eval(toks)
    m.eval(toks)
    for i in range(8):
        a, b = r.logits.to_numpy(), m.logits.to_numpy()
        assert np.array_equal(a, b), "position %d" % (9000 + i)
        t = int(a.argmax())
        r.eval([t])
        m.eval([t])
    # above the 32768 positions the kernels' LDS row can hold: the probability row in global memory, prompts token by token
    r = ref.open_llm(p, context_length=40000, batch_size=64, threads=16)
    m = open_hip(p, context_length=40000, batch_size=64)
    short = synth.prompt_tokens(70, hp["n_vocab"])
    r.eval(short)
    m.eval(short)
    for i in range(3):
        a, b = r.logits.to_numpy(), m.logits.to_numpy()
        assert np.array_equal(a, b), "context 40000, position %d" % (70 + i)
        t = int(a.argmax())
        r.eval([t])
        m.eval([t])


def test_context_2048_gqa_64_8(ref, tmp_path):
    """a 2048-position context at the GQA shape of Llama-2-70B (64 query heads on 8 KV heads, head_dim 128; the
    2-layer model at the real widths): a 1900-token prompt in reference batches of 128 (chunks of 128 here, 15 attention passes of
    growing length, every query head of a group reading the same K/V rows), then 40 greedy steps up to position 1940, every
    logits vector bit-identical to the reference CPU build."""
    p = str(tmp_path / "g.gguf")
    hp = synth.write_llama_gguf(p, "llama-70b-2l", "Q5_K_M", seed=41)
    toks = synth.prompt_tokens(1900, hp["n_vocab"])
    r = ref.open_llm(p, context_length=2048, batch_size=128, threads=16)
    m = open_hip(p, context_length=2048, batch_size=128)
    r.eval(toks)
    m.eval(toks)
    assert chunk_tokens(m) == 1900
    for i in range(40):
        a, b = r.logits.to_numpy(), m.logits.to_numpy()
        assert np.array_equal(a, b), "position %d" % (1900 + i)
        t = int(a.argmax())
        r.eval([t])
        m.eval([t])


@pytest.mark.parametrize("name,tg", [("tiny-q4km", 0), ("tiny-q5km", 0), ("tiny-q4km", 16), ("tiny-q5km", 16), ("tiny-q4km", 32),
                                     ("tiny-q5km", 32)])
def test_prompt_chunk_matrix_core_forms(name, tg, monkeypatch):
    """kernels_pg.h: the token-group size the launch heuristics pick (tg = 0) and 16 / 32 tokens per workgroup forced, on the
    tiny models (Q4_K + Q6_K and Q5_K + Q6_K files; 45 tokens: ragged last group) against the reference's golden one-batch logits."""
    if tg:
        monkeypatch.setenv("CT_AMD_PG_TG", str(tg))
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = open_hip(os.path.join(GOLDEN, name + ".gguf"), batch_size=64)
    m.eval(list(g["long_prompt"]))
    assert np.array_equal(m.logits.to_numpy(), g["long_one"])
    assert chunk_tokens(m) == 45


@pytest.mark.parametrize("shape,ftype", [("llama-7b-2l", "Q4_K_M"), ("llama-70b-2l", "Q5_K_M")])
def test_prompt_chunks_equal_token_by_token_and_reference(ref, tmp_path, monkeypatch, shape, ftype):
    """Real layer widths (K = 4096 / 11008 and K = 8192 / 28672: all three token forms), ragged chunking — 64 + 6 tokens, then
    2, then 1 — three ways: chunk kernels, the decode kernels token by token (CT_AMD_PF=0), the reference CPU build."""
    p = str(tmp_path / "m.gguf")
    hp = synth.write_llama_gguf(p, shape, ftype, seed=5)
    toks = synth.prompt_tokens(73, hp["n_vocab"])
    cuts = ((0, 70), (70, 72), (72, 73))
    r = ref.open_llm(p, context_length=96, batch_size=64, threads=16)
    want = []
    for lo, hi in cuts:
        r.eval(toks[lo:hi])
        want.append((r.logits.to_numpy().copy(), r.embeddings.to_numpy().copy()))
    del r
    for pf, expect_chunked in (("1", 64 + 6 + 2), ("0", 0)):
        monkeypatch.setenv("CT_AMD_PF", pf)
        m = open_hip(p, context_length=96, batch_size=64)
        for (lo, hi), (wl, we) in zip(cuts, want):
            m.eval(toks[lo:hi])
            assert np.array_equal(m.logits.to_numpy(), wl) and np.array_equal(m.embeddings.to_numpy(), we), (pf, lo)
        assert chunk_tokens(m) == expect_chunked
        del m


@pytest.mark.parametrize("shape,ftype,bs", [("llama-7b-2l", "Q4_K_M", 8), ("llama-small", "Q5_K_M", 24), ("falcon-small", "Q4_K_M", 8),
                                            ("llama-small", "Q8_0", 16)])
def test_batches_coalesced_equal_reference_batches(ref, tmp_path, shape, ftype, bs):
    """ctransformers_llm_batch_eval with a small batch_size (the reference's default is 8): the library evaluates the whole
    request in one pass of the chunk kernels and must reproduce the reference's batch-by-batch logits — a 77-token prompt
    (ragged last batch), then a second request at n_past = 77, then greedy steps."""
    p = str(tmp_path / "m.gguf")
    hp = (synth.write_falcon_gguf if shape.startswith("falcon") else synth.write_llama_gguf)(p, shape, ftype, seed=9)
    toks = synth.prompt_tokens(77 + 19, hp["n_vocab"])
    r = ref.open_llm(p, context_length=128, batch_size=bs, threads=16)
    m = open_hip(p, context_length=128, batch_size=bs)
    for lo, hi in ((0, 77), (77, 96)):
        r.eval(toks[lo:hi])
        m.eval(toks[lo:hi])
        assert np.array_equal(r.logits.to_numpy(), m.logits.to_numpy()), lo
        assert np.array_equal(r.embeddings.to_numpy(), m.embeddings.to_numpy()), lo
    assert chunk_tokens(m) == 96
    for i in range(4):
        t = int(r.logits.to_numpy().argmax())
        r.eval([t])
        m.eval([t])
        assert np.array_equal(r.logits.to_numpy(), m.logits.to_numpy()), "step %d" % i


def test_batch_coalescing_many_shapes(ref, tmp_path):
    """Random request shapes against the reference on one pair of handles: request lengths 1..40, batch sizes 1..17 (also
    larger than the request), running n_past, a rollback to an earlier position in between."""
    p = str(tmp_path / "m.gguf")
    hp = synth.write_llama_gguf(p, "llama-tiny", "Q4_K_M", seed=11)
    rng = np.random.default_rng(5)
    r = ref.open_llm(p, context_length=256, batch_size=8, threads=8)
    m = open_hip(p, context_length=256, batch_size=8)
    n_done = 0
    for case in range(14):
        n = int(rng.integers(1, 41))
        bs = int(rng.integers(1, 18))
        if n_done + n > 250 or case == 7:   # roll back: overwrite the KV cache from an earlier position on
            n_done = int(rng.integers(0, 20))
            r._context = r._context[:n_done]
            m._context = m._context[:n_done]
        toks = [int(t) for t in rng.integers(3, hp["n_vocab"], size=n)]
        r.eval(toks, batch_size=bs)
        m.eval(toks, batch_size=bs)
        n_done += n
        assert np.array_equal(r.logits.to_numpy(), m.logits.to_numpy()), (case, n, bs, n_done)
        assert np.array_equal(r.embeddings.to_numpy(), m.embeddings.to_numpy()), (case, n, bs, n_done)


def test_batch_structure_matters(ref, tmp_path, monkeypatch):
    """Test of the tests above: with the batch structure deliberately ignored (CT_AMD_DBG_ONE_BATCH=1: the coalesced request
    treated as ONE reference batch) the logits of the same 77-token, batch_size-8 request differ from the reference's — so
    the equality asserted by test_batches_coalesced_equal_reference_batches does exercise the per-token batch end."""
    p = str(tmp_path / "m.gguf")
    hp = synth.write_llama_gguf(p, "llama-7b-2l", "Q4_K_M", seed=9)
    toks = synth.prompt_tokens(77, hp["n_vocab"])
    r = ref.open_llm(p, context_length=128, batch_size=8, threads=16)
    r.eval(toks)
    monkeypatch.setenv("CT_AMD_DBG_ONE_BATCH", "1")
    m = open_hip(p, context_length=128, batch_size=8)
    m.eval(toks)
    assert not np.array_equal(r.logits.to_numpy(), m.logits.to_numpy())


def _stage_count(m):
    import ctypes
    f = m._lib.ctamd_n_stages
    f.restype, f.argtypes = ctypes.c_int, [ctypes.c_void_p]
    return int(f(m._llm))


def _handoff(m):
    import ctypes
    m._lib.ctamd_handoff.restype, m._lib.ctamd_handoff.argtypes = ctypes.c_char_p, [ctypes.c_void_p]
    return m._lib.ctamd_handoff(m._llm).decode()


def _force_handoff(monkeypatch, form):
    """Stages that share a device share ONE stream ("stream", the default there); with one stream per stage they hand over by events; "flag" — the form stages on distinct devices take: rows stored
    into the next stage's buffer by a kernel, the next stage's stream waiting on a sequence word — is forced for the test, with the
    two-launch decode form (the runtime serves the stream wait with a polling wave on the device, which the fused launch's residency
    does not survive: csrc/pipeline.cc)."""
    if form in ("stream", "stream-stage-graphs"):   # the default of stages that share a device (round 5): ONE stream, the hand-off is stream order,
        monkeypatch.delenv("CT_AMD_HANDOFF", raising=False)   # a decode step of all stages ONE graph ("stream-stage-graphs": a graph per stage)
        monkeypatch.delenv("CT_AMD_PP_SHARED_STREAM", raising=False)
        monkeypatch.setenv("CT_AMD_PP_ONE_GRAPH", "0" if form == "stream-stage-graphs" else "1")
        return
    monkeypatch.setenv("CT_AMD_PP_SHARED_STREAM", "0")   # one stream per stage: the cross-stream forms
    monkeypatch.setenv("CT_AMD_HANDOFF", form)
    if form == "flag":
        monkeypatch.setenv("CT_AMD_FUSE_QA", "0")


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["stream", "stream-stage-graphs", "event", "flag"])
@pytest.mark.parametrize("name", ["tiny-q4km", "falcon-tiny-q4km"])
def test_inprocess_pipeline_on_gpu(name, form, monkeypatch):
    """The in-process pipeline of the library (csrc/pipeline.cc) on hardware: CT_AMD_DEVICES=0,0 puts two stages on the one GPU of
    the test box — their own streams and both hand-off forms are the N-GPU code path.  Goldens of the
    reference build: prompt (reference batches 8 + 3, micro-batches of 4), greedy steps."""
    monkeypatch.setenv("CT_AMD_DEVICES", "0,0")
    monkeypatch.setenv("CT_AMD_PP_MB", "4")
    _force_handoff(monkeypatch, form)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = open_hip(os.path.join(GOLDEN, name + ".gguf"))
    assert _stage_count(m) == 2 and _handoff(m) == form.split("-")[0]
    m.eval(list(g["prompt"]))
    assert np.array_equal(m.logits.to_numpy(), g["logits"][0])
    assert np.array_equal(m.embeddings.to_numpy(), g["embeddings"][0])
    for i, t in enumerate(g["greedy"]):
        assert m.sample(top_k=1, repetition_penalty=1.0) == int(t)
        m.eval([int(t)])
        assert np.array_equal(m.logits.to_numpy(), g["logits"][i + 1]), "step %d" % i


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["stream", "stream-stage-graphs", "event", "flag"])
def test_inprocess_pipeline_7b_widths_vs_reference(ref, tmp_path, monkeypatch, form):
    """Two real-width 7B layers, one per stage: 40-token prompt in reference batches of 8 (micro-batches of 16) + greedy steps against
    the reference build on the same file, with either hand-off form."""
    monkeypatch.setenv("CT_AMD_DEVICES", "0,0")
    monkeypatch.setenv("CT_AMD_PP_MB", "16")
    _force_handoff(monkeypatch, form)
    p = str(tmp_path / "m.gguf")
    hp = synth.write_llama_gguf(p, "llama-7b-2l", "Q4_K_M", seed=5)
    cfg = dict(context_length=128, batch_size=8, threads=8)
    m = open_hip(p, **cfg)
    r = ref.open_llm(p, **cfg)
    toks = synth.prompt_tokens(40, hp["n_vocab"])
    m.eval(toks)
    r.eval(toks)
    for _ in range(4):
        a, b = r.logits.to_numpy(), m.logits.to_numpy()
        assert np.array_equal(a, b)
        t = int(a.argmax())
        m.eval([t])
        r.eval([t])


@pytest.mark.gpu
def test_inprocess_pipeline_eight_stages_70b_widths(ref, tmp_path, monkeypatch):
    """The stage count north_star names (8), on the one GPU of the test box: CT_AMD_DEVICES=0,0,0,0,0,0,0,0 over an 8-layer slice at
    the Llama-2-70B widths (GQA 64/8, K = 8192 / 28672, Q5_K + Q6_K) — partition_layers at eight stages (the last one also streams the
    head), the default 32-token micro-batch of pipelines beyond four stages, seven hand-offs and event waits per micro-batch and
    per token, per-stage chunk graphs keyed by micro-batch offset — against the reference build on the same file: a 72-token prompt
    (three micro-batches, the last one ragged; twice more so the stages replay their graphs) and greedy steps."""
    monkeypatch.setenv("CT_AMD_DEVICES", "0,0,0,0,0,0,0,0")
    monkeypatch.delenv("CT_AMD_PP_MB", raising=False)
    p = str(tmp_path / "m.gguf")
    hp = synth.write_llama_gguf(p, "llama-70b-2l", "Q5_K_M", seed=9, overrides=dict(n_layer=8))
    cfg = dict(context_length=128, batch_size=128, threads=16)
    r = ref.open_llm(p, **cfg)
    toks = synth.prompt_tokens(72, hp["n_vocab"])
    r.eval(toks)
    want = [np.array(r.logits.to_numpy(), copy=True)]
    for _ in range(4):
        t = int(want[-1].argmax())
        r.eval([t])
        want.append(np.array(r.logits.to_numpy(), copy=True))
    del r
    m = open_hip(p, **cfg)
    assert _stage_count(m) == 8
    import ctypes
    m._lib.ctamd_stage_range.restype = ctypes.c_int
    m._lib.ctamd_stage_range.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    covered = []
    for s in range(8):
        a, b = ctypes.c_int(), ctypes.c_int()
        assert m._lib.ctamd_stage_range(m._llm, s, ctypes.byref(a), ctypes.byref(b)) == 0 and b.value > a.value
        covered += list(range(a.value, b.value))
    assert covered == list(range(8))          # contiguous, every layer exactly once, no empty stage
    for rep in range(3):
        m._context = []
        m.eval(toks)
        assert np.array_equal(m.logits.to_numpy(), want[0]), "prompt pass %d" % rep
    for i in range(4):
        t = m.sample(top_k=1, repetition_penalty=1.0)
        assert t == int(want[i].argmax())
        m.eval([t])
        assert np.array_equal(m.logits.to_numpy(), want[i + 1]), "step %d" % i
