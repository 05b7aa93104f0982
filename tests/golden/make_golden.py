"""Regenerates the committed golden vectors from the REAL reference build (oracle/_ref, built from /root/reference by
oracle/Makefile).  Run in the build container:  python tests/golden/make_golden.py
Outputs (small, committed):
  tiny-q4km / -q5km / -q80 / -q40 .gguf   synthetic llama-tiny models (the files themselves, so fixtures do not depend on numpy RNG)
  tiny-*.npz                        reference logits / embeddings / greedy tokens / sampled tokens for those models
  ops.npz                           op-level vectors: Q8_K/Q8_0 activation quantization, the five weight dot products,
                                    rope, rms_norm*w, scale+softmax, fp16 mat-mul, silu*mul
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tools import gguf as G, synth  # noqa: E402
from oracle import ref  # noqa: E402


def model_golden(name, ftype, seed, shape="llama-tiny", quantizer=None):
    path = os.path.join(HERE, name + ".gguf")
    mt = None
    if shape.startswith("gpt2"):
        path = os.path.join(HERE, name + ".bin")     # legacy GGML container
        hp = synth.write_gpt2_ggml(path, shape, seed=seed)
        mt = "gpt2"
    elif shape.startswith("starcoder"):
        path = os.path.join(HERE, name + ".bin")     # same container, read by the reference's starcoder loader
        hp = synth.write_gpt2_ggml(path, shape, seed=seed, ftype={"Q4_0": 2, "Q8_0": 7}[ftype], pieces=synth.STARCODER_PIECES)
        mt = "starcoder"
    elif shape.startswith("mpt"):
        path = os.path.join(HERE, name + ".bin")     # legacy container with the MPT header (models/llms/mpt.cc)
        hp = synth.write_mpt_ggml(path, shape, seed=seed, ftype={"Q4_0": 2, "Q8_0": 7}[ftype], pieces=[b"<|endoftext|>"])
        mt = "mpt"
    elif shape.startswith("falcon"):
        hp = synth.write_falcon_gguf(path, shape, ftype, seed=seed)
    else:
        hp = synth.write_llama_gguf(path, shape, ftype, seed=seed, quantizer=quantizer)   # "reference": blocks from ggml_quantize_chunk itself
    cfg = dict(context_length=96, batch_size=8, threads=4, model_type=mt)
    r = ref.open_llm(path, **cfg)
    prompt = synth.prompt_tokens(11, hp["n_vocab"])
    r.eval(prompt)
    logits = [r.logits.to_numpy().copy()]
    emb = [r.embeddings.to_numpy().copy()]   # empty for legacy models
    toks = []
    for _ in range(40):
        t = r.sample(top_k=1, repetition_penalty=1.0)
        toks.append(t)
        r.eval([t])
        logits.append(r.logits.to_numpy().copy())
        emb.append(r.embeddings.to_numpy().copy())
    # sampler chain on the final logits (host-side path of the ABI)
    samples = []
    for seed_s, (k, p, temp, pen) in enumerate([(40, 0.95, 0.8, 1.1), (5, 0.5, 1.3, 1.0), (100, 1.0, 0.7, 1.3)]):
        samples.append([k, p, temp, pen, seed_s + 11,
                        r.sample(top_k=k, top_p=p, temperature=temp, repetition_penalty=pen, last_n_tokens=64, seed=seed_s + 11)])
    # batch structure: a 45-token prompt as ONE batch vs in chunks of 8 (the reference's default batch_size) — the
    # results differ in the last bits once n_past+N crosses 32 (vec_dot_f16's fma/leftover split), both are golden.
    long_prompt = synth.prompt_tokens(45, hp["n_vocab"])
    r2 = ref.open_llm(path, context_length=96, batch_size=64, threads=4, model_type=mt)
    r2.eval(long_prompt)
    long_one = r2.logits.to_numpy().copy()
    r3 = ref.open_llm(path, context_length=96, batch_size=8, threads=4, model_type=mt)
    r3.eval(long_prompt)
    long_chunked = r3.logits.to_numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), prompt=np.array(prompt, dtype=np.int32),
                        greedy=np.array(toks, dtype=np.int32), logits=np.array(logits), embeddings=np.array(emb),
                        samples=np.array(samples, dtype=np.float64), context=np.array(list(r._context), dtype=np.int32),
                        long_prompt=np.array(long_prompt, dtype=np.int32), long_one=long_one, long_chunked=long_chunked)
    print(name, 'one-batch vs chunked identical:', np.array_equal(long_one, long_chunked))
    print(name, "greedy head:", toks[:8])


def big_batch_golden():
    """Requests longer than 128 tokens evaluated as ONE reference batch (batch_size > 128): every token's V*P dot runs over the whole batch
    (llama.cpp:2373-2378).  tiny-q4km-refq-batch.npz: logits for a 140-token request (batch 160, context 192: the emulator's size) and a 200-token
    one (batch 256, context 320)."""
    path = os.path.join(HERE, "tiny-q4km-refq.gguf")
    out = {}
    for n, bs, ctx in ((140, 160, 192), (200, 256, 320)):
        r = ref.open_llm(path, context_length=ctx, batch_size=bs, threads=4)
        toks = synth.prompt_tokens(n, 512)
        r.eval(toks)
        out["logits_%d" % n] = r.logits.to_numpy().copy()
        out["prompt_%d" % n] = np.array(toks, dtype=np.int32)
        out["cfg_%d" % n] = np.array([bs, ctx], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "tiny-q4km-refq-batch.npz"), **out)
    print("tiny-q4km-refq-batch: ok")


def ops_golden():
    rng = np.random.default_rng(2024)
    ops = ref.GgmlOps()
    out = {}
    K = 1024
    x = (rng.standard_normal(K) * 1.7).astype(np.float32)
    x[5] = -x[300]  # |x| tie inside one block with opposite signs
    x[512:768] = 0.0  # an all-zero block
    out["act_x"] = x
    out["act_q8_K"], _ = ref.quantize_activation(x, G.Q4_K)
    out["act_q8_0"], _ = ref.quantize_activation(x, G.Q8_0)
    for t in (G.Q4_K, G.Q5_K, G.Q6_K, G.Q8_0, G.Q4_0):
        w = rng.standard_normal((24, K), dtype=np.float32) * 0.2
        raw = synth.quantize(w, t)
        out["w_%s" % G.TYPE_NAMES[t]] = raw
        out["y_%s" % G.TYPE_NAMES[t]] = ref.matvec(t, raw, x, K)
        out["deq_%s" % G.TYPE_NAMES[t]] = ref.dequantize(raw[:2], t, K)
    h = rng.standard_normal((3, 4, 64)).astype(np.float32)
    out["rope_x"] = h
    out["rope_pos"] = np.array([0, 17, 511], dtype=np.int32)
    out["rope_y"] = np.stack([ops.rope(h[i][None], int(p))[0] for i, p in enumerate(out["rope_pos"])])
    nx = rng.standard_normal(256).astype(np.float32) * 3
    nw = (1 + 0.1 * rng.standard_normal(256)).astype(np.float32)
    out["norm_x"], out["norm_w"], out["norm_y"] = nx, nw, ops.rms_norm_mul(nx, nw, 1e-5)
    sc = rng.standard_normal((4, 77)).astype(np.float32) * 4
    out["sm_x"], out["sm_scale"], out["sm_y"] = sc, np.float32(0.125), ops.scale_softmax(sc, 0.125)
    a = (rng.standard_normal((5, 77)) * 0.5).astype(np.float16)
    b = rng.random((2, 77)).astype(np.float32)
    out["mm_a"], out["mm_b"], out["mm_y"] = a.view(np.uint16), b, ops.mul_mat_f16(a, b)
    g, u = rng.standard_normal(300).astype(np.float32) * 3, rng.standard_normal(300).astype(np.float32)
    out["silu_g"], out["silu_u"], out["silu_y"] = g, u, ops.silu_mul(g, u)
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **out)


def falcon_ops_golden():
    """Op-level vectors for the falcon-only ops: LayerNorm*w+b, neox RoPE, GELU through the fp16 table."""
    rng = np.random.default_rng(4040)
    ops = ref.GgmlOps()
    out = {}
    x = (rng.standard_normal(768) * 2.5 + 0.3).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(768)).astype(np.float32)
    b = (0.2 * rng.standard_normal(768)).astype(np.float32)
    out["ln_x"], out["ln_w"], out["ln_b"], out["ln_y"] = x, w, b, ops.norm_mul_add(x, w, b, 1e-5)
    h = rng.standard_normal((3, 6, 64)).astype(np.float32)
    out["neox_x"], out["neox_pos"] = h, np.array([0, 17, 301], dtype=np.int32)
    out["neox_y"] = np.stack([ops.rope(h[i][None], int(p), mode=2)[0] for i, p in enumerate(out["neox_pos"])])
    g = np.concatenate([(rng.standard_normal(1500) * 3).astype(np.float32), np.array([0.0, -0.0, 1e-8, 65504.0, -65504.0, 1e6], np.float32)])
    out["gelu_x"], out["gelu_y"] = g, ops.gelu(g)
    np.savez_compressed(os.path.join(HERE, "falcon_ops.npz"), **out)


SPM_TEXTS = ["hello world", "Hello, world!", "the and in that when", " hello", "hello  world   twice", "naïve café", "12345", "",
             "a", "unknownzq", "▁already▁marked", "when you hear the thing", "\ttab and\nnewline", "日本"]
BPE_TEXTS = ["ab cd\n", "Hello, world! it's 12345 times   spaced\n\nxyz abcab", " ", "a", "don't you're I'll\tTAB", "héllo ünicode ✓ 中文", "ababababab abc", "abab cdcd abcd"]
STARCODER_TEXTS = ["<fim-prefix>ab cd<fim-suffix>xy<fim-middle>", "<|system|>\nab<|end|>\n<|user|>\ncd ef<|end|>\n<|assistant|>", "<|end_of_turn|> <fim-pad>",
                   "ab<|endoftext|>cd", "<<|user|>>", "<fim-prefix", "plain text, no markers", "<|end|><|end|>", "", "<|user|>"]
GPT2_TEXTS = ["ab cd xyz\n", "Hello, it's 42!  aaa", " ", "don't stop\tnow", "héllo ✓", "xyzabc  qq"]


def tokenizer_golden():
    """Host-path vectors produced by the reference build: sentencepiece (score-ordered merges, byte fallback, the leading-space
    rule), falcon's byte-level BPE, and the legacy gpt2 tokenizer + its sampler.  The SPM model file is committed (tiny, one layer:
    only its vocabulary matters)."""
    import json
    p = os.path.join(HERE, "spm-vocab.gguf")
    synth.write_llama_gguf(p, "llama-tiny", "Q4_K_M", seed=12, overrides=dict(n_layer=1), vocab=synth.make_spm_vocab(512))
    r = ref.open_llm(p, context_length=32, batch_size=8, threads=2)
    spm = {t: [int(i) for i in r.tokenize(t)] for t in SPM_TEXTS}
    spm_detok = {t: r.detokenize(r.tokenize(t)) for t in SPM_TEXTS}
    json.dump(dict(tokenize=spm, detokenize=spm_detok), open(os.path.join(HERE, "spm_golden.json"), "w"), ensure_ascii=False, indent=1)
    r = ref.open_llm(os.path.join(HERE, "falcon-tiny-q4km.gguf"), context_length=96, batch_size=8, threads=2)
    json.dump({t: [int(i) for i in r.tokenize(t)] for t in BPE_TEXTS}, open(os.path.join(HERE, "falcon_bpe.json"), "w"), ensure_ascii=False, indent=1)
    r = ref.open_llm(os.path.join(HERE, "gpt2-tiny-q40.bin"), model_type="gpt2", context_length=96, batch_size=8, threads=2)
    g = np.load(os.path.join(HERE, "gpt2-tiny-q40.npz"))
    tok = {t: [int(i) for i in r.tokenize(t)] for t in GPT2_TEXTS}
    r.eval(list(g["prompt"]))
    samples = []
    for seed_s, (k, p_, temp, pen) in enumerate([(40, 0.95, 0.8, 1.1), (5, 0.5, 1.3, 1.0), (100, 1.0, 0.7, 1.3), (1, 1.0, 1.0, 1.0), (200, 0.9, 1.5, 1.2)]):
        samples.append([k, p_, temp, pen, seed_s + 3, int(r.sample(top_k=k, top_p=p_, temperature=temp, repetition_penalty=pen, seed=seed_s + 3))])
    json.dump(dict(tokenize=tok, samples=samples, detok_300_10=r.detokenize([300, 10])), open(os.path.join(HERE, "gpt2_host.json"), "w"),
              ensure_ascii=False, indent=1)
    r = ref.open_llm(os.path.join(HERE, "mpt-tiny-q80.bin"), model_type="mpt", context_length=96, batch_size=8, threads=2)
    g = np.load(os.path.join(HERE, "mpt-tiny-q80.npz"))
    tok = {t: [int(i) for i in r.tokenize(t)] for t in GPT2_TEXTS + ["ab<|endoftext|>cd"]}
    r.eval(list(g["prompt"]))
    samples = []
    for seed_s, (k, p_, temp, pen) in enumerate([(40, 0.95, 0.8, 1.1), (5, 0.5, 1.3, 1.0), (1, 1.0, 1.0, 1.0), (200, 0.9, 1.5, 1.2)]):
        samples.append([k, p_, temp, pen, seed_s + 5, int(r.sample(top_k=k, top_p=p_, temperature=temp, repetition_penalty=pen, seed=seed_s + 5))])
    json.dump(dict(tokenize=tok, samples=samples, eos=int(r.eos_token_id), detok=r.detokenize([300, 233, 10]),
                   ctx_default=int(ref.open_llm(os.path.join(HERE, "mpt-tiny128-q40.bin"), model_type="mpt").context_length),
                   ctx_capped=int(ref.open_llm(os.path.join(HERE, "mpt-tiny-q80.bin"), model_type="mpt", context_length=4096).context_length)),
              open(os.path.join(HERE, "mpt_host.json"), "w"), ensure_ascii=False, indent=1)
    sc = os.path.join(HERE, "starcoder-tiny-q80.bin")
    host = {}
    for mt in ("starcoder", "gpt_bigcode", "gpt2"):   # gpt2 on the same file: no special pieces registered
        r = ref.open_llm(sc, model_type=mt, context_length=96, batch_size=8, threads=2)
        host[mt] = dict(tokenize={t: [int(i) for i in r.tokenize(t)] for t in STARCODER_TEXTS}, eos=int(r.eos_token_id), model_type=r.model_type)
    json.dump(host, open(os.path.join(HERE, "starcoder_host.json"), "w"), ensure_ascii=False, indent=1)
    print("tokenizer goldens:", {k: v for k, v in list(spm.items())[:4]})


if __name__ == "__main__":
    only = sys.argv[1:]   # e.g. `make_golden.py tiny-q80 tiny-q40` regenerates just those
    for name, ftype, seed, shape in (("tiny-q4km", "Q4_K_M", 3, "llama-tiny"), ("tiny-q5km", "Q5_K_M", 4, "llama-tiny"),
                                     ("tiny-q80", "Q8_0", 5, "llama-tiny"), ("tiny-q40", "Q4_0", 6, "llama-tiny"),
                                     ("falcon-tiny-q4km", "Q4_K_M", 7, "falcon-tiny"),      # 40B style: two norms, GQA 4/2
                                     ("falcon-tiny7-q4km", "Q4_K_M", 8, "falcon-tiny7"),    # 7B style: one norm, MQA 4/1
                                     ("gpt2-tiny-q40", "Q4_0", 9, "gpt2-tiny"),             # config 1 family: legacy GGML, F32 KV
                                     ("starcoder-tiny-q80", "Q8_0", 10, "starcoder-tiny"),  # same container, starcoder loader
                                     ("mpt-tiny-q80", "Q8_0", 11, "mpt-tiny"),              # ALiBi, clamp, 6 heads of 64
                                     ("mpt-tiny128-q40", "Q4_0", 12, "mpt-tiny128")):       # heads of 128, no clamp
        if not only or name in only:
            model_golden(name, ftype, seed, shape)
    # the same llama-tiny Q4_K_M mix with every block out of the reference's OWN quantizer (ggml_quantize_chunk: Q6_K scales and d of
    # either sign, make_qkx1_quants' scale searches) — what files in the field hold; tools/synth.py's numpy quantizers never emit those
    if not only or "tiny-q4km-refq" in only:
        model_golden("tiny-q4km-refq", "Q4_K_M", 13, "llama-tiny", quantizer="reference")
    if not only or "batch" in only:
        big_batch_golden()
    if not only or "ops" in only:
        ops_golden()
    if not only or "falcon_ops" in only:
        falcon_ops_golden()
    if not only or "tokenizers" in only:
        tokenizer_golden()
    print("golden vectors written to", HERE)
