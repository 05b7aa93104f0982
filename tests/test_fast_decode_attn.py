"""Order-free long-context decode attention (ctransformers_amd/csrc/kernels_attn9.h:attn_decode9_free_kernel; CT_AMD_DECODE_ATTN=fast, opt-in).

What it keeps of the reference's chain (ggml.c:2392-2425, :12009-12078): the K.Q dot per position in the reference's lane
order (scores bit-identical), the fp16 input of the exp table, the exact double sum, the fp16 probabilities, V*P as f32 fma steps of 32 positions and
the scalar double tail.  What it gives up: the ORDER in which the 32-position steps of one channel are added (the workgroup's eight waves take them
interleaved; partial sums meet in double).  A launch's output row therefore differs from the bit-identical kernel's by a few f32 ulp; on models
deeper than one layer that moves int8 roundings of the next activation quantization, after which two runs differ by the reference's own quantization
noise — so the bar here is (1) the launch itself on ONE-layer models, where the K / V / Q rows are identical in both runs, and (2) sane
logits on deeper models; the bit-identical kernels stay the default.
"""
import ctypes
import os

import numpy as np
import pytest

from conftest import HIP_LIB, has_gpu
from tools import synth
from ctransformers_amd.llm import LLM, Config


def _free_launches(lib):
    """launch calls of the order-free kernel by this process (a token step replayed from a hipGraph counts once, at its capture)"""
    f = ctypes.CDLL(lib or HIP_LIB).ctamd_attn_free_launches
    f.restype = ctypes.c_longlong
    return int(f())


def _attn_row(m):
    f = m._lib.ctamd_debug_read_attn_out
    f.restype, f.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    buf = np.zeros(16384, np.float32)
    E = f(m._llm, buf.ctypes.data, 0)
    assert E > 0
    return buf[:E].copy()


def _decode_rows(path, lib, fast, n_prompt, n_steps, ctx, monkeypatch, n_vocab, share=True, batch=128):
    """prompt with the default (bit-identical) chunk kernels, then n_steps token steps feeding FIXED tokens (so both runs see the same inputs):
    the decode attention's output row and the logits of every step"""
    monkeypatch.setenv("CT_AMD_PREFILL", "exact")
    monkeypatch.setenv("CT_AMD_ATTN_SHARE", "1" if share else "0")
    if fast:
        monkeypatch.setenv("CT_AMD_DECODE_ATTN", "fast")
    else:
        monkeypatch.delenv("CT_AMD_DECODE_ATTN", raising=False)
    before = _free_launches(lib)
    m = LLM(path, config=Config(context_length=ctx, batch_size=batch, threads=1), lib=lib)
    toks = synth.prompt_tokens(n_prompt + n_steps, n_vocab)
    m.eval(toks[:n_prompt])
    rows, logits = [], []
    for i in range(n_steps):
        m.eval([toks[n_prompt + i]])
        rows.append(_attn_row(m))
        logits.append(m.logits.to_numpy().copy())
    n = _free_launches(lib) - before
    del m
    return np.array(rows), np.array(logits), n


@pytest.mark.parametrize("n_prompt", [12, 31, 70])
def test_order_free_decode_attention_on_the_emulator(emu_lib, n_prompt, tmp_path, monkeypatch):
    """One layer, context 1100 (the long-context form starts above 1024): all positions in the scalar tail (12), the first whole chunk with and without a
    tail (31 -> 32, 33), two chunks and a tail (70).  The emulator runs workgroups one after the other, so every workgroup computes its own score row
    (the form the product falls back to when the device is shared).  (Prompts stay short: an emulated prompt token at this context costs half a second.)"""
    shape = dict(synth.LLAMA_SHAPES["llama-tiny"], n_layer=1)
    path = str(tmp_path / "tiny1l.gguf")
    synth.write_llama_gguf(path, shape, "Q4_K_M", seed=5)
    a, la, n0 = _decode_rows(path, emu_lib, False, n_prompt, 2, 1100, monkeypatch, shape["n_vocab"])
    b, lb, n1 = _decode_rows(path, emu_lib, True, n_prompt, 2, 1100, monkeypatch, shape["n_vocab"])
    assert n0 == 0 and n1 == 2 * shape["n_layer"]   # (no graphs on the emulator: every launch counts)
    assert np.isfinite(b).all()
    rel = float(np.abs(a - b).max() / np.abs(a).max())
    assert rel < 1e-5, rel
    if n_prompt == 12:   # no 32-position step at all: nothing whose order could differ
        assert np.array_equal(a, b) and np.array_equal(la, lb)


def test_default_is_bit_identical_form(emu_lib, tmp_path, monkeypatch):
    """Without the knob no order-free launch is issued (and "exact" / unknown values do not switch it on)."""
    shape = dict(synth.LLAMA_SHAPES["llama-tiny"], n_layer=1)
    path = str(tmp_path / "tiny1l.gguf")
    synth.write_llama_gguf(path, shape, "Q4_K_M", seed=5)
    for v in (None, "exact"):
        if v is None:
            monkeypatch.delenv("CT_AMD_DECODE_ATTN", raising=False)
        else:
            monkeypatch.setenv("CT_AMD_DECODE_ATTN", v)
        before = _free_launches(emu_lib)
        m = LLM(path, config=Config(context_length=1100, batch_size=64, threads=1), lib=emu_lib)
        m.eval(synth.prompt_tokens(6, shape["n_vocab"]))
        m.eval([5])
        assert _free_launches(emu_lib) == before
        del m


# ---- MI355X ---------------------------------------------------------------------------------------------------------------------------------------

def _model_1l(shape):
    falcon = shape.startswith("falcon")
    hp = dict((synth.FALCON_SHAPES if falcon else synth.LLAMA_SHAPES)[shape], n_layer=1)
    path = "/tmp/ctamd_fast_attn_%s_1l.gguf" % shape.replace("-", "_")   # (the file tests/test_fast_prefill.py writes)
    if not os.path.exists(path):
        (synth.write_falcon_gguf if falcon else synth.write_llama_gguf)(path + ".tmp", hp, "Q4_K_M", seed=11)
        os.replace(path + ".tmp", path)
    return path, hp


@pytest.mark.gpu
@pytest.mark.parametrize("shape,n_prompt,ctx,share", [("llama-7b-2l", 2001, 2304, True), ("llama-7b-2l", 1200, 2304, False), ("llama-70b-2l", 2001, 2304, True),
                                                        ("falcon-40b-2l", 1500, 2048, True), ("llama-7b-2l", 5000, 6144, True), ("llama-7b-2l", 20, 2048, True)])
def test_order_free_decode_attention(shape, n_prompt, ctx, share, monkeypatch):
    """One layer at the 7B / 70B / Falcon-40B widths on the GPU (head sizes 128 and 64; 8, 4 and 2 channel-group workgroups per head): the decode
    attention's output rows against the bit-identical kernel's on identical K / V / Q, with the shared score row and with every workgroup computing its
    own; 5000 positions: the V ring's re-request rounds; 20: the scalar tail alone (bit-identical)."""
    path, hp = _model_1l(shape)
    a, la, n0 = _decode_rows(path, None, False, n_prompt, 4, ctx, monkeypatch, hp["n_vocab"], share)
    b, lb, n1 = _decode_rows(path, None, True, n_prompt, 4, ctx, monkeypatch, hp["n_vocab"], share)
    assert n0 == 0 and n1 >= 1
    rel = float(np.abs(a - b).max() / np.abs(a).max())
    rel_logits = float(np.abs(la - lb).max() / np.abs(la).max())
    print("attn_decode9_free_kernel vs the bit-identical decode attention, %s, %d positions: rows differ by %.3g of the largest, logits by %.3g" %
          (shape, n_prompt, rel, rel_logits))
    assert np.isfinite(b).all() and rel < 1e-5, rel
    if n_prompt == 20:
        assert np.array_equal(a, b) and np.array_equal(la, lb)


@pytest.mark.gpu
def test_order_free_decode_attention_two_layers_sane(monkeypatch):
    """Two layers at the 7B widths, greedy steps behind a 1500-token prompt: logits finite and inside the reference's quantization-noise band of the
    bit-identical run (3-5e-2 of the largest logit on synthetic weights), and a residency give-up (forced) is replayed with the
    own-row form of the SAME order-free kernel."""
    from test_fast_prefill import _model
    path = _model("llama-7b-2l", "Q4_K_M", "llama_7b_2l_q4_k_m")
    hp = synth.LLAMA_SHAPES["llama-7b-2l"]
    a, la, _ = _decode_rows(path, None, False, 1500, 6, 2048, monkeypatch, hp["n_vocab"])
    b, lb, n = _decode_rows(path, None, True, 1500, 6, 2048, monkeypatch, hp["n_vocab"])
    assert n >= 1 and np.isfinite(lb).all()
    rel = float(np.abs(la - lb).max() / np.abs(la).max())
    print("two layers, order-free decode attention: logits differ by %.3g of the largest" % rel)
    assert rel < 8e-2, rel
    monkeypatch.setenv("CT_AMD_DBG_QA_TIMEOUT", "3")
    c, lc, n = _decode_rows(path, None, True, 1500, 6, 2048, monkeypatch, hp["n_vocab"])
    monkeypatch.delenv("CT_AMD_DBG_QA_TIMEOUT")
    assert np.isfinite(lc).all() and n >= 1
    rel = float(np.abs(la - lc).max() / np.abs(la).max())
    assert rel < 8e-2, rel
