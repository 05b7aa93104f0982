"""Q8_0 / Q4_0 rows longer than 12288 elements (the K = 4 d_model down projections of MPT-7B, StarCoder-7B / -15B): the
MAXK = 32768 instantiation of the decode mat-vec (kernels_v9.h, 32-block records) and the chunk kernel with 8 (K <= 16384) or 4 (K <= 32768)
token images per workgroup, against the reference build on the same file.  The llama graph lets n_ff be wide while everything
else stays small, so the cases run in seconds on the emulator build; the full MPT-7B / StarCoder widths run on the GPU
(tests/test_gpu_parity.py)."""
import os

import numpy as np
import pytest

from tools import synth
from ctransformers_amd.llm import LLM, Config

CASES = [("Q8_0", 13312), ("Q4_0", 16384), ("Q8_0", 16512), ("Q4_0", 32768)]


def _run(lib, ref, tmp_path, ftype, n_ff, n_decode, n_prompt=21):
    p = str(tmp_path / "wide.gguf")
    hp = synth.write_llama_gguf(p, "llama-tiny", ftype, seed=31, overrides=dict(n_layer=1, n_ff=n_ff))
    toks = synth.prompt_tokens(n_prompt, hp["n_vocab"])
    r = ref.open_llm(p, context_length=64, batch_size=64, threads=4)
    m = LLM(p, config=Config(context_length=64, batch_size=64, threads=1), lib=lib)
    r.eval(toks)
    m.eval(toks)          # one chunk of 21 tokens: 8 + 8 + 5 (or 4 x 5 + 1) images per workgroup
    for i in range(n_decode):
        a, b = r.logits.to_numpy(), m.logits.to_numpy()
        assert np.array_equal(a, b), "step %d" % i
        t = int(a.argmax())
        r.eval([t])
        m.eval([t])       # the decode kernel
    assert np.array_equal(r.logits.to_numpy(), m.logits.to_numpy())


@pytest.mark.parametrize("ftype,n_ff", [("Q8_0", 13312), ("Q4_0", 32768)])   # one case per chunk kernel; all four run on the GPU
def test_wide_rows_on_emulator_build(emu_lib, ref, tmp_path, ftype, n_ff):
    _run(emu_lib, ref, tmp_path, ftype, n_ff, 1, n_prompt=9)   # 9 tokens: a full group of 8 (4 + 4) and a ragged one


def test_rows_above_32768_are_refused(emu_lib, tmp_path):
    p = str(tmp_path / "too_wide.gguf")
    synth.write_llama_gguf(p, "llama-tiny", "Q8_0", seed=31, overrides=dict(n_layer=1, n_ff=32768 + 128))
    with pytest.raises(RuntimeError):
        LLM(p, config=Config(context_length=64), lib=emu_lib)


@pytest.mark.gpu
@pytest.mark.parametrize("ftype,n_ff", CASES)
def test_wide_rows_on_hip_build(ref, tmp_path, ftype, n_ff):
    _run(None, ref, tmp_path, ftype, n_ff, 12)
