"""The drop-in boundary: include/ctransformers_llm.h == Python-side ABI table == symbols exported by the HIP library."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, has_gpu
from ctransformers_amd.llm import ABI_SYMBOLS, LLM, Config, ConfigStruct, load_library


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ctransformers_llm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.findall(r"\b(ctransformers_llm_\w+)\s*\(", text)


def test_header_declares_exactly_the_17_reference_symbols():
    syms = header_symbols()
    assert sorted(syms) == sorted(ABI_SYMBOLS)
    assert len(set(syms)) == 17


def test_hip_library_exports_every_header_symbol(hip_lib):
    lib = ctypes.CDLL(hip_lib)  # loads without a GPU; only create() needs a device
    for s in header_symbols():
        assert hasattr(lib, s), s


def test_hip_library_exports_every_extension_symbol(hip_lib):
    """include/ctransformers_amd_ext.h: measurement hooks and the pipeline-stage entry points."""
    text = open(os.path.join(ROOT, "include", "ctransformers_amd_ext.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    syms = re.findall(r"\b(ctamd_\w+)\s*\(", text)
    assert {"ctamd_profile_decode", "ctamd_weight_bytes", "ctamd_trace_site", "ctamd_stage_create", "ctamd_stage_eval",
            "ctamd_stage_eval_batched", "ctamd_n_layer", "ctamd_n_embd", "ctamd_chunk_tokens"} <= set(syms)
    lib = ctypes.CDLL(hip_lib)
    for s in syms:
        assert hasattr(lib, s), s


def test_config_struct_matches_header_layout():
    # struct ctransformers_config { int; int; bool; bool; } -> 12 bytes on x86-64, passed by value
    assert ctypes.sizeof(ConfigStruct) == 12
    assert [f[0] for f in ConfigStruct._fields_] == ["context_length", "gpu_layers", "mmap", "mlock"]


@pytest.mark.skipif(has_gpu(), reason="checks the loud failure on a GPU-less host")
def test_create_fails_loudly_without_gpu(hip_lib, capfd):
    path = os.path.join(ROOT, "tests", "golden", "tiny-q4km.gguf")
    with pytest.raises(RuntimeError):
        LLM(path, config=Config(context_length=64), lib=hip_lib)
    assert "no CPU fallback" in capfd.readouterr().err
    with pytest.raises(RuntimeError):   # legacy GGML (gpt2) path: same loud failure
        LLM(os.path.join(ROOT, "tests", "golden", "gpt2-tiny-q40.bin"), "gpt2", config=Config(context_length=64), lib=hip_lib)
    assert "no CPU fallback" in capfd.readouterr().err


def test_missing_library_raises_not_falls_back(tmp_path):
    with pytest.raises(OSError):
        load_library(str(tmp_path / "nope.so"))
    with pytest.raises(ValueError):
        LLM(str(tmp_path / "nope.gguf"))
