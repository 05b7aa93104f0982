"""Prompt chunks of K-quant models on the f16 matrix cores (ctransformers_amd/csrc/kernels_pg.h), run through the CPU emulation of
the HIP sources: golden logits of the real reference build, every token-group size, ragged groups."""
import ctypes
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_emu_engine import open_emu, chunk_tokens


def pg_launches(lib):
    f = lib.ctamd_pg_launches
    f.restype, f.argtypes = ctypes.c_longlong, []
    return int(f())


@pytest.mark.parametrize("name", ["tiny-q4km", "tiny-q5km", "falcon-tiny-q4km", "falcon-tiny7-q4km"])
def test_prompt_on_f16_matrix_cores_matches_reference(emu_lib, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = open_emu(emu_lib, name)
    n0 = pg_launches(m._lib)
    m.eval(list(g["prompt"]))
    assert pg_launches(m._lib) > n0 and chunk_tokens(m) == len(g["prompt"])
    assert np.array_equal(m.logits.to_numpy(), g["logits"][0])
    assert np.array_equal(m.embeddings.to_numpy(), g["embeddings"][0])
    t = m.sample(top_k=1, repetition_penalty=1.0)
    m.eval([t])
    assert np.array_equal(m.logits.to_numpy(), g["logits"][1])


@pytest.mark.parametrize("name,tg,batch,key", [("tiny-q4km", 16, 64, "long_one"), ("tiny-q4km", 32, 8, "long_chunked"),
                                               ("tiny-q5km", 32, 64, "long_one")])
def test_token_group_sizes_and_ragged_groups(emu_lib, monkeypatch, name, tg, batch, key):
    """45 tokens = 2 groups of 32 (13 live slots in the second) or 3 of 16: same bits as the reference."""
    monkeypatch.setenv("CT_AMD_PG_TG", str(tg))
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = open_emu(emu_lib, name, batch_size=batch)
    n0 = pg_launches(m._lib)
    m.eval(list(g["long_prompt"]))
    assert pg_launches(m._lib) > n0 and chunk_tokens(m) == 45
    assert np.array_equal(m.logits.to_numpy(), g[key])


@pytest.mark.parametrize("name,mode", [("tiny-q4km", "tile"), ("tiny-q4km", "long"), ("tiny-q4km", "wave"), ("tiny-q4km", "fused"),
                                       ("falcon-tiny-q4km", "tile"), ("tiny-q5km", "wave")])   # every golden test at context 96 takes "long" as well
def test_chunk_attention_kernels(emu_lib, monkeypatch, name, mode):
    """The four chunk-attention kernels (kernels_exact.h): K/V of a head in LDS for 16 tokens (all positions below 128; needs
    n_ctx >= 128), K/V tiles of 64 positions through LDS for any position (contexts up to 4096), a wave per (head, token), and the
    decode kernel with all channels per workgroup — same goldens of the reference."""
    if mode != "tile":
        monkeypatch.setenv("CT_AMD_ATTN_TILE", "0")
    if mode in ("wave", "fused"):
        monkeypatch.setenv("CT_AMD_ATTN_LONG", "0")
    if mode == "fused":
        monkeypatch.setenv("CT_AMD_ATTN_WAVE", "0")
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = open_emu(emu_lib, name, context_length=160, batch_size=8)
    if "long_prompt" in g.files:
        m.eval(list(g["long_prompt"]))   # 45 tokens in reference batches of 8: ragged batch ends (leftover positions of V*P), one chunk
        assert np.array_equal(m.logits.to_numpy(), g["long_chunked"])
    else:
        m.eval(list(g["prompt"]))
        assert np.array_equal(m.logits.to_numpy(), g["logits"][0])


@pytest.mark.parametrize("name,n_prompt,bs,tile_first,ctx", [("tiny-q4km", 139, 8, "0", 2500),   # 2500: 8 tokens per workgroup
                                                             ("falcon-tiny-q4km", 139, 64, "1", 256)])
def test_chunk_attention_across_position_tiles(emu_lib, ref, monkeypatch, name, n_prompt, bs, tile_first, ctx):
    """Prompts longer than one chunk and one 64-position tile: the later chunk attends through several K / V tiles with the
    accumulators carried from tile to tile (attn_chunk_long_kernel), ragged reference batch ends included; with the 128-position
    kernel switched off the first chunk takes the tiled kernel as well.  Against the reference build on the same file."""
    from tools import synth
    monkeypatch.setenv("CT_AMD_ATTN_TILE", tile_first)
    path = os.path.join(GOLDEN, name + ".gguf")
    toks = synth.prompt_tokens(n_prompt, 512)
    r = ref.open_llm(path, context_length=ctx, batch_size=bs, threads=4)
    m = open_emu(emu_lib, name, context_length=ctx, batch_size=bs)
    r.eval(toks)
    m.eval(toks)
    assert chunk_tokens(m) == n_prompt
    assert np.array_equal(r.logits.to_numpy(), m.logits.to_numpy())
    t = int(r.logits.to_numpy().argmax())
    r.eval([t]); m.eval([t])
    assert np.array_equal(r.logits.to_numpy(), m.logits.to_numpy())
