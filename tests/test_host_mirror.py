"""Host-side mirror of the reference's Python interface: streaming / stop-sequence logic (modelled on the reference's
own tests/test_llm.py:27-54 with a mock in place of the library), Config defaults, GGUF io, synthetic-model writer."""
import numpy as np
import pytest

from tools import gguf as G, synth
from ctransformers_amd.llm import LLM, Config, utf8_split_incomplete


class MockLLM(LLM):
    def __init__(self, pieces):
        self._config = Config()
        self._pieces = pieces
        self._llm = None
        self._lib = None
        self._context = []

    def tokenize(self, text, add_bos_token=None):
        return [0]

    def generate(self, tokens, **kw):
        yield from range(len(self._pieces))

    def detokenize(self, tokens, decode=True):
        out = b"".join(self._pieces[t].encode() for t in tokens)
        return out.decode() if decode else out


@pytest.mark.parametrize("stop,expect", [
    (None, "the quick brown fox"),
    (["brown"], "the quick "),
    (["ck br"], "the qui"),          # stop sequence spanning two tokens
    (["zzz", "fox"], "the quick brown "),
    (["the"], ""),
])
def test_stop_sequences(stop, expect):
    m = MockLLM(["the ", "qui", "ck ", "br", "own ", "fox"])
    assert m("prompt", stop=stop) == expect
    assert "".join(m("prompt", stop=stop, stream=True)) == expect


def test_max_new_tokens_and_utf8():
    m = MockLLM(["a", "b", "c", "d"])
    assert m("x", max_new_tokens=2) == "ab"
    assert utf8_split_incomplete("é".encode()[:1]) == (b"", "é".encode()[:1])
    assert utf8_split_incomplete(b"ab") == (b"ab", b"")


def test_config_defaults_match_reference():
    c = Config()
    assert (c.top_k, c.top_p, c.temperature, c.repetition_penalty, c.last_n_tokens, c.seed) == (40, 0.95, 0.8, 1.1, 64, -1)
    assert (c.batch_size, c.threads, c.max_new_tokens, c.context_length, c.gpu_layers) == (8, -1, 256, -1, 0)


def test_gguf_roundtrip_and_mix(tmp_path):
    p = str(tmp_path / "m.gguf")
    hp = synth.write_llama_gguf(p, "llama-tiny", "Q4_K_M", seed=9)
    f = G.GGUFFile(p)
    assert f.kv["general.architecture"] == "llama" and int(f.kv["llama.block_count"]) == hp["n_layer"]
    assert f.tensors["output.weight"][1] == G.Q6_K and f.tensors["blk.0.attn_q.weight"][1] == G.Q4_K
    assert f.tensors["blk.1.attn_v.weight"][1] == G.Q6_K  # use_more_bits layer
    shape, t, data = f.tensors["blk.0.ffn_down.weight"]
    assert shape == (hp["n_ff"], hp["n_embd"]) and data.size == G.tensor_nbytes(t, shape)
    # the Q4_K_M mix at the real 7B shape reproduces the survey's bytes/token without writing 4 GB
    types = synth.llama_tensor_types("Q4_K_M", 32)
    sh = synth.LLAMA_SHAPES["llama-2-7b"]
    tot = 0
    for name, t in types.items():
        if name == "token_embd.weight":
            tot += G.row_bytes(t, sh["n_embd"])
            continue
        rows, K = {"attn_q": (4096, 4096), "attn_k": (4096, 4096), "attn_v": (4096, 4096), "attn_output": (4096, 4096),
                   "ffn_gate": (11008, 4096), "ffn_up": (11008, 4096), "ffn_down": (4096, 11008),
                   "output": (32000, 4096)}[name.split(".")[-2] if name.startswith("blk") else "output"]
        tot += rows * G.row_bytes(t, K)
    assert abs(tot - 4.0065e9) < 2e6


def test_quantizers_roundtrip():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((4, 512)).astype(np.float32)
    for t, tol in ((G.Q4_K, 0.08), (G.Q5_K, 0.04), (G.Q6_K, 0.02), (G.Q8_0, 0.006), (G.Q4_0, 0.1)):
        y = synth.dequantize(synth.quantize(x, t), t, 512)
        assert np.abs(y - x).max() / np.abs(x).max() < tol
