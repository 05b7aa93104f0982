"""Row (e): the multi-GPU layer pipeline.  CPU coverage of the N > 1 path: stage entry points of the C library (emulator
build of the HIP sources), the layer partition, and a world_size-2 gloo run of the real Pipeline driver."""
import ctypes
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from tools import rccl_pipeline as pipeline


def test_partition_layers():
    for n_layer, world in [(32, 1), (32, 2), (32, 4), (32, 8), (80, 8), (60, 4), (2, 2), (3, 2)]:
        b = pipeline.partition_layers(n_layer, world)
        assert len(b) == world and b[0][0] == 0 and b[-1][1] == n_layer
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1)) and all(e > s for s, e in b)
        sizes = [e - s for s, e in b]
        assert max(sizes) - min(sizes) <= 1
        if world > 1 and n_layer % world:
            assert sizes[-1] == min(sizes)  # the stage that also streams the lm_head gets the short end
    assert pipeline.partition_layers(32, 8) == [(4 * i, 4 * i + 4) for i in range(8)]
    with pytest.raises(ValueError):
        pipeline.partition_layers(2, 3)


def test_stage_chain_equals_whole_model(emu_lib, mirror):
    """stage[0,1) -> stage[1,2) through the C stage API == the whole model, bit for bit; the hand-off rows equal the
    oracle's residual stream after layer 0."""
    path = os.path.join(GOLDEN, "tiny-q4km.gguf")
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    lib = ctypes.CDLL(emu_lib)
    s0 = pipeline.HipStage(path, 0, 1, context_length=96, device="cpu", lib=lib)
    s1 = pipeline.HipStage(path, 1, 2, context_length=96, device="cpu", lib=lib)
    assert s0.first and not s0.last and s1.last and not s1.first and s0.n_layer == 2
    prompt = [int(t) for t in g["prompt"]]
    x = s0.forward(prompt, 0)
    orc = mirror.MirrorLlama(path, 96)
    assert np.array_equal(x.numpy(), orc.eval_range(prompt, 0, 0, 1))
    logits = s1.forward([0] * len(prompt), 0, x)
    assert np.array_equal(logits.numpy(), g["logits"][0])
    for i in range(3):  # decode steps through the two stages
        t = int(g["greedy"][i])
        logits = s1.forward([0], len(prompt) + i, s0.forward([t], len(prompt) + i))
        assert np.array_equal(logits.numpy(), g["logits"][i + 1])
    # misuse is refused loudly
    with pytest.raises(ValueError):
        s1.forward([0], 0, None)
    # a stage handle refuses the whole-model entry point
    lib.ctransformers_llm_batch_eval.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_int]
    lib.ctransformers_llm_batch_eval.restype = ctypes.c_bool
    assert not lib.ctransformers_llm_batch_eval(s1._h, (ctypes.c_int * 1)(1), 1, 0, 8, 1)


def test_stage_chain_micro_batches_larger_than_the_batch(emu_lib):
    """ctamd_stage_eval_batched: micro-batches of 24 tokens evaluated as reference batches of 8 through two stages equal the
    reference's batch-by-batch result for the 45-token prompt (golden long_chunked)."""
    path = os.path.join(GOLDEN, "tiny-q4km.gguf")
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    lib = ctypes.CDLL(emu_lib)
    s0 = pipeline.HipStage(path, 0, 1, context_length=96, device="cpu", lib=lib)
    s1 = pipeline.HipStage(path, 1, 2, context_length=96, device="cpu", lib=lib)
    prompt = [int(t) for t in g["long_prompt"]]
    pipe = pipeline.Pipeline(_Chain(s0, s1), 0, 1, "cpu")
    logits = pipe.prefill(prompt, 0, micro_batch=24, batch_size=8)
    assert np.array_equal(logits.numpy(), g["long_chunked"])


class _Chain:
    """Two stages behind the one-stage interface Pipeline drives (world 1)."""
    def __init__(self, s0, s1):
        self.s0, self.s1, self.first, self.last, self.n_embd = s0, s1, True, True, s0.n_embd

    def forward(self, tokens, n_past, x_in=None, batch=0):
        return self.s1.forward([0] * len(tokens), n_past, self.s0.forward(tokens, n_past, None, batch), batch)


def test_falcon_stage_chain(emu_lib, mirror):
    """The falcon graph through two stages (40B-style block: two norms, GQA) == the reference goldens."""
    path = os.path.join(GOLDEN, "falcon-tiny-q4km.gguf")
    g = np.load(os.path.join(GOLDEN, "falcon-tiny-q4km.npz"))
    lib = ctypes.CDLL(emu_lib)
    s0 = pipeline.HipStage(path, 0, 1, context_length=96, device="cpu", lib=lib)
    s1 = pipeline.HipStage(path, 1, 2, context_length=96, device="cpu", lib=lib)
    prompt = [int(t) for t in g["prompt"]]
    logits = s1.forward([0] * len(prompt), 0, s0.forward(prompt, 0))
    assert np.array_equal(logits.numpy(), g["logits"][0])
    t = int(g["greedy"][0])
    logits = s1.forward([0], len(prompt), s0.forward([t], len(prompt)))
    assert np.array_equal(logits.numpy(), g["logits"][1])
    assert pipeline.model_dims(path)["arch"] == "falcon"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gloo_pipeline_world2(emu_lib, tmp_path, mirror):
    """Two processes, gloo, 127.0.0.1.  Micro-batched prefill through the pipeline is bit-identical to the REFERENCE's
    single-process result for the same chunking (golden long_chunked = batch 8); greedy decode returns the same ids on
    rank 0 and on the last rank, equal to the oracle's, with bit-identical final logits."""
    path = os.path.join(GOLDEN, "tiny-q4km.gguf")
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "pipeline_worker.py"), path, emu_lib,
                                       str(tmp_path), "8", "3"], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    recs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert recs[0]["layers"] == [0, 1] and recs[1]["layers"] == [1, 2]
    assert np.array_equal(np.load(tmp_path / "prefill_logits.npy"), g["long_chunked"])
    assert recs[0]["tokens"] == recs[1]["tokens"] and len(recs[0]["tokens"]) == 3
    orc = mirror.MirrorLlama(path, 96)
    prompt = [int(t) for t in g["long_prompt"]]
    for s in range(0, len(prompt), 8):
        lg = orc.eval(prompt[s:s + 8], s)
    want, pos = [], len(prompt)
    for _ in range(3):
        t = int(np.argmax(lg))
        want.append(t)
        lg = orc.eval([t], pos)
        pos += 1
    assert recs[1]["tokens"] == want
    assert np.array_equal(np.load(tmp_path / "logits.npy"), lg)


def test_gloo_pipeline_world3_middle_rank(emu_lib, tmp_path, mirror):
    """Three ranks: the middle rank only receives, runs its layers and sends on (no token ids, no sampling).  A 3-layer
    model, one layer per rank; tokens and final logits equal the oracle's."""
    from tools import synth
    path = str(tmp_path / "three.gguf")
    synth.write_llama_gguf(path, "llama-tiny", "Q4_K_M", seed=41, overrides=dict(n_layer=3))
    port = _free_port()
    procs = []
    for rank in range(3):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "pipeline_worker.py"), path, emu_lib,
                                       str(tmp_path), "8", "2"], env=env))
    for p in procs:
        assert p.wait(timeout=900) == 0
    recs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(3)]
    assert [r["layers"] for r in recs] == [[0, 1], [1, 2], [2, 3]]
    assert recs[0]["tokens"] == recs[2]["tokens"] and len(recs[0]["tokens"]) == 2
    orc = mirror.MirrorLlama(path, 96)
    prompt = synth.prompt_tokens(13, 512)
    for s in range(0, len(prompt), 8):
        lg = orc.eval(prompt[s:s + 8], s)
    want, pos = [], len(prompt)
    for _ in range(2):
        t = int(np.argmax(lg))
        want.append(t)
        lg = orc.eval([t], pos)
        pos += 1
    assert recs[2]["tokens"] == want
    assert np.array_equal(np.load(tmp_path / "logits.npy"), lg)


# ---- the in-process pipeline behind ctransformers_llm_create (csrc/pipeline.cc): CT_AMD_DEVICES names the stages ---------------

def _stages(m):
    import ctypes
    L = m._lib
    L.ctamd_n_stages.restype, L.ctamd_n_stages.argtypes = ctypes.c_int, [ctypes.c_void_p]
    L.ctamd_stage_range.restype = ctypes.c_int
    L.ctamd_stage_range.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    n = L.ctamd_n_stages(m._llm)
    out = []
    for s in range(n):
        a, b = ctypes.c_int(), ctypes.c_int()
        if L.ctamd_stage_range(m._llm, s, ctypes.byref(a), ctypes.byref(b)) == 0:
            out.append((a.value, b.value))
    return n, out


@pytest.mark.parametrize("name,devices", [("tiny-q4km", "2"), ("falcon-tiny-q4km", "0,1"), ("tiny-q5km", "2"),
                                          ("tiny-q4km", "0,0")])   # two stages on ONE device: they share a stream (round 5), the hand-off is stream order
def test_inprocess_pipeline_equals_reference(emu_lib, monkeypatch, name, devices):
    """`AutoModelForCausalLM`-style use, nothing but the C ABI: the handle spans two (emulated) devices, one stage each; logits,
    embeddings and greedy tokens equal the reference build's goldens — prompt in reference batches of 8 (micro-batched through the
    stages), then token by token; KV rollback across the stages."""
    from conftest import GOLDEN
    from ctransformers_amd.llm import LLM, Config
    monkeypatch.setenv("CT_EMU_DEVICES", "2")
    monkeypatch.setenv("CT_AMD_DEVICES", devices)
    monkeypatch.setenv("CT_AMD_PP_MB", "4")   # several micro-batches inside the 11-token prompt (reference batches: 8 + 3)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = LLM(os.path.join(GOLDEN, name + ".gguf"), config=Config(context_length=96, batch_size=8, threads=1), lib=emu_lib)
    n, ranges = _stages(m)
    assert n == 2 and ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == 2
    import ctypes
    m._lib.ctamd_handoff.restype, m._lib.ctamd_handoff.argtypes = ctypes.c_char_p, [ctypes.c_void_p]
    assert m._lib.ctamd_handoff(m._llm).decode() == ("stream" if devices == "0,0" else "event")
    m.eval(list(g["prompt"]))
    assert np.array_equal(m.logits.to_numpy(), g["logits"][0])
    assert np.array_equal(m.embeddings.to_numpy(), g["embeddings"][0])
    for i in range(2):
        t = m.sample(top_k=1, repetition_penalty=1.0)
        assert t == int(g["greedy"][i])
        m.eval([t])
        assert np.array_equal(m.logits.to_numpy(), g["logits"][i + 1])
    # roll the context back by 3 tokens and re-evaluate them: same logits (every stage overwrites its own KV rows)
    keep, redo = m._context[:-3], m._context[-3:]
    m._context = list(keep)
    m.eval(redo)
    assert np.array_equal(m.logits.to_numpy(), g["logits"][2])


def test_inprocess_pipeline_long_prompt_batches(emu_lib, monkeypatch):
    """More stages than layers is refused; two stages with micro-batches of 16 reproduce the reference's batch-by-batch result for
    the 45-token prompt (reference batch size 8)."""
    from conftest import GOLDEN
    from ctransformers_amd.llm import LLM, Config
    monkeypatch.setenv("CT_EMU_DEVICES", "3")
    monkeypatch.setenv("CT_AMD_DEVICES", "3")
    with pytest.raises(Exception):
        LLM(os.path.join(GOLDEN, "tiny-q4km.gguf"), config=Config(context_length=96, batch_size=8, threads=1), lib=emu_lib)
    monkeypatch.setenv("CT_AMD_DEVICES", "2")
    monkeypatch.setenv("CT_AMD_PP_MB", "16")
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    m = LLM(os.path.join(GOLDEN, "tiny-q4km.gguf"), config=Config(context_length=96, batch_size=8, threads=1), lib=emu_lib)
    m.eval(list(g["long_prompt"]))
    assert np.array_equal(m.logits.to_numpy(), g["long_chunked"])


def test_gpu_layers_give_the_default_partition(emu_lib, monkeypatch):
    """Without CT_AMD_DEVICES the stages come from `gpu_layers` and the visible GPUs (csrc/pipeline.cc:plan_devices): a GPU takes at
    most gpu_layers layers, the layers beyond go to the next GPU (the reference leaves them on the CPU, llama.cpp:1913-1919) —
    gpu_layers=1 on a 2-layer file and 4 visible devices is two stages; "everything" (1000) and 0 are one GPU; CT_AMD_DEVICES, when
    set, decides.  Logits are the reference build's either way."""
    from conftest import GOLDEN
    from ctransformers_amd.llm import LLM, Config
    monkeypatch.delenv("CT_AMD_DEVICES", raising=False)
    monkeypatch.setenv("CT_EMU_DEVICES", "4")
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    path = os.path.join(GOLDEN, "tiny-q4km.gguf")
    for gpu_layers, want in ((1, 2), (2, 1), (1000, 1), (0, 1)):
        m = LLM(path, config=Config(context_length=96, batch_size=8, threads=1, gpu_layers=gpu_layers), lib=emu_lib)
        n, ranges = _stages(m)
        assert n == want, (gpu_layers, n)
        if want == 2:
            assert ranges == [(0, 1), (1, 2)]
            m.eval(list(g["prompt"]))
            assert np.array_equal(m.logits.to_numpy(), g["logits"][0])
    monkeypatch.setenv("CT_AMD_DEVICES", "0,1")
    m = LLM(path, config=Config(context_length=96, batch_size=8, threads=1, gpu_layers=1000), lib=emu_lib)
    assert _stages(m)[0] == 2
    monkeypatch.delenv("CT_AMD_DEVICES")
    monkeypatch.setenv("CT_EMU_DEVICES", "1")
    m = LLM(path, config=Config(context_length=96, batch_size=8, threads=1, gpu_layers=1), lib=emu_lib)
    assert _stages(m)[0] == 1   # one visible device: one stage whatever gpu_layers says


def test_inprocess_pipeline_without_chunk_kernels(emu_lib, monkeypatch):
    """A multi-stage handle reports that it coalesces the reference's batches (c_api.cc sends the whole request down at once)
    also when the prompt-chunk kernels are off (CT_AMD_PF=0): the token-step attention kernel derives each token's reference batch
    from the cursor, so the 45-token prompt evaluated as one request equals the reference's batch-by-batch result."""
    from conftest import GOLDEN
    from ctransformers_amd.llm import LLM, Config
    monkeypatch.setenv("CT_EMU_DEVICES", "2")
    monkeypatch.setenv("CT_AMD_DEVICES", "2")
    monkeypatch.setenv("CT_AMD_PF", "0")
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    m = LLM(os.path.join(GOLDEN, "tiny-q4km.gguf"), config=Config(context_length=96, batch_size=8, threads=1), lib=emu_lib)
    m.eval(list(g["long_prompt"]))
    assert np.array_equal(m.logits.to_numpy(), g["long_chunked"])
