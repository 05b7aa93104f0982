"""An eval never fails because the GPU is shared (reference: models/llm.h:40-54 has no such failure; ctransformers/llm.py:404-412 raises on `false`).

The fused QKV + attention launch and the shared score rows of long contexts wait for other workgroups of their own grid; when those are not resident
(another handle, process or a profiler holds waves) a sweep gives up.  Round 6: the stage then switches to the forms that need no residency and the
request is evaluated AGAIN inside the same call (KV overwrite semantics make the replay safe) — results stay the reference's bits, `eval` stays true.
"""
import ctypes
import os
import threading

import numpy as np
import pytest

from conftest import GOLDEN
from tools import synth
from ctransformers_amd.llm import LLM, Config


def _replays(m):
    f = m._lib.ctamd_resident_replays
    f.restype, f.argtypes = ctypes.c_longlong, [ctypes.c_void_p]
    return int(f(m._llm))


def _greedy(m, g, steps, check_logits=True):
    m.eval(list(g["prompt"]))
    if check_logits:
        assert np.array_equal(m.logits.to_numpy(), g["logits"][0])
    for i in range(steps):
        t = m.sample(top_k=1, repetition_penalty=1.0)
        assert t == int(g["greedy"][i]), "step %d" % i
        m.eval([t])
        if check_logits:
            assert np.array_equal(m.logits.to_numpy(), g["logits"][i + 1]), "step %d" % i


@pytest.mark.parametrize("nth", [2, 5])
def test_forced_give_up_is_replayed_on_the_emulator(emu_lib, nth, monkeypatch):
    """CT_AMD_DBG_QA_TIMEOUT=N: the N-th wait of the handle behaves as if a sweep had given up (2: the first token step, taken as a plain eval; 5: a step of
    the greedy chain).  The eval succeeds, the logits are the golden ones, the handle has replayed exactly one request and continues on the two-launch form."""
    monkeypatch.setenv("CT_AMD_DBG_QA_TIMEOUT", str(nth))
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    m = LLM(os.path.join(GOLDEN, "tiny-q4km.gguf"), config=Config(context_length=96, batch_size=8, threads=1), lib=emu_lib)
    _greedy(m, g, 8)
    assert _replays(m) == 1


# ---- MI355X ---------------------------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("name,nth", [("tiny-q4km", 2), ("tiny-q4km", 3), ("tiny-q4km", 6), ("falcon-tiny-q4km", 3), ("falcon-tiny7-q4km", 6)])
def test_forced_give_up_is_replayed(name, nth, monkeypatch):
    """The same on the HIP build: graph replays, queued continuation steps of a greedy chain (the give-up is noticed on a continuation step for nth = 6);
    the llama graph and the falcon graph (round 6: its token steps take the fused launch too, and fall back to the five-launch form)."""
    monkeypatch.setenv("CT_AMD_DBG_QA_TIMEOUT", str(nth))
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = LLM(os.path.join(GOLDEN, name + ".gguf"), config=Config(context_length=96, batch_size=8))
    _greedy(m, g, min(20, len(g["greedy"]) - 1))
    assert _replays(m) == 1


@pytest.mark.gpu
def test_two_handles_take_turns():
    """Two handles on one GPU decoding alternately from one thread: every eval succeeds, both produce the golden logits."""
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    path = os.path.join(GOLDEN, "tiny-q4km.gguf")
    a, b = (LLM(path, config=Config(context_length=96, batch_size=8)) for _ in range(2))
    for m in (a, b):
        m.eval(list(g["prompt"]))
        assert np.array_equal(m.logits.to_numpy(), g["logits"][0])
    for i in range(20):
        for m in (a, b):
            t = m.sample(top_k=1, repetition_penalty=1.0)
            assert t == int(g["greedy"][i])
            m.eval([t])
            assert np.array_equal(m.logits.to_numpy(), g["logits"][i + 1])


def _model_7b_2l():
    try:
        from oracle import ref
        q = "reference" if ref.available() else None
    except Exception:   # noqa: BLE001
        q = None
    path = "/tmp/ctamd_shared_7b2l_%s.gguf" % ("refq" if q else "r2")
    if not os.path.exists(path):
        synth.write_llama_gguf(path + ".tmp", "llama-7b-2l", "Q4_K_M", seed=5, quantizer=q)
        os.replace(path + ".tmp", path)
    return path


@pytest.mark.gpu
@pytest.mark.timeout(180)
def test_two_threads_decode_on_one_gpu():
    """Two host threads, a handle each, 7B widths (the fused launch's grid = every CU): both greedy chains equal the chain a handle decodes alone.  Whether a
    sweep gives up depends on how the hardware interleaves the two grids; if one does, the request is replayed — never a failed eval."""
    path = _model_7b_2l()
    toks = synth.prompt_tokens(40, 32000)

    def chain(out, idx, n):
        try:
            # (created concurrently: a load copies on the legacy stream, which must not meet a graph capture of the other thread — the library
            # serialises creation, deletion and captures process-wide, engine.h:capture_mutex)
            m = LLM(path, config=Config(context_length=256, batch_size=64))
            m.eval(toks)
            seq, lg = [], []
            for _ in range(n):
                t = m.sample(top_k=1, repetition_penalty=1.0)
                seq.append(int(t))
                m.eval([t])
                lg.append(np.array(m.logits.to_numpy(), copy=True))
            out[idx] = (seq, lg, _replays(m))
        except Exception as e:   # noqa: BLE001
            out[idx] = e

    alone = {}
    chain(alone, 0, 48)
    assert not isinstance(alone[0], Exception), alone[0]
    both = {}
    th = [threading.Thread(target=chain, args=(both, i, 48)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(2):
        assert not isinstance(both[i], Exception), both[i]
        assert both[i][0] == alone[0][0]
        assert all(np.array_equal(x, y) for x, y in zip(both[i][1], alone[0][1]))
    print("replays: alone %d, two threads %d + %d" % (alone[0][2], both[0][2], both[1][2]))


@pytest.mark.gpu
@pytest.mark.timeout(180)
def test_decode_beside_a_long_prompt():
    """One handle decodes while another thread's handle evaluates a 2k-token prompt (every CU busy with chunk kernels): the decoded chain is the lone chain's."""
    path = _model_7b_2l()
    toks = synth.prompt_tokens(40, 32000)
    m = LLM(path, config=Config(context_length=256, batch_size=64))
    m.eval(toks)
    want = []
    for _ in range(64):
        t = m.sample(top_k=1, repetition_penalty=1.0)
        want.append(int(t))
        m.eval([t])
    del m
    stop = threading.Event()
    err = []

    def prompts():
        try:
            h = LLM(path, config=Config(context_length=2304, batch_size=128))
            p = synth.prompt_tokens(2048, 32000)
            while not stop.is_set():
                h._context = []
                h.eval(p)
        except Exception as e:   # noqa: BLE001
            err.append(e)

    th = threading.Thread(target=prompts)
    th.start()
    try:
        m = LLM(path, config=Config(context_length=256, batch_size=64))
        m.eval(toks)
        got = []
        for _ in range(64):
            t = m.sample(top_k=1, repetition_penalty=1.0)
            got.append(int(t))
            m.eval([t])
        n_replays = _replays(m)
    finally:
        stop.set()
        th.join()
    assert not err, err
    assert got == want
    print("replays beside the prompt thread: %d" % n_replays)


@pytest.mark.gpu
def test_failed_handoff_self_check_falls_back_to_events(monkeypatch):
    """Two stages with a stream each on the one GPU, hand-off form "flag" requested, the load-time self-check told to report a mismatch: the pipeline hands
    over by copy + event and evaluates the golden logits."""
    import ctypes as C
    monkeypatch.setenv("CT_AMD_DEVICES", "0,0")
    monkeypatch.setenv("CT_AMD_PP_SHARED_STREAM", "0")
    monkeypatch.setenv("CT_AMD_HANDOFF", "flag")
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    path = os.path.join(GOLDEN, "tiny-q4km.gguf")
    for fail, want in (("0", b"flag"), ("1", b"event")):
        monkeypatch.setenv("CT_AMD_DBG_HANDOFF_FAIL", fail)
        m = LLM(path, config=Config(context_length=96, batch_size=8, gpu_layers=1))
        f = m._lib.ctamd_handoff
        f.restype, f.argtypes = C.c_char_p, [C.c_void_p]
        assert f(m._llm) == want, f(m._llm)
        _greedy(m, g, 6)


@pytest.mark.gpu
@pytest.mark.timeout(240)
def test_handles_created_used_and_deleted_from_four_threads():
    """Four host threads, each three times: create a handle, evaluate the golden prompt, decode eight greedy steps with every logits vector fetched, delete it —
    while the other threads are anywhere in the same cycle (loads copy on the legacy stream, first steps capture graphs, deletes synchronize the device:
    the library serialises those three process-wide, engine.h:capture_mutex).  Every chain is the golden chain, bit for bit."""
    g = np.load(os.path.join(GOLDEN, "tiny-q4km.npz"))
    path = os.path.join(GOLDEN, "tiny-q4km.gguf")
    errs = []

    def worker():
        try:
            for _ in range(3):
                m = LLM(path, config=Config(context_length=96, batch_size=8))
                _greedy(m, g, 8)
                del m
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=worker) for _ in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
