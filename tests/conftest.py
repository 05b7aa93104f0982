import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
EMU_LIB = os.environ.get("CTAMD_EMU_LIB") or os.path.join(ROOT, "tests", "emu", "_build", "libctransformers_emu.so")   # (CTAMD_EMU_LIB: a sanitizer build of the same sources)
HIP_LIB = os.path.join(ROOT, "ctransformers_amd", "lib", "libctransformers.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    return os.path.exists("/dev/kfd")


def pytest_sessionstart(session):
    """On a GPU box, bring up torch's HIP runtime before the library's: torch ships its own copy of the runtime, and
    initialising it after another runtime instance of the process has created and destroyed contexts fails with
    "No HIP GPUs are available" (seen when a pipeline test — the only user of torch device tensors — ran after a plain
    C-ABI test with only this one test file collected).  The pipeline launcher itself always starts with torch."""
    if has_gpu():
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:   # noqa: BLE001 — a box without a usable torch runs the C-ABI tests only
            pass


@pytest.fixture(scope="session")
def emu_lib():
    """CPU emulation of the HIP kernels (tests/emu) — kernel-logic checks without a GPU."""
    import fcntl
    with open(os.path.join(ROOT, "tests", "emu", ".build.lock"), "w") as lk:   # xdist workers: one of them builds, the rest wait
        fcntl.flock(lk, fcntl.LOCK_EX)
        subprocess.run(["make"], cwd=os.path.join(ROOT, "tests", "emu"), check=True, stdout=subprocess.DEVNULL)
    return EMU_LIB


@pytest.fixture(scope="session")
def hip_lib():
    if not os.path.isfile(HIP_LIB):
        subprocess.run(["make"], cwd=os.path.join(ROOT, "ctransformers_amd", "csrc"), check=True, stdout=subprocess.DEVNULL)
    return HIP_LIB


@pytest.fixture(scope="session")
def mirror():
    import fcntl
    from oracle import mirror as m
    with open(os.path.join(ROOT, "oracle", ".build.lock"), "w") as lk:   # make rebuilds a stale library; one xdist worker at a time
        fcntl.flock(lk, fcntl.LOCK_EX)
        subprocess.run(["make", "mirror_lib"], cwd=os.path.join(ROOT, "oracle"), check=True, stdout=subprocess.DEVNULL)
    return m


@pytest.fixture(scope="session")
def ref():
    from oracle import ref as r
    if not r.available():
        pytest.skip("reference build oracle/_ref not present (needs /root/reference to build)")
    return r


# The emulator-heavy tests go to the xdist workers first: `--dist load` hands tests out in collection order, and a three-minute
# test that starts last is three minutes of one busy worker and seven idle ones.
_HEAVY_FIRST = ("test_edge_requests_match_reference_build", "test_matvec9_ragged_and_wide_rows", "test_chunk_attention_across_position_tiles",
                "test_mpt_on_emulator_build", "test_reference_package_drives_this_library", "test_reference_hf_transformers_shim",
                "test_wide_rows_on_emulator_build", "test_starcoder_on_emulator_build", "test_wide_k_rows", "test_inprocess_pipeline_equals_reference",
                "test_gloo_pipeline", "test_stage_chain_micro_batches", "test_one_reference_batch_of_140_tokens_on_the_emulator")


def pytest_collection_modifyitems(config, items):
    def rank(item):
        for i, name in enumerate(_HEAVY_FIRST):
            if name in item.nodeid:
                return i
        return len(_HEAVY_FIRST)
    items.sort(key=rank)   # stable: everything else keeps its order
