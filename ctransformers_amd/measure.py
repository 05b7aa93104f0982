"""Measurement helpers shared by bench.py and the pipeline bench: per-launch-site HIP-event timings of one decode step
(ctamd_profile_decode, include/ctransformers_amd_ext.h) folded into the `roofline` object of the bench JSON line."""
import ctypes

PMC_FILE = "r06_v9_pmc_traffic.json"   # the committed separate-pass PMC summary `traffic` is read from (profiles/)
HBM_PEAK = 8.0e12  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s is the measured copy ceiling)
MATVEC_SITES = ("qkv", "wo", "gate_up", "lm_head")  # the K=4096 instantiation of the dominant kernel
KERNEL = "matvec_v9_kernel<16384,TA,TB,LN> at K = 4096 (QKV, Wo, gate+up, lm_head launch sites; `down` is the same kernel at K = 11008)"
# a handle whose token steps take the fused QKV + attention launch (kernels_qa9.h) runs the QKV rows inside that launch: the dominant
# kernel's own launches are then Wo, gate+up and the head
MATVEC_SITES_FUSED = ("wo", "gate_up", "lm_head")
KERNEL_FUSED = ("matvec_v9_kernel<16384,TA,0,LN> at K = 4096 (Wo, gate+up, lm_head launch sites; `down` is the same kernel at K = 11008; the QKV rows "
                "stream through the same v9_run inside qkv_attn9_kernel, timed as part of that launch)")


class LaunchStat(ctypes.Structure):
    _fields_ = [("site", ctypes.c_char * 32), ("bytes", ctypes.c_double), ("ms", ctypes.c_double), ("launches", ctypes.c_int)]


def profile_sites(lib, handle, iters=8):
    """NOTE: replays the last evaluated token (eager launches, one event pair per launch): call it AFTER the timed region.
    On a pipeline stage the replay consumes whatever the hand-off buffer holds, so no decoding may follow it."""
    lib.ctamd_profile_decode.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(LaunchStat), ctypes.c_int]
    lib.ctamd_profile_decode.restype = ctypes.c_int
    buf = (LaunchStat * 32)()
    n = lib.ctamd_profile_decode(handle, iters, buf, 32)
    return [dict(site=buf[i].site.decode(), bytes=buf[i].bytes, ms=buf[i].ms, launches=buf[i].launches) for i in range(max(n, 0))]


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC pass (profiles/, collected by a separate
    `rocprofv3 --pmc FETCH_SIZE` run as the guide prescribes: it cannot share a pass with the timing run)."""
    import json
    import os
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", PMC_FILE)
    try:
        return int(json.load(open(p))["dominant_kernel"]["traffic_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def roofline(sites, traffic="pmc", fused=False):
    """Dominant kernel = the K=4096 mat-vec instantiation (qkv / wo / gate_up / lm_head launch sites).  Its launch
    duration comes from the "@sweep" entries when the library provides them: ONE HIP-event pair around the launches of
    all layers of a site, back to back, different weights each (HBM-cold like the real step) — the per-launch cost inside
    a graph replay.  The single-launch event timings (eager, ~6.6 us event floor each) are reported beside it."""
    if traffic == "pmc":
        traffic = pmc_traffic()
    sweep = {s["site"].split("@")[0]: s for s in sites if s["site"].endswith("@sweep")}
    single = [s for s in sites if not s["site"].endswith("@sweep")]
    names = MATVEC_SITES_FUSED if fused else MATVEC_SITES
    src = [sweep[k] for k in names if k in sweep] or [s for s in single if s["site"] in names]
    if not src:
        return None
    b = sum(s["bytes"] for s in src)
    ms = sum(s["ms"] for s in src)
    nl = sum(s["launches"] for s in src)
    ach = b / (ms * 1e-3)

    def per_site(lst):
        return {s["site"].split("@")[0]: dict(GBps=round(s["bytes"] / (s["ms"] * 1e-3) / 1e9, 1) if s["bytes"] else None,
                                               us=round(s["ms"] * 1e3 / s["launches"], 2)) for s in lst}

    return dict(bound="hbm", kernel=KERNEL_FUSED if fused else KERNEL, achieved=round(ach / 1e9, 1), peak=HBM_PEAK / 1e9, unit="GB/s",
                frac=round(ach / HBM_PEAK, 4), traffic=traffic,
                traffic_source="profiles/%s: separate rocprofv3 --pmc FETCH_SIZE pass of the same build (tools/measure_round.sh; x2 gfx950 correction) — a counter pass cannot share a run with the timing" % PMC_FILE if traffic else None,
                bytes_per_launch=round(b / nl),
                us_per_launch=round(ms * 1e3 / nl, 2),
                timing="HIP events on the library stream around the back-to-back launches of all layers of a site" if sweep
                else "HIP events around single eager launches",
                sites=per_site(sweep.values()) if sweep else per_site(single), sites_single_launch=per_site(single))
