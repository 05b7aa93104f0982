"""Measurement helpers shared by bench.py and the pipeline bench: per-launch-site HIP-event timings of one decode step
(ctamd_profile_decode, include/ctransformers_amd_ext.h) folded into the `roofline` object of the bench JSON line."""
import ctypes

HBM_PEAK = 8.0e12  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s is the measured copy ceiling)
MATVEC_SITES = ("qkv", "wo", "gate_up", "lm_head")  # the K=4096 instantiation of the dominant kernel
KERNEL = "matvec_v5_kernel<4096,1,T,2,TA,TB> (QKV, Wo, gate+up, lm_head launch sites; `down` is the <12288,3,2,1> instantiation)"


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


def roofline(sites, traffic=None):
    dom = [s for s in sites if s["site"] in MATVEC_SITES]
    if not dom:
        return None
    b = sum(s["bytes"] for s in dom)
    ms = sum(s["ms"] for s in dom)
    nl = sum(s["launches"] for s in dom)
    ach = b / (ms * 1e-3)
    allw = [s for s in sites if s["bytes"]]
    return dict(bound="hbm", kernel=KERNEL, achieved=round(ach / 1e9, 1), peak=HBM_PEAK / 1e9, unit="GB/s",
                frac=round(ach / HBM_PEAK, 4), traffic=traffic, bytes_per_launch=round(b / nl),
                us_per_launch=round(ms * 1e3 / nl, 2),
                all_weight_sites_GBps=round(sum(s["bytes"] for s in allw) / (sum(s["ms"] for s in allw) * 1e-3) / 1e9, 1),
                sites={s["site"]: dict(GBps=round(s["bytes"] / (s["ms"] * 1e-3) / 1e9, 1) if s["bytes"] else None,
                                       us=round(s["ms"] * 1e3 / s["launches"], 2)) for s in sites})
