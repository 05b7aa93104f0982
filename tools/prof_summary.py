"""Summarise rocprofv3 output directories into the text files committed under profiles/.
usage: prof_summary.py <rocprof_out_dir> [--pmc]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    return name if len(name) <= 100 else name[:97] + "..."


def kernel_stats(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    for f in files:
        rows = list(csv.DictReader(open(f)))
        print("# %s" % os.path.relpath(f, d))
        print("%-100s %8s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
        for r in rows:
            print("%-100s %8s %12.1f %10.2f %10.2f %10.2f %7s" % (
                short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    if not files:  # fall back to the raw trace
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            agg = defaultdict(list)
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            print("# %s (aggregated)" % os.path.relpath(f, d))
            tot = sum(sum(v) for v in agg.values())
            for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                print("%-100s %8d %12.1f %10.2f %10.2f %10.2f %6.2f%%" % (short(k), len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3,
                                                                       min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / tot))


def pmc(d):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("# %s  (per-dispatch averages)" % os.path.relpath(f, d))
        for k, cs in sorted(agg.items()):
            print("%-100s " % short(k) + " ".join("%s: n=%d avg=%.4g" % (c, len(v), sum(v) / len(v)) for c, v in sorted(cs.items())))


if __name__ == "__main__":
    d = sys.argv[1]
    kernel_stats(d)
    if "--pmc" in sys.argv:
        pmc(d)
