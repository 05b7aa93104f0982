"""Per-site durations of the prompt-chunk launches from a rocprofv3 kernel trace (CT_AMD_GRAPH=0 run of decode_loop.py).
The chunk path launches, per layer: quantize, matvec_pf[m](qkv) (one launch per kernel kind), attention, quantize, matvec_pf[m](wo), quantize,
matvec_pf[m]<GU>(gate_up), quantize, matvec_pf[m](down) — a mat-vec launch belongs to the site of the quantize launch before it.
usage: pf_sites.py <rocprof_out_dir>"""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    agg, k = defaultdict(list), 0
    for r in rows:
        n, dur = r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        z = int(r.get("Grid_Size_Z", r.get("Grid_Size_z", 1)) or 1)
        if "matvec_pf_kernel" in n or "matvec_pfm_kernel" in n or "matmul_pg_kernel" in n:   # belongs to the site of the preceding quantize launch
            kind = "f16mc " + n.split("<")[1].split(",")[0] + " " if "matmul_pg" in n else ("mfma " if "pfm" in n else "dot4 ")
            agg[kind + ("qkv", "wo", "gate_up", "down")[(k - 1) % 4]].append(dur)
        elif "pf_quantize" in n or "pg_quantize" in n:
            agg["pf quantize"].append(dur)
            k += 1
        elif "attn_fused" in n:
            agg["attention (chunk)" if z > 1 else "attention (token)"].append(dur)
        else:
            agg[n[:60]].append(dur)
    print("# %s" % os.path.relpath(f, d))
    print("%-62s %8s %12s %10s %10s %10s" % ("site / kernel", "calls", "total_us", "avg_us", "min_us", "max_us"))
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-62s %8d %12.1f %10.2f %10.2f %10.2f" % (n, len(v), sum(v), sum(v) / len(v), min(v), max(v)))
