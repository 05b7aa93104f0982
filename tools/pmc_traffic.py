"""Per-kernel HBM traffic from a rocprofv3 --pmc FETCH_SIZE (and optionally WRITE_SIZE) pass -> JSON.
FETCH_SIZE is reported in KB and, on gfx950 for wide coalesced streaming reads, at exactly half the bytes
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section): values are doubled here.  WRITE_SIZE is left uncorrected.
usage: pmc_traffic.py <fetch_counter_collection.csv> [<write_counter_collection.csv>]"""
import csv
import json
import sys
from collections import defaultdict

DOMINANT = ("matvec_v9_kernel<16384, 12, 0, false, false",)  # the Q4_K decode mat-vec: wo / gate_up / down(Q4_K) launch sites (round 5: the QKV rows run inside qkv_attn9_kernel)


def load(path):
    agg = defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].replace("void ", "")].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch = load(sys.argv[1])
    write = load(sys.argv[2]) if len(sys.argv) > 2 else {}
    out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (own pass) / WRITE_SIZE (own pass); FETCH_SIZE KB x1024 x2 "
                   "(gfx950 correction for 16 B/lane streaming reads); per-dispatch averages",
           "kernels": {}}
    tot_b = tot_n = 0.0
    for k, v in sorted(fetch.items()):
        fb = sum(v) / len(v) * 1024.0 * 2.0
        ent = {"dispatches": len(v), "fetch_bytes_per_dispatch": round(fb)}
        if k in write:
            w = write[k]
            ent["write_bytes_per_dispatch_uncorrected"] = round(sum(w) / len(w) * 1024.0)
        out["kernels"][k] = ent
        if k.startswith(DOMINANT):
            tot_b += fb * len(v)
            tot_n += len(v)
    if tot_n:
        out["dominant_kernel"] = {"match": "|".join(DOMINANT), "dispatches": int(tot_n), "traffic_bytes_per_launch": round(tot_b / tot_n)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
