"""Token-step timeline from a rocprofv3 --kernel-trace CSV: kernel durations and the GAPS between consecutive kernels of the decode
steps (graph replay), per launch site, and the gap between one token's last kernel and the next token's first.
usage: timeline.py <rocprof_out_dir> [--skip-tokens N]"""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()


def short(n):
    n = n.replace("void ", "").replace("ctamd::", "")
    return n[:70]


# decode steps: find sequences that start after the prompt: take the last 60 % of the trace
start = len(rows) * 2 // 5
seq = rows[start:]
# a token boundary = the head launch (the mat-vec with the largest duration ~ 20 us, EMB instantiation: 'Lb1E' after LN flag) — detect by name
def is_head(n):
    return "matvec_v9_kernel" in n and ("ELb0ELb1E" in n or "ELb1ELb1E" in n or "false, true" in n or "true, true" in n)
heads = [i for i, r in enumerate(seq) if is_head(r[2])]
print("kernels in window: %d, head launches: %d" % (len(seq), len(heads)))
steps = []
for a, b in zip(heads[:-1], heads[1:]):
    steps.append(seq[a + 1:b + 1])   # kernels after head a up to and including head b
steps = [s for s in steps if len(s) == max(len(x) for x in steps)] if steps else []
if not steps:
    sys.exit("no full steps found")
n = len(steps[0])
print("launches per token step: %d; steps analysed: %d" % (n, len(steps)))
dur = defaultdict(list); gap = defaultdict(list)
tot_k = []; tot_g = []; wall = []
for si, s in enumerate(steps):
    k = sum(e - b for b, e, _ in s)
    g = 0
    for i in range(1, len(s)):
        g += s[i][0] - s[i - 1][1]
    tot_k.append(k); tot_g.append(g)
    for i, (b, e, nm) in enumerate(s):
        dur[(i % 5 if 0 < i < n - 1 else i, short(nm))].append(e - b)
        if i > 0:
            gap[(i % 5 if 0 < i < n - 1 else i, short(nm))].append(b - s[i - 1][1])
for a, b in zip(steps[:-1], steps[1:]):
    wall.append(b[-1][1] - a[-1][1])
    tot_g[-1]  # noqa
inter = [b[0][0] - a[-1][1] for a, b in zip(steps[:-1], steps[1:])]
print("per token: kernels %.1f us, gaps inside the step %.1f us, gap head(k) -> first kernel(k+1) %.1f us (min %.1f max %.1f), head-to-head %.1f us" % (
    sum(tot_k) / len(tot_k) / 1e3, sum(tot_g) / len(tot_g) / 1e3, sum(inter) / max(1, len(inter)) / 1e3, min(inter) / 1e3 if inter else 0, max(inter) / 1e3 if inter else 0,
    sum(wall) / max(1, len(wall)) / 1e3))
print("%-4s %-72s %6s %9s %9s" % ("slot", "kernel", "n", "avg_us", "gap_before_us"))
for key in sorted(dur):
    v = dur[key]; g = gap.get(key, [0])
    print("%-4s %-72s %6d %9.2f %9.2f" % (key[0], key[1], len(v), sum(v) / len(v) / 1e3, sum(g) / len(g) / 1e3))
