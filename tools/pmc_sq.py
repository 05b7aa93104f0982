"""Per-kernel averages of SQ counters from a rocprofv3 --pmc pass (counter_collection.csv) -> table.
usage: pmc_sq.py <counter_collection.csv> [kernel-substring]"""
import csv
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void ", "")
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    n = max(len(v) for v in cs.values())
    print("%s  (%d dispatches)" % (k[:110], n))
    for c, v in sorted(cs.items()):
        print("    %-28s %14.1f" % (c, sum(v) / len(v)))
