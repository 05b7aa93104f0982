cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/mm8_check.py llama-2-7b Q4_K_M 128 2 > /dev/null 2>&1   # creates the model file
CT_AMD_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fast -o fast -- python tools/mm8_check.py --worker fast llama-2-7b Q4_K_M 128 2 512 > gpurun_out/prof_fast.log 2>&1
f=$(find gpurun_out/prof_fast -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:25]:
    print("%-100s calls %6s avg_us %9.2f total_ms %8.2f pct %5s" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
