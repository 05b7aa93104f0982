# attention launches of a 1920-token prompt at context 2048: duration per chunk (position) from a rocprofv3 kernel trace
O=gpurun_out/r2a; mkdir -p $O
python -c "import bench; bench.ensure_model()" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/prof -o a -- python /root/repo/tools/prefill_2k.py > /root/repo/$O/prof.log 2>&1
cd /root/repo
python - <<'PY'
import csv, glob, collections, statistics
f = glob.glob("gpurun_out/r2a/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
att = [(r["Kernel_Name"][:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "attn_chunk" in r["Kernel_Name"]]
# first prompt pass: 15 chunks x 32 layers
per = collections.defaultdict(list)
for i, (n, d) in enumerate(att[:15 * 32]):
    per[i // 32].append((n, d))
for c in sorted(per):
    print("chunk %2d (positions %4d..%4d): %s median %.1f us" % (c, c * 128, c * 128 + 127, per[c][0][0], statistics.median(d for _, d in per[c])))
tot = collections.defaultdict(float)
for r in rows:
    tot[r["Kernel_Name"][:50]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:8]:
    print("%-52s %10.1f us" % (n, v))
PY
find $O -name "*.csv" -size +1M -delete
