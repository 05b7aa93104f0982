#!/bin/bash
# SQ counters of the order-free prompt kernels (kernels_mm8.h): separate rocprofv3 --pmc passes over one 128-token prompt (eager launches), per-kernel averages.
# usage (GPU box): bash tools/pmc_mm8.sh <outdir> [shape] [ftype] [prompt tokens]
O=${1:-gpurun_out/pmc_mm8}; R=$PWD; SHAPE=${2:-llama-2-7b}; FT=${3:-Q4_K_M}; NTOK=${4:-128}
mkdir -p $O
python tools/mm8_check.py $SHAPE $FT $NTOK 1 2304 > $O/check.txt 2>&1
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/p$i -o p -- python $R/tools/mm8_check.py --worker fast $SHAPE $FT $NTOK 0 2304 > $R/$O/p$i.log 2>&1
done
cd $R
python - <<PY > $O/mm8_pmc.txt
import csv, glob
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "")
        if "mm8_" in k or "attn_chunk" in k or "attn_mm" in k:
            agg[k[:60] + " grid " + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("SQ counters of the order-free prompt kernels, $SHAPE $FT, $NTOK-token prompts (rocprofv3 --pmc, separate passes, eager launches); per-dispatch averages")
for k, cs in sorted(agg.items()):
    n = max(len(v) for v in cs.values())
    avg = {c: sum(v) / len(v) for c, v in cs.items()}
    share = ""
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and avg.get("SQ_BUSY_CU_CYCLES"):
        share = "  mfma_busy/(4*busy_cu) = %.3f" % (avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * avg["SQ_BUSY_CU_CYCLES"]))
    print("%s  (%d dispatches)%s" % (k, n, share))
    for c in sorted(avg):
        print("    %-32s %16.1f" % (c, avg[c]))
PY
find $O -name "*.csv" -size +1M -delete
cat $O/check.txt | tail -2; cat $O/mm8_pmc.txt
