#!/bin/bash
# MFMA-busy counters of the prompt-chunk kernels (north_star: "MFMA utilisation for prefill"): separate rocprofv3 --pmc passes over a 128-token
# prompt of the 7B Q4_K_M file (eager launches), summarised per kernel.  usage (GPU box): bash tools/pmc_mfma.sh <outdir>
O=${1:-gpurun_out/pmc_mfma}; R=$PWD; M=${CTAMD_BENCH_MODEL:-/tmp/ctamd_llama2_7b_q4km_r2.gguf}
mkdir -p $O
python - <<PY
import os, sys
sys.path.insert(0, os.getcwd())
from tools import synth
p = "$M"
if not os.path.exists(p): synth.write_llama_gguf(p, "llama-2-7b", "Q4_K_M", seed=1234)
PY
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $R/$O/mfma_counters_available.txt
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/p$i -o p -- python $R/tools/decode_loop.py --model $M --prompt 128 --decode 1 > $R/$O/p$i.log 2>&1
done
cd $R
python - <<PY > $O/prefill_mfma_pmc.txt
import csv, glob
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "")
        if "matmul_pg" in k or "pg_quantize" in k or "attn_chunk" in k or "matvec_pfm" in k:
            agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("MFMA-busy counters of the prompt-chunk kernels, 7B Q4_K_M, one 128-token prompt (rocprofv3 --pmc, separate passes, eager launches)")
print("per-dispatch averages; MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) where both exist")
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
cat $O/mfma_counters_available.txt | tr '\n' ' '; echo; head -40 $O/prefill_mfma_pmc.txt
