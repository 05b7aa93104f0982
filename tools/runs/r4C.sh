#!/bin/bash
# instruction-cache counters of the decode loop's kernels (eager launches in the step's order), one --pmc pass, no other trace domains
R="$GRAFT_REPO_ROOT"; cd /tmp; export TMPDIR=/tmp
O=gpurun_out/r4C; mkdir -p $R/$O
M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
CT_AMD_GRAPH=0 timeout 400 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/$O/pmc_ic -o p -- python $R/tools/decode_loop.py --model $M --shape llama-2-7b --prompt 8 --decode 6 > $R/$O/pmc_ic.log 2>&1
cd $R
f=$(find $O/pmc_ic -name "*counter_collection.csv" | head -1); python tools/pmc_sq.py $f > $O/icache_counters.txt 2>&1
rm -rf $O/pmc_ic
cat $O/icache_counters.txt | head -150; tail -3 $O/pmc_ic.log
