# round 2: kernels_pg.h — A/B check against the int8 form, bench, per-site trace, SQ counters of a 128-token prompt
cd /root/repo
O=gpurun_out/r2h; rm -rf $O; mkdir -p $O
M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
python - <<'PY'
import os
from ctransformers_amd import synth
p = "/tmp/ctamd_llama2_7b_q4km_r2.gguf"
if not os.path.exists(p): synth.write_llama_gguf(p, "llama-2-7b", "Q4_K_M", seed=1234)
PY
PG_CHECK_REPS=4 timeout 900 python tools/pg_check.py $M 24 33 128 > $O/pg_check.txt 2>&1
cat $O/pg_check.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_1.json 2> $O/bench_1.err; tail -2 $O/bench_1.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2h/bench_1.json") if l.startswith("{")][-1])
print("decode", d["value"], "prefill", d["prefill_tok_s"], "load", d["load_s"])
PY
cd /tmp && export TMPDIR=/tmp
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/prof_prefill -o pf -- python /root/repo/tools/decode_loop.py --model $M --prompt 128 --decode 2 > /root/repo/$O/prof_prefill.log 2>&1
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d /root/repo/$O/pmc1 -o p -- python /root/repo/tools/decode_loop.py --model $M --prompt 128 --decode 1 > /root/repo/$O/pmc1.log 2>&1
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD --output-format csv -d /root/repo/$O/pmc2 -o p -- python /root/repo/tools/decode_loop.py --model $M --prompt 128 --decode 1 > /root/repo/$O/pmc2.log 2>&1
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS --output-format csv -d /root/repo/$O/pmc3 -o p -- python /root/repo/tools/decode_loop.py --model $M --prompt 128 --decode 1 > /root/repo/$O/pmc3.log 2>&1
cd /root/repo
python tools/pf_sites.py $O/prof_prefill > $O/prefill_sites.txt 2>&1
head -14 $O/prefill_sites.txt
for d in pmc1 pmc2 pmc3; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); echo "== $d $f"; [ -n "$f" ] && python tools/pmc_sq.py $f matmul_pg ; done > $O/sq.txt 2>&1
grep -A12 "12, 32, 8, true\|14, 32, 8, false" $O/sq.txt | head -90
tail -2 $O/pmc3.log
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
