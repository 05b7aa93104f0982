# round 3: SQ counters of the generation-9 decode kernels (separate rocprofv3 --pmc passes, kernel-trace only) + the counter list
cd /root/repo
O=gpurun_out/r3p; rm -rf $O; mkdir -p $O
M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
python -c "
import os,sys
sys.path.insert(0,'/root/repo')
from ctransformers_amd import synth
p='$M'
if not os.path.exists(p): synth.write_llama_gguf(p, 'llama-2-7b', 'Q4_K_M', seed=1234)
"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > /root/repo/$O/counters.txt 2>&1
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d /root/repo/$O/pmc1 -o p -- python /root/repo/tools/decode_loop.py --model $M --prompt 8 --decode 6 > /root/repo/$O/pmc1.log 2>&1
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d /root/repo/$O/pmc2 -o p -- python /root/repo/tools/decode_loop.py --model $M --prompt 8 --decode 6 > /root/repo/$O/pmc2.log 2>&1
cd /root/repo
for d in pmc1 pmc2; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); echo "== $d $f"; python tools/pmc_sq.py $f matvec_v9 ; done 2>&1 | tee $O/sq.txt | head -70
grep -c . $O/counters.txt; grep -i -E "^.*(TA_|TCP_|TCC_).*" $O/counters.txt | head -5
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
