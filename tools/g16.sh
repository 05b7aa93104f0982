mkdir -p gpurun_out/g16; R=$PWD; M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from tools import synth
p = "/tmp/ctamd_llama2_7b_q4km_r2.gguf"
if not os.path.exists(p): synth.write_llama_gguf(p, "llama-2-7b", "Q4_K_M", seed=1234)
PY
cd /tmp; export TMPDIR=/tmp
for v in fused plain; do
  if [ $v = fused ]; then E="CT_AMD_QA_PHASE1=1"; F="qkv_attn9_kernel<12, 14"; else E="CT_AMD_FUSE_QA=0"; F="matvec_v9_kernel<16384, 12, 14"; fi
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_IFETCH"; do
    tag=$(echo $set | cut -c4-12)
    env $E CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/g16/${v}_$tag -o p -- python $R/tools/decode_loop.py --model $M --prompt 8 --decode 6 > $R/gpurun_out/g16/${v}_$tag.log 2>&1
    f=$(find $R/gpurun_out/g16/${v}_$tag -name "*counter_collection.csv" | head -1)
    echo "== $v $tag"; python $R/tools/pmc_sq.py $f "$F" 2>&1 | head -40
  done
done
cd $R; find gpurun_out/g16 -name "*.csv" -size +1M -delete
B="timeout 600 python bench.py --no-cpu-baseline --no-other-configs"
for i in 1 2; do
$B 2>/dev/null | head -c 120 | cut -c40-120; echo " fused"
CT_AMD_QA_PHASE1=1 $B 2>/dev/null | head -c 120 | cut -c40-120; echo " fused kernel, phase 1 + attention launch"
CT_AMD_FUSE_QA=0 $B 2>/dev/null | head -c 120 | cut -c40-120; echo " plain"
done
