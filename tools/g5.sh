mkdir -p gpurun_out/g5; R=$PWD; M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from tools import synth
p = "/tmp/ctamd_llama2_7b_q4km_r2.gguf"
if not os.path.exists(p): synth.write_llama_gguf(p, "llama-2-7b", "Q4_K_M", seed=1234)
PY
cd /tmp; export TMPDIR=/tmp
for v in 1 2 0; do
  CT_AMD_HEAD_FOLD=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/g5/tr_$v -o t -- python $R/tools/decode_loop.py --model $M --prompt 128 --decode 40 > $R/gpurun_out/g5/tr_$v.log 2>&1
  python $R/tools/timeline.py $R/gpurun_out/g5/tr_$v > $R/gpurun_out/g5/timeline_$v.txt 2>&1
done
cd $R; find gpurun_out/g5 -name "*.csv" -delete
for v in 1 2 0; do grep -E "^1[56][0-9] |per token" gpurun_out/g5/timeline_$v.txt; done
for v in 1 2 0 1 2 0; do CT_AMD_HEAD_FOLD=$v timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | head -c 120 | cut -c40-120; echo " fold=$v"; done
