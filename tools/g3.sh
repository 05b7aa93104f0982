mkdir -p gpurun_out/g3; R=$PWD; M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from tools import synth
p = "/tmp/ctamd_llama2_7b_q4km_r2.gguf"
if not os.path.exists(p): synth.write_llama_gguf(p, "llama-2-7b", "Q4_K_M", seed=1234)
PY
cd /tmp; export TMPDIR=/tmp
for v in spec nospec nofold; do
  case $v in spec) E="";; nospec) E="CT_AMD_SPEC=0";; nofold) E="CT_AMD_SPEC=0 CT_AMD_HEAD_FOLD=0";; esac
  env $E timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/g3/tr_$v -o t -- python $R/tools/decode_loop.py --model $M --prompt 128 --decode 40 > $R/gpurun_out/g3/tr_$v.log 2>&1
  echo "rc=$?" >> $R/gpurun_out/g3/tr_$v.log
  python $R/tools/timeline.py $R/gpurun_out/g3/tr_$v > $R/gpurun_out/g3/timeline_$v.txt 2>&1
done
cd $R; find gpurun_out/g3 -name "*.csv" -size +3M -delete; find gpurun_out/g3 -name "*.db" -delete
for v in spec nospec nofold; do echo "== $v"; tail -2 gpurun_out/g3/tr_$v.log; head -12 gpurun_out/g3/timeline_$v.txt; done
