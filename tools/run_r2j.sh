cd /root/repo
O=gpurun_out/r2j; rm -rf $O; mkdir -p $O
M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
python - <<'PY'
import os
from ctransformers_amd import synth
p = "/tmp/ctamd_llama2_7b_q4km_r2.gguf"
if not os.path.exists(p): synth.write_llama_gguf(p, "llama-2-7b", "Q4_K_M", seed=1234)
PY
for s in gate_up wo; do
CT_AMD_PG_TRACE=$s CT_AMD_GRAPH=0 timeout 300 python tools/decode_loop.py --model $M --prompt 128 --decode 1 2>&1 | grep pg_trace | sed -n 3,4p > $O/trace_$s.txt
cat $O/trace_$s.txt
done
