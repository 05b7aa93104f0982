python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from tools import synth
p = "/tmp/ctamd_llama2_7b_q4km_r2.gguf"
if not os.path.exists(p): synth.write_llama_gguf(p, "llama-2-7b", "Q4_K_M", seed=1234)
PY
for i in 1 2; do
python tools/stamps2.py 2>&1 | grep -v amdgpu.ids
CT_AMD_SPEC=0 python tools/stamps2.py 2>&1 | grep -v amdgpu.ids
CT_AMD_HEAD_FOLD=0 python tools/stamps2.py 2>&1 | grep -v amdgpu.ids
done
