python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from tools import synth
p = "/tmp/ctamd_llama2_7b_q4km_r2.gguf"
if not os.path.exists(p): synth.write_llama_gguf(p, "llama-2-7b", "Q4_K_M", seed=1234)
PY
timeout 900 python -m pytest tests/test_greedy_chain.py -m gpu -x -q 2>&1 | tail -3
python tools/stamps2.py 2>&1 | grep -v amdgpu.ids | cut -c1-330
for v in 1 0 1 0 1 0; do CT_AMD_HEAD_FOLD=$v timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | head -c 120 | cut -c40-120; echo " fold=$v"; done
