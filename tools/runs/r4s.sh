#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4s; mkdir -p $O
cat > /tmp/pf3.py <<'PY'
import os, sys, time, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tools import synth
from ctransformers_amd.llm import LLM, Config
p = "/tmp/ctamd_llama2_7b_q80_r2.gguf"
if not os.path.exists(p): synth.write_llama_gguf(p, "llama-2-7b", "Q8_0", seed=1234)
m = LLM(p, config=Config(context_length=512, batch_size=128), lib=os.environ.get("SITES_LIB") or None)
toks = synth.prompt_tokens(128, 32000)
for _ in range(3):
    m._context = []; m.eval(toks)
ts = []
for _ in range(3):
    m._context = []; t0 = time.perf_counter(); m.eval(toks); ts.append(time.perf_counter() - t0)
print(sys.argv[1], "prefill tok/s %.0f" % (128 / min(ts)))
PY
for v in pfexp1 pfexp2 pfexp3; do
( SITES_LIB=$PWD/ctransformers_amd/lib_$v/libctransformers.so timeout 300 python /tmp/pf3.py $v 2>&1 | tail -1 ) >> $O/pf.txt
done
( timeout 300 python /tmp/pf3.py base 2>&1 | tail -1 ) >> $O/pf.txt
cat $O/pf.txt
