# round 2: load pipeline (staged file + GPU repack): load time, parity suite, bench
cd /root/repo
O=gpurun_out/r2l; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
python - <<'PY' 2>&1 | tee gpurun_out/r2l/load_times.txt
import time, os
from ctransformers_amd.llm import LLM, Config
p = "/tmp/ctamd_llama2_7b_q4km_r2.gguf"
for env in ({}, {"CT_AMD_WARMUP": "0"}, {"CT_AMD_GPU_REPACK": "0"}, {"CT_AMD_TILE8S": "1"}):
    for k in ("CT_AMD_WARMUP", "CT_AMD_GPU_REPACK", "CT_AMD_TILE8S"): os.environ.pop(k, None)
    os.environ.update(env)
    ts = []
    for rep in range(3):
        t0 = time.time(); m = LLM(p, None, config=Config(context_length=512, batch_size=128)); ts.append(time.time() - t0); del m
    print("load 7B Q4_K_M %-28s: %s s" % (env, " ".join("%.2f" % t for t in ts)))
PY
timeout 600 python bench.py --no-cpu-baseline > $O/bench_1.json 2> $O/bench_1.err; tail -2 $O/bench_1.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2l/bench_1.json") if l.startswith("{")][-1])
print("decode", d["value"], "prefill", d["prefill_tok_s"], "load", d["load_s"])
PY
