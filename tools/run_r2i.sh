cd /root/repo
O=gpurun_out/r2i; rm -rf $O; mkdir -p $O
M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
python - <<'PY'
import os
from ctransformers_amd import synth
p = "/tmp/ctamd_llama2_7b_q4km_r2.gguf"
if not os.path.exists(p): synth.write_llama_gguf(p, "llama-2-7b", "Q4_K_M", seed=1234)
PY
PG_CHECK_REPS=${REPS:-4} timeout 900 python tools/pg_check.py $M 24 33 128 > $O/pg_check.txt 2>&1
cat $O/pg_check.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_1.json 2> $O/bench_1.err; tail -2 $O/bench_1.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2i/bench_1.json") if l.startswith("{")][-1])
print("decode", d["value"], "prefill", d["prefill_tok_s"], "load", d["load_s"])
PY
cd /tmp && export TMPDIR=/tmp
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/prof_prefill -o pf -- python /root/repo/tools/decode_loop.py --model $M --prompt 128 --decode 2 > /root/repo/$O/prof_prefill.log 2>&1
cd /root/repo
python tools/pf_sites.py $O/prof_prefill > $O/prefill_sites.txt 2>&1
head -12 $O/prefill_sites.txt
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
CT_AMD_PG_TRACE=gate_up CT_AMD_GRAPH=0 timeout 300 python tools/decode_loop.py --model $M --prompt 128 --decode 1 2>&1 | grep pg_trace | sed -n 3,3p | cut -c1-900
