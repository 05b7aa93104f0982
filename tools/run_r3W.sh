cd /root/repo
O=gpurun_out/r3W; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_wide_rows.py tests/test_mpt.py tests/test_starcoder.py -m gpu -q -x -k "Q8_0 or Q4_0 or q80 or q40 or wide or mpt or starcoder or gpt2 or config3 or falcon" -p no:cacheprovider > $O/pytest.log 2>&1; tail -2 $O/pytest.log
python - <<'PY'
import time, sys, os
sys.path.insert(0, ".")
from ctransformers_amd import synth
from ctransformers_amd.llm import LLM, Config
p = "/tmp/falcon7b_q40.gguf"
synth.write_falcon_gguf(p, "falcon-7b", "Q4_0", seed=1)
t0 = time.perf_counter(); m = LLM(p, config=Config(context_length=512, batch_size=128)); tl = time.perf_counter() - t0
toks = synth.prompt_tokens(128, m.vocab_size)
for _ in range(2):
    m._context = []; m.eval(toks)
m._context = []
t0 = time.perf_counter(); m.eval(toks); tp = time.perf_counter() - t0
tok = m.sample(top_k=1, repetition_penalty=1.0)
for _ in range(4): m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
t0 = time.perf_counter()
for _ in range(32): m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
dt = (time.perf_counter() - t0) / 32
print("Falcon-7B Q4_0 (real widths: n_embd 4544, 71 heads on one KV head; %.2f GB): load %.1f s, prefill 128 tok = %.0f tok/s, decode %.1f tok/s" % (os.path.getsize(p) / 1e9, tl, 128 / tp, 1 / dt))
PY
