cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python - <<'PY'
import os, sys, time, json
sys.path.insert(0, "/root/repo")
from ctransformers_amd import synth, measure
from ctransformers_amd.llm import LLM, Config
for ft in ("Q8_0", "Q4_0"):
    p = "/tmp/l7b_%s.gguf" % ft
    if not os.path.exists(p): synth.write_llama_gguf(p, "llama-2-7b", ft, seed=1234)
    m = LLM(p, config=Config(context_length=512, batch_size=64))
    m.eval(synth.prompt_tokens(64, 32000))
    tok = m.sample(top_k=1, repetition_penalty=1.0)
    for _ in range(8): m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
    t0 = time.perf_counter()
    for _ in range(64): m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
    dt = (time.perf_counter() - t0) / 64
    wb = synth.weight_bytes_per_token(p)
    sites = measure.profile_sites(m._lib, m._llm, 4)
    print(json.dumps(dict(ftype=ft, ms_per_token=round(dt * 1e3, 3), tok_s=round(1 / dt, 1), GBps=round(wb / dt / 1e9, 1),
                          sites={s["site"]: round(s["ms"] * 1e3 / s["launches"], 2) for s in sites})))
    del m
PY
