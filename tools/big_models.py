"""Decode timing of the two large BASELINE configs on ONE MI355X (they fit: 25 GB / 49 GB of 288 GB HBM).
usage: big_models.py falcon-40b|llama-2-70b"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ctransformers_amd import measure
from tools import synth
from ctransformers_amd.llm import LLM, Config
which = sys.argv[1]
ft = "Q4_K_M" if which.startswith("falcon") else "Q5_K_M"
p = "/tmp/%s_%s.gguf" % (which, ft)
t0 = time.perf_counter()
if not os.path.exists(p):
    (synth.write_falcon_gguf if which.startswith("falcon") else synth.write_llama_gguf)(p, which, ft, seed=99)
t_gen = time.perf_counter() - t0
t0 = time.perf_counter()
m = LLM(p, config=Config(context_length=512, batch_size=64))
t_load = time.perf_counter() - t0
m.eval(synth.prompt_tokens(32, m.vocab_size))
tok = m.sample(top_k=1, repetition_penalty=1.0)
for _ in range(4): m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
t0 = time.perf_counter()
N = 32
for _ in range(N): m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
dt = (time.perf_counter() - t0) / N
wb = synth.weight_bytes_per_token(p)
sites = measure.profile_sites(m._lib, m._llm, 2)
print(json.dumps(dict(model=which, ftype=ft, file_GB=round(os.path.getsize(p) / 1e9, 2), gen_s=round(t_gen, 1), load_s=round(t_load, 1),
                      ms_per_token=round(dt * 1e3, 3), tok_s=round(1 / dt, 1), weight_GB_per_token=round(wb / 1e9, 2),
                      GBps=round(wb / dt / 1e9, 1), sites={s["site"]: round(s["ms"] * 1e3 / s["launches"], 2) for s in sites})))
