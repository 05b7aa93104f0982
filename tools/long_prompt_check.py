"""Full 7B file, long prompt: the library (batches coalesced into 128-token chunks) against the reference CPU build run batch
by batch, logits of the last prompt token and of a few greedy steps.  usage (GPU box): long_prompt_check.py <model.gguf> [n] [bs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from tools import synth  # noqa: E402
from ctransformers_amd.llm import LLM, Config  # noqa: E402
from oracle import ref  # noqa: E402

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
if not os.path.exists(path):
    synth.write_llama_gguf(path, "llama-2-7b", "Q4_K_M", seed=1234)
m = LLM(path, config=Config(context_length=2048, batch_size=bs, gpu_layers=1000))
toks = synth.prompt_tokens(n, m.vocab_size)
t0 = time.perf_counter(); m.eval(toks); t_gpu = time.perf_counter() - t0
r = ref.open_llm(path, context_length=2048, batch_size=bs, threads=min(32, os.cpu_count() or 1))
t0 = time.perf_counter(); r.eval(toks); t_cpu = time.perf_counter() - t0
ok = np.array_equal(r.logits.to_numpy(), m.logits.to_numpy())
for _ in range(3):
    t = int(r.logits.to_numpy().argmax())
    r.eval([t]); m.eval([t])
    ok = ok and np.array_equal(r.logits.to_numpy(), m.logits.to_numpy())
print("prompt %d tokens, batch_size %d: bit-identical %s; library %.1f tok/s (first call), reference %.1f tok/s" % (n, bs, ok, n / t_gpu, n / t_cpu))
sys.exit(0 if ok else 1)
