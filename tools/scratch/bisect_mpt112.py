import numpy as np, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ctransformers_amd import synth
from ctransformers_amd.llm import LLM, Config
from oracle import ref
def run(name, shape, n, pf):
    os.environ["CT_AMD_PF"] = pf
    p = "/tmp/bis.bin"
    hp = synth.write_mpt_ggml(p, shape, seed=21, ftype=2)
    ctx = n + 16
    r = ref.open_llm(p, model_type="mpt", context_length=ctx, batch_size=64, threads=16)
    m = LLM(p, "mpt", config=Config(context_length=ctx, batch_size=64))
    toks = synth.prompt_tokens(n, hp["n_vocab"])
    r.eval(toks); m.eval(toks)
    a, b = r.logits.to_numpy(), m.logits.to_numpy()
    e1 = np.array_equal(a, b); rel = float(np.abs(a - b).max() / np.abs(a).max())
    t = int(a.argmax()); r.eval([t]); m.eval([t])
    e2 = np.array_equal(r.logits.to_numpy(), m.logits.to_numpy())
    print("%-40s n=%3d PF=%s prompt_equal=%s (rel %.3g) decode_equal=%s" % (name, n, pf, e1, rel, e2), flush=True)
base = dict(n_vocab=512, max_seq_len=2048, n_layer=2, alibi_bias_max=8.0, clip_qkv=0.0)
cases = [
    ("tiny112", dict(base, n_embd=896, n_head=8)),
    ("E3584 h32x112", dict(base, n_embd=3584, n_head=32)),
    ("E7168 h56x128", dict(base, n_embd=7168, n_head=56)),
    ("E7168 h64x112", dict(base, n_embd=7168, n_head=64)),
]
for name, shape in cases:
    for n in (5, 40):
        for pf in ("1", "0"):
            try:
                run(name, shape, n, pf)
            except Exception as e:
                print(name, n, pf, "EXC", repr(e)[:200], flush=True)
