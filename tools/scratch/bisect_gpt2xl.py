import numpy as np, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ctransformers_amd import synth
from ctransformers_amd.llm import LLM, Config
from oracle import ref
def run(tag, shape, seed, ftype, n=20, steps=12, pf="1"):
    os.environ["CT_AMD_PF"] = pf
    p = "/tmp/bis_gpt2.bin"
    synth.write_gpt2_ggml(p, shape, seed=seed, ftype=ftype)
    r = ref.open_llm(p, model_type="gpt2", context_length=64, batch_size=64, threads=8)
    m = LLM(p, "gpt2", config=Config(context_length=64, batch_size=64))
    toks = synth.prompt_tokens(n, shape["n_vocab"])
    r.eval(toks); m.eval(toks)
    first = None
    for i in range(steps):
        a, b = r.logits.to_numpy(), m.logits.to_numpy()
        if not np.array_equal(a, b) and first is None: first = (i, float(np.abs(a - b).max() / np.abs(a).max()))
        t = int(a.argmax()); r.eval([t]); m.eval([t])
    print("%-44s seed %2d ftype %d PF=%s first mismatch: %s" % (tag, seed, ftype, pf, first), flush=True)
xl = dict(n_vocab=50257, n_ctx=1024, n_embd=1600, n_head=25, n_layer=2)
run("xl-2l", xl, 21, 2); run("xl-2l", xl, 21, 2, pf="0"); run("xl-2l", xl, 5, 2); run("xl-2l", xl, 21, 7)
run("E1536 h24", dict(xl, n_embd=1536, n_head=24), 21, 2)
run("E1600 h25 vocab 512", dict(xl, n_vocab=512), 21, 2)
run("E1600 h25 1 layer", dict(xl, n_layer=1), 21, 2)
run("E1664 h26", dict(xl, n_embd=1664, n_head=26), 21, 2)
