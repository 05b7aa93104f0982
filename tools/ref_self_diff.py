"""How far is the REFERENCE from itself when only its evaluation order changes?  The reference CPU build (oracle/_ref) evaluates one prompt with two
batch sizes: the V*P dots of a token run over the columns of its batch (llama.cpp:2352-2378), so the f32 sums differ in their last bits — the same class of
change as another summation order in a mat-mul.  Prints max|dlogits| / max|logits| and the first greedy step at which the continuations part.
(Test infrastructure: needs /root/reference's build, oracle/_ref.)  usage: python tools/ref_self_diff.py [shape] [ftype] [n_prompt] [n_greedy]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref          # noqa: E402
from tools import synth         # noqa: E402


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "llama-2-7b"
    ftype = sys.argv[2] if len(sys.argv) > 2 else "Q4_K_M"
    n_prompt = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    n_greedy = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    path = "/tmp/ctamd_%s_%s_refq.gguf" % (shape.replace("-", "_"), ftype.lower())
    if not os.path.exists(path):
        synth.write_llama_gguf(path + ".tmp", shape, ftype, seed=1234, quantizer="reference")
        os.replace(path + ".tmp", path)
    toks = synth.prompt_tokens(n_prompt, synth.LLAMA_SHAPES[shape]["n_vocab"])
    out = {}
    for bs in (n_prompt, 8):
        m = ref.open_llm(path, context_length=n_prompt + n_greedy + 8, batch_size=bs, threads=8)
        m.eval(toks)
        lg = np.array(m.logits.to_numpy(), copy=True)
        seq = []
        for _ in range(n_greedy):
            t = m.sample(top_k=1, repetition_penalty=1.0)
            seq.append(int(t))
            m.eval([t])
        out[bs] = (lg, seq)
        del m
    (a, sa), (b, sb) = out[n_prompt], out[8]
    rel = float(np.abs(a - b).max() / np.abs(a).max())
    first = next((i for i, (x, y) in enumerate(zip(sa, sb)) if x != y), None)
    print("%s %s, %d-token prompt, reference build against itself (batch_size %d vs 8): logits differ by %.3g of the largest; greedy continuations %s" %
          (shape, ftype, n_prompt, n_prompt, rel, "identical over %d steps" % n_greedy if first is None else "part at step %d" % first))


if __name__ == "__main__":
    main()
