"""Order-free prompt kernels (kernels_mm8.h) against the bit-identical chunk kernels on one synthetic model: logits of a prompt, the greedy
continuation, and the steady-state prompt rate of both forms.  One child process per form (CT_AMD_PREFILL is read at load).
usage: python tools/mm8_check.py [shape] [ftype] [n_prompt] [n_greedy] [ctx]      (CT_AMD_MM8_SHAPE / CT_AMD_MM8_SITES pass through)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def model_path(shape, ftype):
    from tools import synth
    try:
        from oracle import ref
        q = "reference" if ref.available() else None
    except Exception:   # noqa: BLE001
        q = None
    p = "/tmp/ctamd_%s_%s_%s.gguf" % (shape.replace("-", "_"), ftype.lower(), "refq" if q else "r2")
    if not os.path.exists(p):
        (synth.write_falcon_gguf if shape.startswith("falcon") else synth.write_llama_gguf)(p + ".tmp", shape, ftype, seed=1234, quantizer=q)
        os.replace(p + ".tmp", p)
    return p


def worker(mode, shape, ftype, n_prompt, n_greedy, ctx):
    import numpy as np
    from tools import synth
    os.environ["CT_AMD_PREFILL"] = mode
    from ctransformers_amd.llm import LLM, Config
    p = model_path(shape, ftype)
    t0 = time.perf_counter()
    m = LLM(p, config=Config(context_length=ctx, batch_size=n_prompt))
    load_s = time.perf_counter() - t0
    toks = synth.prompt_tokens(n_prompt, m.vocab_size)
    m.eval(toks)
    lg = np.array(m.logits.to_numpy(), copy=True)
    seq = []
    for _ in range(n_greedy):
        t = m.sample(top_k=1, repetition_penalty=1.0)
        seq.append(int(t))
        m.eval([t])
    lg_end = np.array(m.logits.to_numpy(), copy=True)
    ts = []
    for _ in range(5):
        m._context = []
        t0 = time.perf_counter()
        m.eval(toks)
        ts.append(time.perf_counter() - t0)
    np.save("/tmp/mm8_check_%s.npy" % mode, np.stack([lg, lg_end]))
    print(json.dumps(dict(mode=mode, load_s=round(load_s, 2), prefill_tok_s=round(n_prompt / min(ts), 1), seq=seq)), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]))
        return
    import numpy as np
    shape = sys.argv[1] if len(sys.argv) > 1 else "llama-2-7b"
    ftype = sys.argv[2] if len(sys.argv) > 2 else "Q4_K_M"
    n_prompt = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    n_greedy = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    ctx = int(sys.argv[5]) if len(sys.argv) > 5 else 512
    model_path(shape, ftype)
    out = {}
    for mode in ("exact", "fast"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", mode, shape, ftype, str(n_prompt), str(n_greedy), str(ctx)],
                           capture_output=True, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            print("worker %s failed:\n%s\n%s" % (mode, r.stdout[-2000:], r.stderr[-4000:]))
            sys.exit(1)
        out[mode] = json.loads(lines[-1])
        if r.stderr.strip():
            print("[%s stderr] %s" % (mode, r.stderr.strip()[-600:]))
    a, b = np.load("/tmp/mm8_check_exact.npy"), np.load("/tmp/mm8_check_fast.npy")
    rel = [float(np.abs(a[i] - b[i]).max() / np.abs(a[i]).max()) for i in range(2)]
    same = out["exact"]["seq"] == out["fast"]["seq"]
    first = next((i for i, (x, y) in enumerate(zip(out["exact"]["seq"], out["fast"]["seq"])) if x != y), None)
    print("%s %s, %d-token prompt: logits rel diff fast vs exact %.3g (prompt), %.3g (after %d greedy steps); greedy continuation identical: %s%s" %
          (shape, ftype, n_prompt, rel[0], rel[1], n_greedy, same, "" if same else " (first difference at step %d)" % first))
    print("prompt tok/s: exact %.0f, fast %.0f (x%.2f); load %.2f s / %.2f s" % (out["exact"]["prefill_tok_s"], out["fast"]["prefill_tok_s"],
          out["fast"]["prefill_tok_s"] / out["exact"]["prefill_tok_s"], out["exact"]["load_s"], out["fast"]["load_s"]))


if __name__ == "__main__":
    main()
