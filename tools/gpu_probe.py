"""GPU-box probe: parity vs the real reference .so (oracle/_ref) + first timings.  Test/bench tooling, not product."""
import argparse, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ctransformers_amd import synth  # noqa: E402
from ctransformers_amd.llm import LLM, Config  # noqa: E402
from oracle import ref  # noqa: E402


def parity(path, n_vocab, n_prompt, n_decode, ctx, threads, tag, out):
    cfg = dict(context_length=ctx, batch_size=max(8, n_prompt), threads=threads)
    r = ref.open_llm(path, **cfg)
    m = LLM(path, config=Config(**cfg))
    toks = synth.prompt_tokens(n_prompt, n_vocab)
    t0 = time.time(); r.eval(toks); t_ref_prefill = time.time() - t0
    t0 = time.time(); m.eval(toks); t_gpu_prefill = time.time() - t0
    rel, agree, ref_ms, identical = [], 0, [], 0
    for step in range(n_decode + 1):
        a = r.logits.to_numpy(); b = m.logits.to_numpy()
        identical += bool(np.array_equal(a, b))
        rel.append(float(np.abs(a - b).max() / np.abs(a).max()))
        ta, tb = int(a.argmax()), int(b.argmax())
        agree += ta == tb
        if step == n_decode:
            break
        t0 = time.time(); r.eval([ta]); ref_ms.append((time.time() - t0) * 1e3)
        m.eval([ta])  # teacher-forced with the reference's token so one flip does not cascade
    res = dict(tag=tag, n_prompt=n_prompt, n_decode=n_decode, bit_identical_steps=identical, max_rel=max(rel), rel_first=rel[0], greedy_agree=agree,
               greedy_total=n_decode + 1, ref_prefill_s=t_ref_prefill, gpu_prefill_s=t_gpu_prefill,
               ref_decode_ms_median=float(np.median(ref_ms)) if ref_ms else None)
    print(json.dumps(res), flush=True)
    out.append(res)
    del r, m


def timing(path, n_vocab, n_prompt, n_decode, ctx, tag, out, env=None):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k); os.environ[k] = str(v)
    t0 = time.time()
    m = LLM(path, config=Config(context_length=ctx, batch_size=n_prompt))
    t_load = time.time() - t0
    toks = synth.prompt_tokens(n_prompt, n_vocab)
    t0 = time.time(); m.eval(toks); t_prefill = time.time() - t0
    tok = m.sample(top_k=1, repetition_penalty=1.0)
    for _ in range(4):
        m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0)
    ts = []
    t_all = time.time()
    for _ in range(n_decode):
        t0 = time.time(); m.eval([tok]); tok = m.sample(top_k=1, repetition_penalty=1.0); ts.append(time.time() - t0)
    t_all = time.time() - t_all
    res = dict(tag=tag, env=env or {}, load_s=t_load, prefill_s=t_prefill, prefill_tok_s=n_prompt / t_prefill,
               decode_tok_s=n_decode / t_all, decode_ms_median=float(np.median(ts) * 1e3), decode_ms_min=float(np.min(ts) * 1e3))
    print(json.dumps(res), flush=True)
    out.append(res)
    del m
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", default="small,7b2l,7b")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "probe.json"))
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    out = []
    stages = a.stage.split(",")
    print("host cores:", os.cpu_count(), flush=True)
    if "small" in stages:
        for shape, ft in (("llama-tiny", "Q4_K_M"), ("llama-small", "Q4_K_M"), ("llama-small", "Q5_K_M")):
            p = "/tmp/%s-%s.gguf" % (shape, ft)
            hp = synth.write_llama_gguf(p, shape, ft, seed=7)
            for g in (0, 1):
                os.environ["CT_AMD_GRAPH"] = str(g)
                parity(p, hp["n_vocab"], 12, 40, 64, 4, "%s-%s-graph%d" % (shape, ft, g), out)
    os.environ["CT_AMD_GRAPH"] = "1"
    if "7b2l" in stages:
        p = "/tmp/l7b2l.gguf"
        hp = synth.write_llama_gguf(p, "llama-7b-2l", "Q4_K_M", seed=11)
        parity(p, 32000, 32, 16, 512, a.threads, "7b-2l", out)
        timing(p, 32000, 32, 128, 512, "7b-2l-graph", out)
    if "7b" in stages:
        p = "/tmp/l7b.gguf"
        t0 = time.time()
        if not os.path.exists(p):
            synth.write_llama_gguf(p, "llama-2-7b", "Q4_K_M", seed=1234)
        print("7B synth write s:", time.time() - t0, flush=True)
        parity(p, 32000, 32, 32, 512, a.threads, "7b-exact", out)
        timing(p, 32000, 128, 128, 512, "7b-exact-graph", out)
        timing(p, 32000, 16, 64, 512, "7b-exact-eager", out, env=dict(CT_AMD_GRAPH=0))
        for mw in (512, 768, 2048):
            timing(p, 32000, 16, 64, 512, "7b-exact-maxwg%d" % mw, out, env=dict(CT_AMD_MAXWG=mw))
        timing(p, 32000, 16, 64, 512, "7b-designB", out, env=dict(CT_AMD_DESIGN=1))
        timing(p, 32000, 16, 64, 512, "7b-legacy-graph", out, env=dict(CT_AMD_EXACT=0))
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
