"""Worker for tests/test_pipeline.py: one rank of a world_size-N gloo pipeline on CPU.  The stage executor is the
PRODUCT's HipStage over the emulator build of the HIP sources (tests/emu), so everything but RCCL itself is exercised."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from tools import rccl_pipeline as pipeline  # noqa: E402


def main():
    model, emu, out_dir, micro, n_new = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    rank, world, _, device = pipeline.init_distributed("gloo")
    dims = pipeline.model_dims(model)
    l0, l1 = pipeline.partition_layers(dims["n_layer"], world)[rank]
    if emu == "hip":  # GPU box, one GPU: both ranks put their stage on cuda:0, gloo hand-off staged through the host
        stage = pipeline.HipStage(model, l0, l1, context_length=96, device="cuda:0")
        pipe = pipeline.Pipeline(stage, rank, world, device, stage_device="cuda:0")
    else:
        stage = pipeline.HipStage(model, l0, l1, context_length=96, device="cpu", lib=ctypes.CDLL(emu))
        pipe = pipeline.Pipeline(stage, rank, world, device)
    npz = os.path.splitext(model)[0] + ".npz"
    if os.path.exists(npz):
        prompt = [int(t) for t in np.load(npz)["long_prompt"]]
    else:   # ad-hoc model: a fixed pseudo-random prompt
        from tools import synth
        prompt = synth.prompt_tokens(13, dims["n_vocab"])
    pre = pipe.prefill(prompt, 0, micro_batch=micro)
    logits, toks, pos = pre, [], len(prompt)
    for _ in range(n_new):
        tok = pipe._return_token(logits)
        toks.append(tok)
        logits = pipe.eval_chunk([tok], pos)
        pos += 1
    rec = dict(rank=rank, layers=[l0, l1], tokens=toks)
    if rank == world - 1:
        np.save(os.path.join(out_dir, "prefill_logits.npy"), pre.numpy())
        np.save(os.path.join(out_dir, "logits.npy"), logits.numpy())
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(rec, f)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
