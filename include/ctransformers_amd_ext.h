/* Measurement extensions of the MI355X build — NOT part of the reference ABI, never needed by the drop-in path.
 * bench.py uses them to time each launch site of one decode step with HIP events on the library's own stream
 * (torch.cuda.Event would only see torch's stream) and to read the resident weight byte count. */
#ifndef CTRANSFORMERS_AMD_EXT_H_
#define CTRANSFORMERS_AMD_EXT_H_
#include "ctransformers_llm.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct {
    char site[32];   /* "qkv", "wo", "gate_up", "down", "lm_head", "attn_scores", "attn_softmax_pv", "embed" */
    double bytes;    /* algorithmic weight bytes streamed by the launches of this site (summed over launches) */
    double ms;       /* HIP-event time summed over launches */
    int launches;
} ctamd_launch_stat;
/* Replays the last evaluated token `iters` times (eager launches, events around every launch); returns the number of
 * sites written to `out`, or -1. */
int ctamd_profile_decode(ctransformers_llm* llm, int iters, ctamd_launch_stat* out, int max_out);
double ctamd_weight_bytes(ctransformers_llm* llm);
/* Device microseconds per decode step when `n` steps (2..100) are queued back to back with no host round trip between them (HIP events
 * around the burst; continues at the position behind the last eval, token ids arbitrary): the gap to a measured eval + sample loop is
 * what the host adds per token.  -1.0 on failure. */
double ctamd_decode_burst(ctransformers_llm* llm, int n);
/* Greedy chains: evals that were served by a token step the engine had queued ahead (the caller fed back the device-side greedy pick
 * at the next position); *launched = steps queued ahead so far (launched - hits = guesses nobody asked for). */
long long ctamd_spec_hits(ctransformers_llm* llm, long long* launched);
/* measurement only (CT_AMD_STAMPS=1 at load): the (100 MHz wall clock << 4 | tag) stamps taken at the start (tag 1) and end (tag 2) of every
 * token step since the last call; returns how many were copied. */
int ctamd_read_stamps(ctransformers_llm* llm, unsigned long long* out, int max);
int ctamd_read_stamps_stage(ctransformers_llm* llm, int stage, unsigned long long* out, int max);   /* the same for one stage of a pipeline */
/* In-kernel s_memtime stamps of workgroup 0 of the last launch of `site` (16 waves x 16 slots of uint64; slots: 0 entry,
 * 1 first loads issued, 2 prologue done, 3 round-0 block math done, 4 barrier passed, 5 chain+epilogue done, 6 exit). */
int ctamd_trace_site(ctransformers_llm* llm, const char* site, unsigned long long* out, int n);

/* ---- multi-GPU layer pipeline (DESIGN.md row e): one process per GPU, each owning a contiguous layer range ------------
 * The reference has no multi-GPU path in this ABI (its cuBLAS offload splits tensors inside one process,
 * models/ggml/llama.cpp:1938-2070 `tensor_split`); the MI355X design shards LAYERS across ranks instead and hands the
 * [n_tokens][n_embd] f32 residual stream from stage to stage (RCCL send/recv, tools/rccl_pipeline.py).
 * A stage handle is a normal ctransformers_llm*: on the last stage logits_data / sample / embeddings work as usual. */
ctransformers_llm* ctamd_stage_create(const char* model_path, int context_length, int layer_begin, int layer_end,
                                      int device);
/* tokens: host ids (used by the first stage only, may be NULL elsewhere); x_in_dev / x_out_dev: DEVICE pointers to
 * [n_tokens][n_embd] f32 rows (x_in required iff layer_begin > 0, x_out iff layer_end < n_layer).  Returns 0 / -1.
 * The call returns after the stage's stream has drained, so x_out may be handed to a collective on any stream. */
int ctamd_stage_eval(ctransformers_llm* llm, const int* tokens, int n_tokens, int n_past, const void* x_in_dev,
                     void* x_out_dev);
/* ctamd_stage_eval with the reference batch size inside the n_tokens (0: they are one batch).  Lets a pipeline use micro-batches
   larger than the batch size whose results it has to reproduce: the attention kernels derive each token's batch from it. */
int ctamd_stage_eval_batched(ctransformers_llm* llm, const int* tokens, int n_tokens, int n_past, const void* x_in_dev,
                             void* x_out_dev, int batch);
int ctamd_n_layer(ctransformers_llm* llm);
int ctamd_n_embd(ctransformers_llm* llm);
/* Tokens this handle has evaluated through the prompt-chunk kernels (kernels_pf.h) rather than token by token; lets a
   test assert which path produced the logits it compared. */
long long ctamd_chunk_tokens(ctransformers_llm* llm);
/* K-quant decode mat-vec launches (kernels_v9.h) issued by this process so far (eager launches and graph captures). */
long long ctamd_kq_launches(void);
/* fused QKV + attention launches (kernels_qa9.h) this handle has issued (eager launches and graph captures) */
long long ctamd_qa_launches(ctransformers_llm* llm);
/* prompt-chunk launches on the f16 matrix cores (kernels_pg.h) issued by this process so far */
long long ctamd_pg_launches(void);
/* In-process pipeline (CT_AMD_DEVICES, csrc/pipeline.h): number of stages of this handle (1 = single GPU) and the layer range of a
   stage (returns -1 for a single-stage handle). */
int ctamd_n_stages(ctransformers_llm* llm);
int ctamd_stage_range(ctransformers_llm* llm, int stage, int* layer_begin, int* layer_end);
/* the hand-off between the stages of this handle: "none" (one stage), "flag" (rows stored into the peer-mapped buffer by a kernel, the next
   stage's stream waits on a sequence word: csrc/pipeline.h) or "event" (hipMemcpyPeerAsync + event, CT_AMD_HANDOFF=event) */
const char* ctamd_handoff(ctransformers_llm* llm);
/* Host microseconds the one issuing thread of the in-process pipeline has spent queueing stage `stage`'s launches, event waits and peer
   copies since the handle was created; *evals = the multi-stage evals counted (0.0 for a single-stage handle). */
double ctamd_stage_issue_us(ctransformers_llm* llm, int stage, long long* evals);
/* Round 6 test hooks.
   ctamd_mm8_launches: chunk launches of the order-free prompt kernels (csrc/kernels_mm8.h; CT_AMD_PREFILL=fast) issued by this process.
   ctamd_resident_replays: requests this handle evaluated a second time after a launch that needs its whole grid resident gave up (the device was shared);
     the eval that hit it still returned true.
   ctamd_falcon_fold: 1 if this (falcon) handle rotates / stores Q, K, V in the QKV launch's epilogue (attn_qkv rows reordered at load).
   ctamd_debug_read_kv: fp16 K rows [n_head_kv][n_ctx][head_dim] and V rows [n_embd_gqa][stride] of one layer to host memory; returns the V row stride.
   ctamd_debug_read_attn_out: the attention output rows (n_tok x n_embd floats) of the last prompt chunk launched; returns n_embd.
     n_tok = 0: the one row of the last token step (its last layer's attention output).
   ctamd_attn_free_launches: launches of the order-free long-context decode attention (csrc/kernels_attn9.h:attn_decode9_free_kernel;
     CT_AMD_DECODE_ATTN=fast, opt-in) issued by this process. */
long long ctamd_mm8_launches(void);
long long ctamd_attn_free_launches(void);
long long ctamd_resident_replays(ctransformers_llm* llm);
int ctamd_falcon_fold(ctransformers_llm* llm);
int ctamd_debug_read_kv(ctransformers_llm* llm, int layer, unsigned short* k, unsigned short* v);
int ctamd_debug_read_attn_out(ctransformers_llm* llm, float* dst, int n_tok);
#ifdef __cplusplus
}
#endif
#endif
