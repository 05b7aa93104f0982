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
/* In-kernel s_memtime stamps of workgroup 0 of the last launch of `site` (16 waves x 16 slots of uint64; slots: 0 entry,
 * 1 first loads issued, 2 prologue done, 3 round-0 block math done, 4 barrier passed, 5 chain+epilogue done, 6 exit). */
int ctamd_trace_site(ctransformers_llm* llm, const char* site, unsigned long long* out, int n);
#ifdef __cplusplus
}
#endif
#endif
