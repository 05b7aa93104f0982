/* C ABI of the MI355X-native inference core — the drop-in boundary.
 *
 * These are exactly the 17 symbols the reference's Python FFI binds (reference ctransformers/llm.py:117-208) and the
 * reference library exports (reference models/llm.cc:32-138); each declaration cites the reference definition it
 * replaces.  Plain pointers and sizes only; no C++/torch types cross this boundary.  The implementation behind them
 * is HIP (ctransformers_amd/csrc); there is no CPU path: ctransformers_llm_create returns NULL (with a message on
 * stderr) when no MI355X is visible.
 */
#ifndef CTRANSFORMERS_LLM_H_
#define CTRANSFORMERS_LLM_H_

#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference models/llm.h:6-11 — passed BY VALUE to ctransformers_llm_create. */
struct ctransformers_config {
    int context_length; /* <=0: library default (512 for GGUF, reference llama.cpp:5281) */
    int gpu_layers;     /* reference models/llms/llama.cc:88-95 (layers offloaded to the GPU).  Here every layer is GPU-resident; on a host
                           with several visible MI355X the value sets the pipeline STAGE COUNT: 0 < gpu_layers < n_layer spreads the
                           layers over ceil(n_layer / gpu_layers) GPUs (at most the visible ones; stages balanced by weight bytes),
                           <= 0 or >= n_layer keeps the model on one GPU.  CT_AMD_DEVICES overrides (csrc/pipeline.cc:plan_devices). */
    bool mmap;          /* accepted and ignored: the file goes to the GPU once through pinned staging, nothing of it stays mapped */
    bool mlock;         /* accepted and ignored */
};

typedef struct ctransformers_llm ctransformers_llm; /* opaque handle (reference: class LLM, models/llm.h:13) */

/* models/llm.cc:36-76 — NULL on any failure (unknown type, unreadable file, no GPU). */
ctransformers_llm* ctransformers_llm_create(const char* model_path, const char* model_type,
                                            struct ctransformers_config config);
/* models/llm.cc:78 */
void ctransformers_llm_delete(ctransformers_llm* llm);
/* models/llm.cc:80-85 — caller provides room for strlen(text)+1 ints; returns the token count. */
int ctransformers_llm_tokenize(ctransformers_llm* llm, const char* text, bool add_bos_token, int* output);
/* models/llm.cc:87-89 — pointer into callee-owned storage, valid until the next call; "" for out-of-range ids. */
const char* ctransformers_llm_detokenize(ctransformers_llm* llm, int token);
/* models/llm.cc:91-93 */
bool ctransformers_llm_is_eos_token(ctransformers_llm* llm, int token);
/* models/llm.cc:95 */
int ctransformers_llm_eos_token_id(ctransformers_llm* llm);
/* models/llm.cc:97 */
int ctransformers_llm_bos_token_id(ctransformers_llm* llm);
/* models/llm.cc:99 */
int ctransformers_llm_vocab_size(ctransformers_llm* llm);
/* models/llm.cc:101 — effective n_ctx */
int ctransformers_llm_context_length(ctransformers_llm* llm);
/* models/llm.cc:103-105 — "llama"/"falcon" for GGUF files */
const char* ctransformers_llm_architecture(ctransformers_llm* llm);
/* models/llm.cc:107-112 — THE HOT ENTRY.  Evaluates n_tokens at absolute position n_past in chunks of
 * min(n_ctx, batch_size) (models/llm.h:40-54), n_past clamped to n_ctx - chunk (models/llm.h:126); `threads` is
 * accepted and ignored.  n_past is authoritative and may move backwards between calls (KV overwrite semantics). */
bool ctransformers_llm_batch_eval(ctransformers_llm* llm, const int* tokens, int n_tokens, int n_past,
                                  int batch_size, int threads);
/* models/llm.cc:114-116 — host buffer of n_vocab floats for the last evaluated token; the SAME buffer is returned
 * until the next eval, so in-place edits by the caller persist (reference ctransformers/utils.py:13-44). */
float* ctransformers_llm_logits_data(ctransformers_llm* llm);
int ctransformers_llm_logits_size(ctransformers_llm* llm);
/* models/llm.cc:118-124 — n_embd floats: final-norm output of the last token. */
const float* ctransformers_llm_embeddings_data(ctransformers_llm* llm);
int ctransformers_llm_embeddings_size(ctransformers_llm* llm);
/* models/llm.cc:126-132 — host-side sampler chain (models/llms/llama.cc:53-84); seed<0 -> time(NULL). */
int ctransformers_llm_sample(ctransformers_llm* llm, const int* last_tokens, int n_last, int top_k, float top_p,
                             float temperature, float repetition_penalty, int seed);
/* models/llm.cc:134 */
void ctransformers_llm_reset(ctransformers_llm* llm);

#ifdef __cplusplus
}
#endif
#endif /* CTRANSFORMERS_LLM_H_ */
