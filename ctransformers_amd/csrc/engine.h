// Host layer of the MI355X inference core: model load (GGUF -> repacked device planes), per-context state (KV cache,
// scratch, tables) and the per-token launch sequence that replaces the reference's llama_eval_internal
// (models/ggml/llama.cpp:2835-2981) + llm_build_llama graph (:2162-2491) + ggml_graph_compute.
#pragma once
#include <stdint.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "gpu.h"
#include "host_text.h"
#include "quant.h"

struct MatvecArgs;
struct AttnArgsX;

namespace ctamd {

struct HParams {
    std::string arch;
    int n_vocab = 0, n_embd = 0, n_head = 0, n_head_kv = 0, n_layer = 0, n_ff = 0, n_rot = 0, n_ctx_train = 0;
    float rms_eps = 1e-5f, rope_freq_base = 10000.0f, rope_freq_scale = 1.0f;   // rms_eps doubles as falcon's LayerNorm eps
    bool falcon() const { return arch == "falcon"; }
    bool gpt2() const { return arch == "gpt2"; }
    bool mpt() const { return arch == "mpt"; }
    bool legacy() const { return gpt2() || mpt(); }   // pre-GGUF GGML files: the reference's LLM base class (models/llm.h) serves them
    int head_dim() const { return n_embd / n_head; }
    int n_embd_gqa() const { return head_dim() * n_head_kv; }
};

struct Layer {
    float* attn_norm = nullptr;
    float* ffn_norm = nullptr;
    DevMat wq, wk, wv, wo, w_gate, w_up, w_down;
    DevMat w_gu;   // fused gate/up matrix in LAYOUT_R2C4 (decode: LAYOUT_L9 arena, kernels_v9.h; prompt chunks: LAYOUT_R2C4): pair u = (gate row u, up row u)
    // falcon (llm_build_falcon, llama.cpp:2493-2798): LayerNorm biases, optional second norm (40B), fused QKV
    float* attn_norm_b = nullptr;
    float* attn_norm2 = nullptr;
    float* attn_norm2_b = nullptr;
    DevMat wqkv;
    DevMat wq_v, wk_v, wv_v;   // falcon with folded rows (Engine::falcon_fold_): the Q / K / V row ranges of wqkv's LAYOUT_L9 arena as three jobs of one launch
    // gpt2 (gpt2.cc:391-699): second norm (ln_2) reuses ffn_norm + ffn_norm_b; row biases of the four mat-muls
    float* ffn_norm_b = nullptr;
    float *b_qkv = nullptr, *b_wo = nullptr, *b_up = nullptr, *b_down = nullptr;
};

long long pg_launches();   // test hook: prompt-chunk launches of kernels_pg.h issued by this process
long long kq_launches();   // test hook: K-quant decode mat-vec launches (kernels_v9.h) issued by this process
long long attn_free_launches();   // test hook: order-free decode attention launches (kernels_attn9.h:attn_decode9_free_kernel) issued by this process
long long mm8_launches();  // test hook: prompt-chunk launches of the order-free kernels (kernels_mm8.h) issued by this process

// Process-wide and recursive: a stream capture (hipStreamCaptureModeThreadLocal) tolerates other threads' launches on their own streams, but not an operation
// on the legacy stream or a device-wide synchronization anywhere in the process while it is open (hipMemcpy / hipMemset / hipFree of a load or a delete in
// another thread: "operation would make the legacy stream depend on a capturing blocking stream", and the capture is lost).  Handle creation and deletion
// hold this mutex from start to end, every capture holds it from Begin to End; the eval path itself makes no legacy-stream call.
std::recursive_mutex& capture_mutex();

class Engine {
   public:
    Engine() {}
    ~Engine();
    // Loads the model onto the visible GPU(s).  Fails (false + message) when no HIP device is present: there is no
    // CPU path in this library.
    // layer_begin/layer_end select a pipeline STAGE (layers [begin, end) only; -1/-1 = the whole model): a stage with
    // begin > 0 has no embedding table, a stage with end < n_layer has no output head.  `device` is the HIP ordinal.
    bool load(const std::string& path, int context_length, int gpu_layers, std::string& err, int layer_begin = -1,
              int layer_end = -1, int device = 0);
    // Legacy (pre-GGUF) GGML file of the GPT-2 family (reference models/llms/gpt2.cc); n_ctx comes from the file.
    bool load_gpt2(const std::string& path, std::string& err, int device = 0, bool starcoder = false);
    // Legacy GGML file of the MPT family (reference models/llms/mpt.cc); n_ctx = min(max_seq_len, context_length or 2048).
    bool load_mpt(const std::string& path, int context_length, std::string& err, int device = 0);
    // Evaluate `n` tokens at absolute positions n_past..n_past+n-1 (KV cache overwrite semantics); logits and the
    // final-norm embedding of the LAST token land in the pinned host buffers.
    // batch > 0: evaluate the n tokens exactly as the reference would in batches of `batch` (models/llm.h:40-54), in one go
    bool eval(const int* tokens, int n, int n_past, std::string& err, int batch = 0);
    // whether batch_eval may hand a whole multi-batch request to one eval (the chunk kernels then see up to 128 tokens at once)
    bool coalesces_batches() const { return pf_ok_ && l0_ == 0 && l1_ == hp_.n_layer; }
    // Stage form: x_in_dev / x_out_dev are DEVICE pointers to [n][n_embd] f32 residual-stream rows (the hand-off between
    // pipeline stages).  x_in_dev is required iff layer_begin > 0, x_out_dev iff layer_end < n_layer.
    bool eval_stage(const int* tokens, int n, int n_past, const float* x_in_dev, float* x_out_dev, std::string& err, int batch = 0);
    // The same evaluation in pieces, all asynchronous on this stage's stream — what the in-process pipeline (pipeline.cc) drives:
    //   req_begin   cursor + token ids of a whole request -> device (stage 0 uses the ids, every stage the cursor)
    //   req_range   tokens [c0, c0 + nt) of the request through this stage's layers; the rows come from / go to xio() rows
    //               [c0, c0 + nt) (ranges must be submitted in order: the device cursor advances with them)
    //   req_logits  (last stage) logits + final-norm embedding of the request's last token -> pinned host buffers
    //   req_wait    drain the stream; marks the logits valid
    bool req_begin(const int* tokens, int n, int n_past, int batch, std::string& err, bool upload = true);   // upload = false: the cursor reaches the device
                                                                                                              // with the previous stage's hand-off kernel (pipeline.cc)
    // One token step's launches on this stage's stream (the pipeline captures the steps of stages that share a stream into ONE graph: pipeline.cc)
    bool capture_step(bool want_logits, std::string& err) { return token_step(want_logits, err); }
    bool uses_graphs() const { return use_graph_ && !dump_dir_; }
    const int* req_cursor() const { return h_scalars_; }   // {step, pos, n_past + n, batch} of the request in flight
    int* state_dev() { return d_state_; }
    bool req_range(int c0, int nt, bool last_of_request, std::string& err);
    bool req_logits(std::string& err);
    bool req_wait(int n, int n_past, std::string& err);
    float* xio() { return xio_; }
    hipStream_t stream() { return stream_; }
    // Pipeline stages that share a device run on ONE stream (pipeline.cc): the stage gives up its own (idle) stream and queues behind the stage before it —
    // no event, no cross-stream wait between them.  Called once, right after load.
    bool adopt_stream(hipStream_t s);
    int device() const { return device_; }
    int layer_begin() const { return l0_; }
    int layer_end() const { return l1_; }
    bool has_head() const { return l1_ == hp_.n_layer; }

    const HParams& hparams() const { return hp_; }
    const Vocab& vocab() const { return vocab_; }
    int n_ctx() const { return n_ctx_; }
    void reset() { if (hp_.legacy()) have_logits_ = false; }   // reference models/llm.h:106: legacy models forget their logits
    // Logits and embeddings stay on the GPU until somebody asks for them (fetch_outputs: one synchronous 144 KB copy for a 7B); a greedy
    // step needs 4 bytes (greedy_token).  Once the host copy exists the caller may have edited it (the reference's Python exposes
    // the logits as a writable view), so sampling then runs on the host copy as the reference does.
    float* logits() { fetch_outputs(); return h_logits_; }
    int logits_size() const { return have_logits_ ? hp_.n_vocab : 0; }
    const float* embeddings() { fetch_outputs(); return h_emb_; }
    bool greedy_token(int& token) const { if (!have_logits_ || outputs_on_host_ || hp_.legacy()) return false; token = h_scalars_[n_ctx_ + 12 + cur_buf_]; return true; }
    void fetch_outputs();
    // Greedy chains.  ctransformers_llm_sample tells the engine whether the caller's last pick was the device-side
    // first maximum: after such a pick the next eval is, in every generate() loop, that token at the next position — the engine then
    // queues that token step BEHIND the one it is waiting for (token id, cursor and embedding row are already on the device: the head
    // launch prepared them), and an eval that asks for exactly it only waits for its event.  Anything else the caller does is served
    // as before (the speculative step wrote a KV position nobody has evaluated yet and a logits buffer nobody reads).
    void note_sample(bool device_greedy) { greedy_armed_ = device_greedy; }
    // Launches whose workgroups wait for each other (fused QKV + attention, shared score rows: kernels_qa9.h / kernels_attn9.h) need their whole grid
    // resident; on a shared device a sweep may give up and raise a pinned word.  The reference never fails an eval because the machine is shared
    // (models/llm.h:40-54), so: the stage switches to the forms that need no residency (for good) and the caller REPLAYS the request — every position it
    // wrote is simply written again (KV overwrite semantics).  resident_timeout(): the word was raised since the last call (the stream is drained, the
    // forms are off, the word is clear); disable_resident_forms(): the same switch without a timeout (pipeline stages whose device also runs a stream wait).
    bool resident_timeout();
    void disable_resident_forms();
    long long resident_replays() const { return resident_replays_; }
    int embeddings_size() const { return have_logits_ && !hp_.legacy() ? hp_.n_embd : 0; }   // legacy models expose none (models/llm.h:73)
    size_t weight_bytes() const { return weight_bytes_; }
    int read_stamps(unsigned long long* out, int max);   // measurement only: copies and clears the stamps
    // tests / measurement only: this stage's fp16 K and V cache of one layer -> host ([n_head_kv][n_ctx][head_dim] and [n_embd_gqa][v_stride]); returns v_stride
    int debug_read_kv(int layer, uint16_t* k, uint16_t* v);
    int debug_read_attn_out(float* dst, int n_tok);
    bool falcon_fold() const { return falcon_fold_; }
    long long qa_launches() const { return qa_launches_; }   // fused QKV + attention launches issued (eager launches and graph captures)
    long long spec_hits() const { return spec_hits_; }           // evals served by a speculative continuation step
    long long spec_launched() const { return spec_launched_; }   // continuation steps queued
    long long chunk_tokens() const { return chunk_tokens_; }

    // Measurement hook (exported as ctamd_profile_decode): replays the LAST evaluated token `iters` times with eager
    // launches bracketed by HIP events on the engine's stream; one entry per launch site, times summed over iters.
    struct LaunchStat { const char* site; const char* kernel; double bytes; double ms; int launches; };
    bool profile_decode(int iters, std::vector<LaunchStat>& out, std::string& err);
    // measurement only: run ONE launch site of the last token with in-kernel timestamps (CT_AMD_DBG=32) and copy them out
    bool trace_site(const char* site, unsigned long long* out, int n, std::string& err);
    // measurement only (ctamd_decode_burst): n token steps with the head, queued back to back on the stream with no host round trip
    // between them (HIP events around the burst) -> device microseconds per token step.  The cursor continues from the last eval.
    bool decode_burst(int n, double* us_per_token, std::string& err);
    const char* trace_site_ = nullptr;
    unsigned long long* trace_buf_ = nullptr;
    void apply_trace(::MatvecArgs& a, const char* site);
    // measurement only: when set, token_step launches nothing but this site's kernels (one per layer, back to back)
    const char* only_site_ = nullptr;
    int site_launches_ = 0;
    bool site_on(const char* site) {
        if (only_site_ && strcmp(only_site_, site) != 0) return false;
        ++site_launches_;
        return true;
    }

   private:
    bool upload_matrix(const struct GgufTensor* t, DevMat& m, bool keep_raw, std::string& err);
    bool upload_r2c4(const std::vector<std::pair<const struct GgufTensor*, DevMat*>>& parts, bool fuse, std::string& err);
    bool upload_l9b(const std::vector<std::pair<const GgufTensor*, DevMat*>>& parts, bool fuse, std::string& err);
    bool upload_m8(const std::vector<std::pair<const GgufTensor*, DevMat*>>& parts, bool fuse, std::string& err);   // LAYOUT_M8 arenas (kernels_mm8.h), when the handle runs the order-free prompt kernels
    bool upload_f32(const struct GgufTensor* t, float** out, int n, std::string& err);
    bool build_tables(std::string& err);
    bool token_step(bool want_logits, std::string& err);
    bool chunk_step(int c0, int nt, bool want_logits, std::string& err);
    bool chunk_step_falcon(int c0, int nt, bool want_logits, std::string& err);
    bool chunk_step_gpt2(int nt, bool want_logits, std::string& err);
    bool chunk_step_mpt(int nt, bool want_logits, std::string& err);
    bool run_chunk(int c0, int nt, bool want_logits, std::string& err);    // chunk_step, replayed from a hipGraph where it can be   // prompt chunk of 2..kPfChunk tokens (kernels_pf.h)
    bool pg_matvec(MatvecArgs& m, const float* x, int ldx, int nt, int ld_out, int ld_res, std::string& err);
    bool pf_matvec(::MatvecArgs& m, const float* x, int ldx, int nt, int ld_out, int ld_res, const char* site, double bytes, std::string& err);
    bool mm8_can(const ::MatvecArgs& m) const;   // every job of the site has a LAYOUT_M8 arena the order-free kernels take (kernels_mm8.h)
    bool mm8_matvec(::MatvecArgs& m, const float* x, int ldx, int nt, int ld_out, int ld_res, std::string& err);
    void launch_attention(uint16_t* kc, uint16_t* vc, int nt = 0);
    void fill_attn_args(::AttnArgsX& ax, uint16_t* kc, uint16_t* vc, int nt);
    bool qa_can(const Layer& L) const;   // this layer's token step takes the fused QKV + attention launch (kernels_qa9.h)
    bool launch_qkv_attn(::MatvecArgs& a, uint16_t* kc, uint16_t* vc, int il, std::string& err);
    bool token_step_falcon(bool want_logits, std::string& err);
    bool token_step_gpt2(bool want_logits, std::string& err);
    bool token_step_mpt(bool want_logits, std::string& err);
    bool warm_up(std::string& err);       // first-use costs (code object, LDS opt-ins, graph capture) paid at load
    bool alloc_state(std::string& err);   // KV cache, scratch, pinned host buffers, tables
    bool run_matvec(::MatvecArgs& a, std::string& err);
    bool ensure_graphs(std::string& err);
    void free_all();
    void debug_dump(const char* site, int layer);
    const char* dump_dir_ = nullptr;
    int dump_seq_ = 0;

    HParams hp_;
    Vocab vocab_;
    int n_ctx_ = 0, v_stride_ = 0;
    int l0_ = 0, l1_ = 0, device_ = 0;
    float* xio_ = nullptr;  // [n_ctx][n_embd] residual-stream rows handed between pipeline stages (partial stages only)
    DevMat tok_embd_, output_;
    float* output_norm_ = nullptr;
    float* output_norm_b_ = nullptr;
    float *qkv_tmp_ = nullptr, *attn_proj_ = nullptr;   // falcon scratch: un-rotated fused QKV rows, Wo output
    float* wpe_ = nullptr;                              // gpt2 learned position embeddings [n_ctx][n_embd]
    float *kmem_ = nullptr, *vmem_ = nullptr;           // gpt2 F32 KV cache [n_layer][n_ctx][n_embd] each
    float* alibi_ = nullptr;                            // mpt: per-head ALiBi slopes m_k (ggml.c:12228-12247), else null
    float* zero_bias_ = nullptr;                        // mpt: its LayerNorms have no bias; the shared prologue adds this +0 vector
    float clip_qkv_ = 0.0f;                             // mpt: clamp of the fused QKV rows (0 = none)
    std::vector<Layer> layers_;
    size_t weight_bytes_ = 0;

    hipStream_t stream_ = nullptr;
    bool stream_owned_ = true;
    uint16_t* kcache_ = nullptr;
    uint16_t* vcache_ = nullptr;
    float *x_ = nullptr, *attn_out_ = nullptr, *h_ = nullptr, *scores_ = nullptr, *d_logits_ = nullptr, *d_emb_ = nullptr;
    float* f16_tmp_ = nullptr;   // models with F16 weight matrices: raw rows of one mat-vec site (kernels_f16.h)
    bool has_raw_ = false;   // some matrix stays in file layout (F16, Q4_1, Q5_0, Q5_1): two-launch sites, token steps only
    uint16_t* q_f16_ = nullptr;
    // prompt-chunk scratch (rows of kPfChunk tokens): residual stream, attention output, FFN hidden, fp16 queries, Q8_K images
    float *xb_ = nullptr, *attn_out_b_ = nullptr, *hb_ = nullptr;
    uint16_t* q_f16_b_ = nullptr;
    int* acts_ = nullptr;
    float *qkv_tmp_b_ = nullptr, *attn_proj_b_ = nullptr;   // falcon chunks: fused QKV rows, Wo output
    bool pf_ok_ = false;    // llama architecture, every layer matrix a K-quant in the tile layout, K <= 12288
    int pf_min_ = 2;        // chunks shorter than this run token by token
    int pf_chunk_ = 128;    // tokens per chunk_step (<= pf_cap_; CT_AMD_PF_CHUNK lowers it)
    int pf_cap_ = 128;      // rows of the chunk scratch: kPfChunk, or kPfChunkFast where every site of the graph takes the order-free kernels (kernels_mm8.h),
                            // whose launches fill the chip better with more token tiles
    long long chunk_tokens_ = 0;
    const char* pg_trace_site_ = nullptr;   // measurement only (CT_AMD_PG_TRACE)
    int pg_force_tg_ = 0;   // tests: 16 / 32 tokens per workgroup
    // Prompt chunks in the order-free form (kernels_mm8.h; CT_AMD_PREFILL=fast|exact, read at load): the reference's quantization points and exact integer
    // dots, f32 sums in any order — logits within 1e-3 of the reference instead of bit-identical.  Token steps are untouched.
    bool fast_pf_ = false;
    uint8_t* acts8_ = nullptr;    // activation units [K-step][token tile] (kernels_mm8.h)
    size_t acts8_bytes_ = 0;
    int mm8_force_ntt_ = 0, mm8_force_ks_ = 0;   // experiments (CT_AMD_MM8_SHAPE="ntt,ks")
    uint8_t* acts_h_ = nullptr;   // stage images (kernels_pg.h): [layout 45 | layout 6]
    size_t acts_h_half_ = 0;
    float* rope_cs_ = nullptr;
    uint16_t *exp_tab_ = nullptr, *silu_tab_ = nullptr, *gelu_tab_ = nullptr;
    int *d_tokens_ = nullptr, *d_state_ = nullptr;  // token ids of the current chunk; {step, pos} cursor
    float *h_logits_ = nullptr, *h_emb_ = nullptr;
    int* h_scalars_ = nullptr;  // pinned staging for the token ids + cursor
    bool use_graph_ = false;
    // load pipeline: the file's tensor bytes go to the GPU once, in file layout, through pinned staging (pread in parallel, async
    // copies), and kernels repack them into the LAYOUT_R2C4 arenas there
    bool stage_file(const class GgufFile& f, const std::vector<const struct GgufTensor*>& need, std::string& err);
    void release_staged();
    const uint8_t* staged(const struct GgufTensor* t) const;
    bool falcon_fold_ = false;           // falcon: attn_qkv rows permuted at load (NEOX pairs adjacent), RoPE + fp16 Q + KV append in the QKV launch's epilogue
    uint8_t* perm_scratch_ = nullptr;    // load only
    size_t perm_scratch_bytes_ = 0;
    uint8_t* dev_file_ = nullptr;        // [file_lo_, file_hi_) of the mapping, on the device; freed at the end of load()
    const uint8_t* file_lo_ = nullptr;
    const uint8_t* file_hi_ = nullptr;
    double load_stage_s_ = 0.0;
    int req_past_ = 0;               // n_past of the request being evaluated
    bool chunk_below_128_ = false;   // every position of the chunk being launched is < 128 (attn_chunk_tile_kernel applies)
#ifndef CT_EMU
    hipGraphExec_t graph_step_ = nullptr, graph_step_head_ = nullptr;
    static constexpr size_t kMaxChunkGraphs = 64;
    struct ChunkGraph { hipGraphExec_t exec; unsigned long long last_use; };
    std::map<long long, ChunkGraph> chunk_graphs_;   // prompt chunks, keyed by (stage: row offset + 1) << 24 | 4 * n_tokens + 2 * below-128 + want_logits; captured on second use
    std::map<long long, int> chunk_seen_;
    unsigned long long chunk_use_clock_ = 0;         // least-recently-used eviction past kMaxChunkGraphs
    std::vector<hipGraphExec_t> retired_graphs_;     // evicted while launches of them may still be in flight: destroyed behind the next stream sync (req_wait)
#endif
    bool have_logits_ = false;
    bool outputs_on_host_ = true;   // false after an eval until logits() / embeddings() fetched them
    int* d_argmax_ = nullptr;
    // Output buffers come in pairs: [0] is written by every eval the caller asked for, a speculative continuation step writes the one
    // the last committed eval did NOT write.  d_logits_ / d_emb_ / d_argmax_ / pick_host_ name the pair member the launches being
    // issued (or captured) write; cur_buf_ the one holding the last committed eval's outputs.
    float* d_logits2_[2] = {nullptr, nullptr};
    int* d_argmax2_[2] = {nullptr, nullptr};
    int* pick_host_ = nullptr;        // device view of h_scalars_[n_ctx_ + 12 + buffer]
    int* pick_host2_[2] = {nullptr, nullptr};
    uint32_t* xq_ = nullptr;          // fused QKV + attention launch: the KV-head groups' exchange records (kernels_qa9.h)
    int* qa_err_ = nullptr;           // device view of the pinned word h_scalars_[n_ctx_ + 14] a timed-out sweep raises
    bool fuse_qa_ = true;             // CT_AMD_FUSE_QA (read at load)
    bool attn_free_ = false;          // CT_AMD_DECODE_ATTN=fast (opt-in): order-free V*P in the long-context decode attention (kernels_attn9.h:attn_decode9_free_kernel)
    bool attn_share_ = true;          // CT_AMD_ATTN_SHARE: long-context decode attention shares one score row per head (kernels_attn9.h)
    uint32_t* xs_ = nullptr;          // the shared score rows: [n_head][n_ctx] granules
    int cur_layer_ = 0;               // layer whose launches are being issued (tags of the in-launch exchanges)
    long long qa_launches_ = 0;
    long long resident_replays_ = 0;   // requests evaluated a second time after a residency give-up
    int dbg_qa_timeout_ = 0;           // tests (CT_AMD_DBG_QA_TIMEOUT=N): the N-th wait of this handle behaves as if a sweep had given up
    unsigned* pick_ws_ = nullptr;     // head launch: one 64-bit key per wave, [workgroup][16] (kernels_v9.h:v9_pick_store)
    int cur_buf_ = 0;
    void select_out(int buf) { d_logits_ = d_logits2_[buf]; d_emb_ = d_logits_ + hp_.n_vocab; d_argmax_ = d_argmax2_[buf]; pick_host_ = pick_host2_[buf]; }
    bool head_folds() const;          // the head launch picks the greedy token itself (no argmax launch, no 4-byte copy)
    void set_head_fold(::MatvecArgs& a, bool cont);
    void launch_pick(const ::MatvecArgs& head);   // behind the head launch `head` (its grid = the key slots)
    bool head_folded_ = false;        // a head launch of this handle has left per-wave keys and been followed by pick_cont_kernel (set where that is launched or captured)
    int stamps_level_ = 0;
    unsigned long long* stamps_ = nullptr;   // measurement only (CT_AMD_STAMPS=1): wall-clock stamps at the start and end of every token step
    bool head_cont_ = false;          // the head launch being issued belongs to a token step that advanced the cursor before it
    int fold_on_ = 1;       // CT_AMD_HEAD_FOLD (read at load: A/B and tests): 0 argmax over the logits, 1 per-workgroup keys from the head launch + pick_cont_kernel with the continuation, 2 without the continuation
    bool spec_on_ = true;   // CT_AMD_SPEC
    bool cont_mode_ = false;          // token_step is being captured as a continuation step: no embedding launch (the row is there)
    bool greedy_armed_ = false;       // the caller's last sample() was the device-side greedy pick
    bool spec_want_ = false;          // the eval being issued is a step of a greedy chain: queue the guessed next step behind it
    bool spec_inflight_ = false;      // a continuation step for position spec_pos_ is queued (outputs: buffer spec_buf_)
    int spec_buf_ = 0, spec_pos_ = -1;
    int last_pick_ = -1;              // greedy pick of the last committed eval
    long long spec_hits_ = 0, spec_launched_ = 0;
#ifndef CT_EMU
    hipGraphExec_t graph_cont_[2] = {nullptr, nullptr};
    hipEvent_t ev_step_[2] = {nullptr, nullptr};
    bool ev_pending_ = false;         // req_wait waits for ev_step_[0] instead of draining the stream (a speculative step is queued behind)
    bool spec_possible() const;
    bool launch_spec(int buf, int pos, std::string& err);
    bool drain_spec(std::string& err);
#endif
    int last_token_ = -1, last_pos_ = -1;
    int req_n_ = 0;           // tokens of the request in flight (req_begin)
    struct ProfRec { const char* site; const char* kernel; double bytes; void* e0; void* e1; };
    std::vector<ProfRec>* prof_ = nullptr;
    void prof_begin(const char* site, const char* kernel, double bytes);
    void prof_end();
    std::vector<void*> dev_allocs_;
};

}  // namespace ctamd
