// The 17-symbol C ABI (include/ctransformers_llm.h) over ctamd::Engine.
#include "../../include/ctransformers_llm.h"
#include "../../include/ctransformers_amd_ext.h"

#include <ctype.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "engine.h"
#include "pipeline.h"

// A handle = a pipeline of one or more stages (pipeline.h; one stage = the plain single-GPU engine).  `engine` is the stage that
// answers for the model as a whole: hyper-parameters, vocabulary (stage 0) — logits and embeddings come from the last stage.
struct ctransformers_llm {
    ctamd::Pipeline pipe;
    ctamd::Engine& engine() { return pipe.first(); }
    ctamd::Engine& tail() { return pipe.last(); }
    std::string arch;
    std::string piece;  // storage behind ctransformers_llm_detokenize
};

static bool file_is_gguf(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    uint32_t magic = 0;
    const size_t n = fread(&magic, 1, 4, f);
    fclose(f);
    return n == 4 && magic == 0x46554747u;
}

extern "C" {

ctransformers_llm* ctransformers_llm_create(const char* model_path, const char* model_type,
                                            struct ctransformers_config config) {
    if (!model_path || !model_type) return nullptr;
    std::string type;
    for (const char* p = model_type; *p; ++p)
        if (isalnum((unsigned char)*p)) type.push_back(*p);
    const bool gguf = file_is_gguf(model_path);   // the GGUF magic overrides model_type (reference models/llm.cc:45)
    const bool starcoder = type == "starcoder" || type == "gptbigcode";
    const bool mpt = type == "mpt";
    if (!gguf && type != "gpt2" && !starcoder && !mpt) {
        // Of the legacy (pre-GGUF) GGML architectures of the reference (models/llm.cc:47-65) gpt2, starcoder / gptbigcode (one
        // container, one graph) and mpt are served.
        fprintf(stderr, "Model type '%s' is not supported.\n", model_type);
        return nullptr;
    }
    std::lock_guard<std::recursive_mutex> legacy(ctamd::capture_mutex());   // loads copy on the legacy stream: not while another thread captures (engine.h)
    ctransformers_llm* llm = nullptr;
    try {   // nothing may propagate across the C boundary: a malformed file makes create return NULL (as the reference does)
    llm = new ctransformers_llm;
    std::string err;
    // The GPUs whose HBM the layers are spread over, as one in-process pipeline: CT_AMD_DEVICES ("4" or "0,1,2,3") when set, else
    // gpu_layers layers per GPU over the visible devices (pipeline.cc:plan_devices; the Config struct of the ABI cannot grow).
    const bool ok = gguf ? llm->pipe.load(model_path, config.context_length, config.gpu_layers,
                                          ctamd::plan_devices(model_path, config.gpu_layers, getenv("CT_AMD_DEVICES")), err)
                         : mpt ? llm->pipe.load_mpt(model_path, config.context_length, err)
                         : llm->pipe.load_gpt2(model_path, err, starcoder);
    if (!ok) {
        fprintf(stderr, "ctransformers_amd: failed to load '%s': %s\n", model_path, err.c_str());
        delete llm;
        return nullptr;
    }
    if (!gguf) { llm->arch = ""; return llm; }   // legacy models report an empty architecture string (models/llm.h:113)
    llm->arch = llm->engine().hparams().arch;
    return llm;
    } catch (const std::exception& e) {
        fprintf(stderr, "ctransformers_amd: failed to load '%s': %s\n", model_path, e.what());
        delete llm;
        return nullptr;
    }
}

void ctransformers_llm_delete(ctransformers_llm* llm) {
    std::lock_guard<std::recursive_mutex> legacy(ctamd::capture_mutex());   // hipFree synchronizes the device: not while another thread captures (engine.h)
    delete llm;
}

int ctransformers_llm_tokenize(ctransformers_llm* llm, const char* text, bool add_bos_token, int* output) {
    const std::vector<int> t = llm->engine().vocab().tokenize(text, add_bos_token);
    std::copy(t.begin(), t.end(), output);
    return (int)t.size();
}

const char* ctransformers_llm_detokenize(ctransformers_llm* llm, int token) {
    llm->piece = llm->engine().vocab().piece(token);
    return llm->piece.c_str();
}

bool ctransformers_llm_is_eos_token(ctransformers_llm* llm, int token) {
    const auto& v = llm->engine().vocab();
    if (token == v.eos_id) return true;
    // StarChat / Dolly end markers of legacy files with special pieces (reference models/llm.h:78-89 LLM::IsEosToken)
    if (!v.special.empty()) {
        const std::string text = v.piece(token);
        return text == "<|end|>" || text == "### End";
    }
    return false;
}
int ctransformers_llm_eos_token_id(ctransformers_llm* llm) { return llm->engine().vocab().eos_id; }
int ctransformers_llm_bos_token_id(ctransformers_llm* llm) { return llm->engine().vocab().bos_id; }
int ctransformers_llm_vocab_size(ctransformers_llm* llm) { return llm->engine().hparams().n_vocab; }
int ctransformers_llm_context_length(ctransformers_llm* llm) { return llm->engine().n_ctx(); }
const char* ctransformers_llm_architecture(ctransformers_llm* llm) { return llm->arch.c_str(); }

bool ctransformers_llm_batch_eval(ctransformers_llm* llm, const int* tokens, int n_tokens, int n_past, int batch_size,
                                  int threads) {
    (void)threads;
    const int n_ctx = llm->engine().n_ctx();
    batch_size = std::min(n_ctx, batch_size);
    if (batch_size <= 0) return n_tokens <= 0;
    // The reference evaluates batch after batch (models/llm.h:40-54).  Where the prompt-chunk kernels apply, the whole request
    // goes down as one eval: the only thing a batch boundary changes in the arithmetic is the length of the attention-value
    // dot product, which the attention kernel derives per token from the batch size (kernels_exact.h), so the result is
    // bit-identical to the batch-by-batch evaluation while the weights are passed over once per 128 tokens instead of
    // once per batch (the reference's default batch is 8).  Requests that run into the context clamp keep the loop.
    if (n_tokens > batch_size && n_past >= 0 && n_past + n_tokens <= n_ctx && llm->pipe.coalesces_batches()) {
        std::string err;
        if (!llm->pipe.eval(tokens, n_tokens, n_past, err, batch_size)) {
            fprintf(stderr, "ctransformers_amd: eval failed: %s\n", err.c_str());
            return false;
        }
        return true;
    }
    for (int start = 0; start < n_tokens; start += batch_size) {
        const int n = std::min(batch_size, n_tokens - start);
        const int past = std::min(n_ctx - n, n_past);  // reference models/llm.h:126
        std::string err;
        if (!llm->pipe.eval(tokens + start, n, past, err)) {
            fprintf(stderr, "ctransformers_amd: eval failed: %s\n", err.c_str());
            return false;
        }
        n_past += n;
    }
    return true;
}

float* ctransformers_llm_logits_data(ctransformers_llm* llm) { return llm->tail().logits(); }
int ctransformers_llm_logits_size(ctransformers_llm* llm) { return llm->tail().logits_size(); }
const float* ctransformers_llm_embeddings_data(ctransformers_llm* llm) { return llm->tail().embeddings(); }
int ctransformers_llm_embeddings_size(ctransformers_llm* llm) { return llm->tail().embeddings_size(); }

int ctransformers_llm_sample(ctransformers_llm* llm, const int* last_tokens, int n_last, int top_k, float top_p,
                             float temperature, float repetition_penalty, int seed) {
    if (llm->tail().logits_size() == 0) return llm->engine().vocab().eos_id;
    // top_k <= 1 (llama_sample_top_k keeps max(k, 1) candidates) without an effective repetition penalty is the FIRST maximum of the
    // logits whatever top_p / temperature / seed are (llama.cc:53-84: one candidate is left, llama_sample_token draws index 0): taken on
    // the GPU — unless the caller has fetched, and possibly edited, the logits since the eval (then the host chain below runs on them)
    int greedy = 0;
    if (top_k <= 1 && (repetition_penalty == 1.0f || n_last <= 0) && llm->tail().greedy_token(greedy)) {
        llm->tail().note_sample(true);   // a greedy chain: the engine may queue the next token step before it is asked for (engine.h)
        return greedy;
    }
    llm->tail().note_sample(false);
    if (llm->engine().vocab().type == ctamd::VOCAB_GPT)
        return ctamd::sample_token_gpt(llm->tail().logits(), llm->engine().hparams().n_vocab, last_tokens, n_last, top_k, top_p,
                                       temperature, repetition_penalty, seed);
    return ctamd::sample_token(llm->tail().logits(), llm->engine().hparams().n_vocab, last_tokens, n_last, top_k, top_p,
                               temperature, repetition_penalty, seed);
}

// Reference Reset() (models/llm.h:106) clears the logits of LEGACY models only: logits_size() is 0 and sample() returns EOS until the
// next eval; GGUF models keep theirs.
void ctransformers_llm_reset(ctransformers_llm* llm) { llm->tail().reset(); }

// ---- measurement extensions (include/ctransformers_amd_ext.h); not part of the reference ABI ----------------------
int ctamd_profile_decode(ctransformers_llm* llm, int iters, ctamd_launch_stat* out, int max_out) {
    std::vector<ctamd::Engine::LaunchStat> st;
    std::string err;
    if (!llm->engine().profile_decode(iters, st, err)) {
        fprintf(stderr, "ctransformers_amd: profile failed: %s\n", err.c_str());
        return -1;
    }
    int n = 0;
    for (auto& s : st) {
        if (n >= max_out) break;
        snprintf(out[n].site, sizeof(out[n].site), "%s", s.site);
        out[n].bytes = s.bytes;
        out[n].ms = s.ms;
        out[n].launches = s.launches;
        ++n;
    }
    return n;
}
ctransformers_llm* ctamd_stage_create(const char* model_path, int context_length, int layer_begin, int layer_end,
                                      int device) {
    if (!model_path) return nullptr;
    ctransformers_llm* llm = new ctransformers_llm;
    std::string err;
    if (!llm->pipe.load_stage(model_path, context_length, layer_begin, layer_end, device, err)) {
        fprintf(stderr, "ctransformers_amd: failed to load stage [%d,%d) of '%s': %s\n", layer_begin, layer_end, model_path,
                err.c_str());
        delete llm;
        return nullptr;
    }
    llm->arch = llm->engine().hparams().arch;
    return llm;
}

int ctamd_stage_eval(ctransformers_llm* llm, const int* tokens, int n_tokens, int n_past, const void* x_in_dev,
                     void* x_out_dev) {
    std::string err;
    if (!llm->engine().eval_stage(tokens, n_tokens, n_past, (const float*)x_in_dev, (float*)x_out_dev, err)) {
        fprintf(stderr, "ctransformers_amd: stage eval failed: %s\n", err.c_str());
        return -1;
    }
    return 0;
}

int ctamd_stage_eval_batched(ctransformers_llm* llm, const int* tokens, int n_tokens, int n_past, const void* x_in_dev,
                             void* x_out_dev, int batch) {
    std::string err;
    if (!llm->engine().eval_stage(tokens, n_tokens, n_past, (const float*)x_in_dev, (float*)x_out_dev, err, batch)) {
        fprintf(stderr, "ctransformers_amd: stage eval failed: %s\n", err.c_str());
        return -1;
    }
    return 0;
}

int ctamd_n_layer(ctransformers_llm* llm) { return llm->engine().hparams().n_layer; }
int ctamd_n_embd(ctransformers_llm* llm) { return llm->engine().hparams().n_embd; }
long long ctamd_chunk_tokens(ctransformers_llm* llm) { return llm->engine().chunk_tokens(); }
long long ctamd_kq_launches(void) { return ctamd::kq_launches(); }
long long ctamd_qa_launches(ctransformers_llm* llm) { return llm->engine().qa_launches(); }
long long ctamd_pg_launches(void) { return ctamd::pg_launches(); }
long long ctamd_mm8_launches(void) { return ctamd::mm8_launches(); }
long long ctamd_attn_free_launches(void) { return ctamd::attn_free_launches(); }
long long ctamd_resident_replays(ctransformers_llm* llm) {   // requests evaluated a second time after a residency give-up, summed over the stages
    long long n = 0;
    for (int s = 0; s < llm->pipe.n_stages(); ++s) n += llm->pipe.stage(s).resident_replays();
    return n;
}
int ctamd_falcon_fold(ctransformers_llm* llm) { return llm->engine().falcon_fold() ? 1 : 0; }
int ctamd_debug_read_attn_out(ctransformers_llm* llm, float* dst, int n_tok) { return llm->engine().debug_read_attn_out(dst, n_tok); }
int ctamd_debug_read_kv(ctransformers_llm* llm, int layer, unsigned short* k, unsigned short* v) { return llm->engine().debug_read_kv(layer, k, v); }
int ctamd_n_stages(ctransformers_llm* llm) { return llm->pipe.n_stages(); }
const char* ctamd_handoff(ctransformers_llm* llm) { return llm->pipe.handoff(); }
int ctamd_stage_range(ctransformers_llm* llm, int stage, int* layer_begin, int* layer_end) {
    if (stage < 0 || stage >= llm->pipe.n_stages() || llm->pipe.ranges().empty()) return -1;
    *layer_begin = llm->pipe.ranges()[stage].first;
    *layer_end = llm->pipe.ranges()[stage].second;
    return 0;
}

double ctamd_stage_issue_us(ctransformers_llm* llm, int stage, long long* evals) {
    if (evals) *evals = llm->pipe.issue_evals();
    return llm->pipe.issue_us(stage);
}

double ctamd_decode_burst(ctransformers_llm* llm, int n) {
    std::string err;
    double us = 0.0;
    if (llm->pipe.n_stages() != 1 || !llm->engine().decode_burst(n, &us, err)) { fprintf(stderr, "ctransformers_amd: burst failed: %s\n", err.c_str()); return -1.0; }
    return us;
}
long long ctamd_spec_hits(ctransformers_llm* llm, long long* launched) {
    if (launched) *launched = llm->engine().spec_launched();
    return llm->engine().spec_hits();
}
int ctamd_read_stamps(ctransformers_llm* llm, unsigned long long* out, int max) { return llm->engine().read_stamps(out, max); }
int ctamd_read_stamps_stage(ctransformers_llm* llm, int stage, unsigned long long* out, int max) { return llm->pipe.stage(stage).read_stamps(out, max); }
double ctamd_weight_bytes(ctransformers_llm* llm) { return (double)llm->engine().weight_bytes(); }
int ctamd_trace_site(ctransformers_llm* llm, const char* site, unsigned long long* out, int n) {
    std::string err;
    if (!llm->engine().trace_site(site, out, n, err)) { fprintf(stderr, "ctransformers_amd: trace failed: %s\n", err.c_str()); return -1; }
    return 0;
}

}  // extern "C"
