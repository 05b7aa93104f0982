// In-process layer pipeline over the GPUs of one node: what `ctransformers_llm_create` builds when
// CT_AMD_DEVICES names more than one device.  One Engine stage per device, each owning a contiguous block of layers (weights +
// that block's KV cache resident on its GPU; token embedding on stage 0, final norm + lm_head on the last stage).  The only
// exchange is the [n_tokens][n_embd] f32 residual stream from stage s to stage s + 1, IN-STREAM (round 5): a kernel on the producer's
// stream stores the rows straight into the consumer's peer-mapped hand-off buffer over the direct xGMI link and publishes a sequence
// number behind them (system-scope release); the consumer's STREAM waits for that number with hipStreamWaitValue32 — its command
// processor polls a word in its own HBM, no CU is occupied, no event, no SDMA copy (5 us per hop against 17 for the peer copy + event
// pair of round 4: tools/experiments/handoff_probe.cpp, profiles/r05_handoff_probe.txt; CT_AMD_HANDOFF=event keeps the old form for
// A/B).  Consecutive stages on ONE device (more stages than GPUs; the 1-GPU test form) share ONE stream: stream order is their hand-off, and a decode step of a
// pipeline whose stages all share a stream is ONE graph (eval_one_graph).  No host round trip, no collective, no torch.  A prompt is cut into micro-batches so that stage s works on micro-batch c while stage s - 1 already works on c + 1;
// results do not depend on the cut (the cursor carries the reference's batch structure, c_api.cc).
// What the reference does instead: `gpu_layers` / `tensor_split` split tensors inside one process with peer copies per mat-mul
// (reference models/ggml/llama.cpp:1913-1919, :1938-2070; ggml-cuda.cu:5798-6119).
#pragma once
#include <algorithm>
#include <memory>
#include <string>
#include <vector>

#include "engine.h"

namespace ctamd {

// Contiguous layer ranges, one per stage, balancing the per-token HBM bytes (`layer_bytes[i]`; the last stage also streams
// `head_bytes`).  Pure host logic (tested without a GPU).
std::vector<std::pair<int, int>> partition_layers(const std::vector<double>& layer_bytes, double head_bytes, int n_stages);

// "4" -> {0,1,2,3}; "0,2" -> {0,2}; "0,0" -> two stages on device 0 (the 1-GPU test form).  Empty / unset -> {0}.
std::vector<int> parse_devices(const char* spec);
// The devices of a handle: `env` (CT_AMD_DEVICES) when set, else ceil(n_layer / gpu_layers) stages over the visible GPUs (pipeline.cc).
std::vector<int> plan_devices(const std::string& path, int gpu_layers, const char* env);

class Pipeline {
   public:
    // One stage per entry of `devices`.  A single entry is the plain single-GPU engine (no pipeline machinery on its path).
    bool load(const std::string& path, int context_length, int gpu_layers, const std::vector<int>& devices, std::string& err);
    bool load_gpt2(const std::string& path, std::string& err, bool starcoder = false);
    bool load_mpt(const std::string& path, int context_length, std::string& err);
    // one explicit stage (ctamd_stage_create: the multi-process pipeline of tools/rccl_pipeline.py drives it from outside)
    bool load_stage(const std::string& path, int context_length, int layer_begin, int layer_end, int device, std::string& err);
    bool eval(const int* tokens, int n, int n_past, std::string& err, int batch = 0);
    const std::vector<int>& devices() const { return dev_; }
    bool coalesces_batches() const { return st_.front()->coalesces_batches() || st_.size() > 1; }
    Engine& first() { return *st_.front(); }
    Engine& last() { return *st_.back(); }
    Engine& stage(int s) { return *st_[(size_t)std::max(0, std::min(s, (int)st_.size() - 1))]; }
    int n_stages() const { return (int)st_.size(); }
    const std::vector<std::pair<int, int>>& ranges() const { return ranges_; }
    // host time (us) the ONE issuing thread has spent queueing stage s's launches, waits and copies so far, and the evals counted
    // (ctamd_stage_issue_us: bench.py --gpus N prints it — on a real N-GPU node it shows whether that thread binds before the GPUs do)
    double issue_us(int s) const { return s >= 0 && s < (int)issue_us_.size() ? issue_us_[s] : 0.0; }
    long long issue_evals() const { return issue_evals_; }
    ~Pipeline();

   private:
    bool eval_stages(const int* tokens, int n, int n_past, std::string& err, int batch);
    bool handoff_self_check();    // load time, hand-off form "flag" on distinct devices: a known row through every boundary (false: use copy + event)
    void drain_after_failure();   // every stage's stream idle (and a pending flag hand-off released) before the caller sees an error or the request is replayed
    std::vector<std::unique_ptr<Engine>> st_;
    std::vector<int> dev_;
    std::vector<std::pair<int, int>> ranges_;
    std::vector<std::vector<hipEvent_t>> ev_;   // ev_[s][k]: micro-batch k's rows have left stage s (hand-off form "event")
    // hand-off form "flag": per boundary s -> s + 1 a sequence word in signal memory on the CONSUMER's device, the producer's arrival counter
    // + running sequence number on the producer's device, and the number of hand-offs issued so far (what the consumer's stream waits for)
    std::vector<unsigned*> flag_;
    std::vector<unsigned*> prod_;
    std::vector<unsigned> issued_;
    std::vector<char> direct_;   // boundary s -> s + 1: the producer can store into the consumer's buffer (same device or peer access)
    bool flag_mode_ = false;
    // Stages that all share one device and one stream: the token steps of all stages and the hand-offs between them as ONE graph — a decode step is one
    // host-to-device copy (stage 0's cursor + token id) and one graph launch, whatever the stage count (graph launches follow each other ~11 us apart).
#ifndef CT_EMU
    hipGraphExec_t step_graph_ = nullptr;
#endif
    bool step_graph_off_ = false;
    bool eval_one_graph(const int* tokens, int n_past, std::string& err, int batch, bool& taken);
   public:
    // "stream": every boundary lies between stages that share one device and one stream (nothing to wait for); else the form of the cross-stream boundaries
    // boundary s -> s + 1 lies inside one stream (the two stages sit on one device and queue on the same stream: load())
    bool shares_stream(int s) const { return dev_[(size_t)s] == dev_[(size_t)s + 1] && st_[(size_t)s]->stream() == st_[(size_t)s + 1]->stream(); }
    const char* handoff() const {
        if (st_.size() < 2) return "none";
        bool all_shared = true;
        for (size_t s = 0; s + 1 < st_.size(); ++s) all_shared = all_shared && shares_stream((int)s);
        return all_shared ? "stream" : (flag_mode_ ? "flag" : "event");
    }
   private:
    int micro_batch_ = 32;
    std::vector<double> issue_us_;
    long long issue_evals_ = 0;
};

}  // namespace ctamd
