#include "pipeline.h"

#include <chrono>

#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <thread>
#include <string.h>

#include "gguf_reader.h"

namespace ctamd {

#ifndef CT_EMU
// The hand-off of rows [row0, row0 + n / E) from a stage to its successor: `src` is the producer's own stage buffer, `dst` the consumer's
// (peer-mapped: the stores travel over xGMI).  Every thread fences its stores at system scope; the last workgroup to arrive advances the
// boundary's sequence number and publishes it in the consumer's signal word, which the consumer's stream is waiting on.
// cur_dst (first micro-batch of a request; else null): the consumer's cursor {step, pos, n_past + n, batch} goes with the rows — one host-to-device copy
// per stage and request less in the consumer's stream (~5 us each: 35 us of an eight-stage token step).
__global__ void __launch_bounds__(256) handoff_rows_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int n4, unsigned* flag, unsigned* prod,
                                                           int* cur_dst, int c0, int c1, int c2, int c3, const int* cur_src) {
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < n4; i += (int)(gridDim.x * blockDim.x)) dst[i] = src[i];
    if (cur_dst && blockIdx.x == 0 && threadIdx.x == 0) {
        if (cur_src) {   // (inside the one-graph token step: the producer's cursor on its device, one token step further than the consumer's)
            cur_dst[0] = cur_src[0] - 1; cur_dst[1] = cur_src[1] - 1; cur_dst[2] = cur_src[2]; cur_dst[3] = cur_src[3];
        } else { cur_dst[0] = c0; cur_dst[1] = c1; cur_dst[2] = c2; cur_dst[3] = c3; }
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!flag) return;   // (hand-off form "event": the copy only; an event behind this kernel orders the next stage)
        const unsigned old = __hip_atomic_fetch_add(prod, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {
            __hip_atomic_store(prod, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned seq = prod[1] + 1u;   // (only ever touched here, one launch at a time: the stream serialises them)
            prod[1] = seq;
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
#endif

std::vector<std::pair<int, int>> partition_layers(const std::vector<double>& layer_bytes, double head_bytes, int n_stages) {
    const int L = (int)layer_bytes.size();
    std::vector<std::pair<int, int>> out;
    if (n_stages < 1 || L < n_stages) return out;
    double total = head_bytes;
    for (double b : layer_bytes) total += b;
    int begin = 0;
    double acc = 0.0;
    for (int s = 0; s < n_stages; ++s) {
        int end;
        if (s == n_stages - 1) {
            end = L;
        } else {
            const double target = (total - acc) / (n_stages - s);   // what each remaining stage should stream
            double got = 0.0;
            end = begin;
            while (end < L - (n_stages - 1 - s) && (end == begin || fabs(got + layer_bytes[end] - target) <= fabs(got - target))) got += layer_bytes[end++];
            acc += got;
        }
        out.emplace_back(begin, end);
        begin = end;
    }
    return out;
}

std::vector<int> parse_devices(const char* spec) {
    std::vector<int> d;
    if (!spec || !*spec) return {0};
    std::string s(spec);
    if (s.find(',') == std::string::npos) {
        const int n = atoi(s.c_str());
        for (int i = 0; i < std::max(1, n); ++i) d.push_back(i);
        return d;
    }
    size_t p = 0;
    while (p <= s.size()) {
        const size_t q = s.find(',', p);
        const std::string tok = s.substr(p, q == std::string::npos ? std::string::npos : q - p);
        if (!tok.empty()) d.push_back(atoi(tok.c_str()));
        if (q == std::string::npos) break;
        p = q + 1;
    }
    if (d.empty()) d.push_back(0);
    return d;
}

// The devices of a handle.  CT_AMD_DEVICES, when set, decides.  Otherwise `gpu_layers` and the visible GPUs do (north_star; reference
// knob models/llms/llama.cc:88-95 -> llama.cpp:1913-1919, where layers beyond n_gpu_layers stay on the CPU): gpu_layers sets the STAGE
// COUNT — ceil(n_layer / gpu_layers), at most one stage per visible device — and partition_layers then balances the stages by weight
// bytes (a stage may hold a layer more or less than gpu_layers; the last one also streams the head).  NOTE the consequence on a multi-GPU
// host: any 0 < gpu_layers < n_layer spreads the model (gpu_layers = 1 takes every visible GPU); gpu_layers <= 0 or >= n_layer (the usual
// "everything": 50, 100, 1000) is one GPU.  CT_AMD_DEVICES=0 pins a handle to one device regardless.  This library has no CPU path
// either way.  (include/ctransformers_llm.h documents the same.)
std::vector<int> plan_devices(const std::string& path, int gpu_layers, const char* env) {
    if (env && *env) return parse_devices(env);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 1 || gpu_layers <= 0) return {0};
    GgufFile f;
    std::string arch;
    uint32_t nl = 0;
    if (!f.open(path) || !f.get_str("general.architecture", arch) || !f.get_u32(arch + ".block_count", nl) || nl == 0) return {0};
    const int stages = std::min(ndev, ((int)nl + gpu_layers - 1) / gpu_layers);
    std::vector<int> d;
    for (int i = 0; i < std::max(1, stages); ++i) d.push_back(i);
    return d;
}

Pipeline::~Pipeline() {
    for (size_t s = 0; s < ev_.size(); ++s) {
        if (s < dev_.size()) (void)hipSetDevice(dev_[s]);
        for (hipEvent_t e : ev_[s]) (void)hipEventDestroy(e);
    }
#ifndef CT_EMU
    for (size_t s = 0; s < st_.size(); ++s) {   // the stages' streams may still reference the words: drain first
        (void)hipSetDevice(dev_[s]);
        (void)hipStreamSynchronize(st_[s]->stream());
    }
    for (size_t s = 0; s < flag_.size(); ++s) {
        if (flag_[s]) { (void)hipSetDevice(dev_[s + 1]); (void)hipFree(flag_[s]); }
        if (prod_[s]) { (void)hipSetDevice(dev_[s]); (void)hipFree(prod_[s]); }
    }
#endif
#ifndef CT_EMU
    if (step_graph_) (void)hipGraphExecDestroy(step_graph_);
#endif
    while (!st_.empty()) st_.pop_back();   // last stage first: a stage that shares the stream of the one before it goes before the stream's owner
}

bool Pipeline::load_gpt2(const std::string& path, std::string& err, bool starcoder) {
    st_.clear();
    st_.emplace_back(new Engine());
    dev_ = {0};
    ranges_.clear();
    return st_[0]->load_gpt2(path, err, 0, starcoder);
}

bool Pipeline::load_mpt(const std::string& path, int context_length, std::string& err) {
    st_.clear();
    st_.emplace_back(new Engine());
    dev_ = {0};
    ranges_.clear();
    return st_[0]->load_mpt(path, context_length, err);
}

bool Pipeline::load_stage(const std::string& path, int context_length, int layer_begin, int layer_end, int device, std::string& err) {
    st_.clear();
    st_.emplace_back(new Engine());
    dev_ = {device};
    ranges_.clear();
    return st_[0]->load(path, context_length, 1000, err, layer_begin, layer_end, device);
}

bool Pipeline::load(const std::string& path, int context_length, int gpu_layers, const std::vector<int>& devices, std::string& err) {
    st_.clear();
    dev_ = devices.empty() ? std::vector<int>{0} : devices;
    if (dev_.size() == 1) {
        st_.emplace_back(new Engine());
        ranges_.clear();
        return st_[0]->load(path, context_length, gpu_layers, err, -1, -1, dev_[0]);
    }
    // per-layer bytes from the tensor table of the file (nothing is uploaded here)
    std::vector<double> layer_bytes;
    double head_bytes = 0.0;
    {
        GgufFile f;
        if (!f.open(path)) { err = f.error(); return false; }
        std::string arch;
        uint32_t nl = 0;
        if (!f.get_str("general.architecture", arch) || !f.get_u32(arch + ".block_count", nl) || nl == 0) { err = "block_count missing"; return false; }
        layer_bytes.assign(nl, 0.0);
        for (const GgufTensor& t : f.tensors()) {
            if (t.name.compare(0, 4, "blk.") == 0) {
                const int i = atoi(t.name.c_str() + 4);
                if (i >= 0 && i < (int)nl) layer_bytes[i] += (double)t.nbytes;
            } else if (t.name == "output.weight") {
                head_bytes += (double)t.nbytes;
            }
        }
    }
    if ((int)layer_bytes.size() < (int)dev_.size()) { err = "more pipeline stages than layers"; return false; }
    ranges_ = partition_layers(layer_bytes, head_bytes, (int)dev_.size());
    if (ranges_.size() != dev_.size()) { err = "layer partitioning failed"; return false; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        err = "no HIP device visible: this library runs on MI355X only and has no CPU fallback";
        return false;
    }
    for (int d : dev_)
        if (d < 0 || d >= ndev) { err = "CT_AMD_DEVICES names device " + std::to_string(d) + " but " + std::to_string(ndev) + " are visible"; return false; }
    // the stages load at the same time, one host thread each (its own device, stream, pinned slots and byte range of the file);
    // stages that share a device (the 1-GPU test form "0,0") load one after the other
    for (size_t s = 0; s < dev_.size(); ++s) st_.emplace_back(new Engine());
    std::vector<std::string> errs(dev_.size());
    std::vector<char> oks(dev_.size(), 0);
    bool distinct = true;
    for (size_t s = 0; s < dev_.size(); ++s)
        for (size_t t = s + 1; t < dev_.size(); ++t) distinct = distinct && dev_[s] != dev_[t];
    auto load_one = [&](size_t s) {   // runs on its own thread: nothing may escape it (an exception there would be std::terminate, not a NULL handle)
        try {
            oks[s] = st_[s]->load(path, context_length, gpu_layers, errs[s], ranges_[s].first, ranges_[s].second, dev_[s]) ? 1 : 0;
        } catch (const std::exception& e) {
            errs[s] = std::string("exception while loading: ") + e.what();
            oks[s] = 0;
        } catch (...) {
            errs[s] = "unknown exception while loading";
            oks[s] = 0;
        }
    };
    if (distinct && kConcurrentLaunches) {
        std::vector<std::thread> th;
        for (size_t s = 0; s < dev_.size(); ++s) th.emplace_back(load_one, s);
        for (auto& t : th) t.join();
    } else {
        for (size_t s = 0; s < dev_.size(); ++s) load_one(s);
    }
    for (size_t s = 0; s < dev_.size(); ++s) {   // every failed stage is named (the stages that did load are released with the handle)
        if (!oks[s]) {
            if (!err.empty()) err += "; ";
            err += "stage " + std::to_string(s) + " (layers " + std::to_string(ranges_[s].first) + ".." + std::to_string(ranges_[s].second) + " on device " +
                   std::to_string(dev_[s]) + "): " + errs[s];
        }
    }
    if (!err.empty()) return false;
    // direct peer stores where the link allows them AND the mapping was really established (hipMemcpyPeerAsync works either way): a boundary whose
    // hipDeviceEnablePeerAccess failed for another reason than "already enabled" keeps the copy + event form — a store into an unmapped peer
    // buffer would be a GPU fault, not a fallback
    std::vector<char> peer_ok(dev_.size() > 1 ? dev_.size() - 1 : 0, 0);
    for (size_t s = 0; s + 1 < dev_.size(); ++s) {
        if (dev_[s] == dev_[s + 1]) { peer_ok[s] = 1; continue; }
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dev_[s], dev_[s + 1]) == hipSuccess && can) {
            (void)hipSetDevice(dev_[s]);
            const hipError_t pe = hipDeviceEnablePeerAccess(dev_[s + 1], 0);
#ifndef CT_EMU
            peer_ok[s] = pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled;
#else
            peer_ok[s] = pe == hipSuccess;
#endif
            (void)hipGetLastError();
        }
    }
    ev_.assign(dev_.size(), {});
    // Consecutive stages on ONE device (the 1-GPU test form "0,0", or more stages than GPUs) share a stream: the hand-off between them is stream order —
    // no event record + cross-stream wait (~22 us per hop on a shared device, profiles/r05_pipeline_handoff.txt).  CT_AMD_PP_SHARED_STREAM=0: one stream per stage.
    {
        const char* sh = getenv("CT_AMD_PP_SHARED_STREAM");
        const bool share = !(sh && *sh == '0');
        for (size_t s = 1; share && s < dev_.size(); ++s) {
            if (dev_[s] != dev_[s - 1]) continue;
            (void)hipSetDevice(dev_[s]);
            if (!st_[s]->adopt_stream(st_[s - 1]->stream())) { err = "pipeline: sharing the stream of stage " + std::to_string(s - 1) + " failed"; return false; }
        }
    }
#ifndef CT_EMU
    // hand-off words of the in-stream form (pipeline.h): signal memory for the consumer's stream wait, on the consumer's device; the
    // producer needs the direct peer mapping (a stage pair without it keeps the copy + event form, and so does CT_AMD_HANDOFF=event)
    {
        // Default: "flag" where every stage has its own device; stages that SHARE a device share a stream (above) — given a stream each
        // (CT_AMD_PP_SHARED_STREAM=0) they hand over by events: the
        // runtime serves a pending hipStreamWaitValue32 with a polling wave on that device, which takes a CU away from the producer stage's
        // one-workgroup-per-CU kernels (+1.2 us per launch, measured) and breaks the residency of its fused QKV + attention launch.
        // CT_AMD_HANDOFF=flag / event forces either.
        const char* hm = getenv("CT_AMD_HANDOFF");
        bool distinct = true;
        for (size_t s = 0; s < dev_.size(); ++s)
            for (size_t t = s + 1; t < dev_.size(); ++t) distinct = distinct && dev_[s] != dev_[t];
        flag_mode_ = hm && *hm ? !strcmp(hm, "flag") : distinct;
        const bool rows_of_float4 = st_[0]->hparams().n_embd % 4 == 0;   // handoff_rows_kernel moves 16-byte pieces
        if (!rows_of_float4) flag_mode_ = false;
        for (size_t s = 0; flag_mode_ && s + 1 < dev_.size(); ++s)
            if (!peer_ok[s]) flag_mode_ = false;
        direct_.assign(dev_.size() - 1, 0);
        for (size_t s = 0; s + 1 < dev_.size(); ++s) direct_[s] = peer_ok[s] && rows_of_float4 ? 1 : 0;
        flag_.assign(dev_.size() - 1, nullptr);
        prod_.assign(dev_.size() - 1, nullptr);
        issued_.assign(dev_.size() - 1, 0u);
        for (size_t s = 0; flag_mode_ && s + 1 < dev_.size(); ++s) {
            (void)hipSetDevice(dev_[s + 1]);
            if (hipExtMallocWithFlags((void**)&flag_[s], 8, hipMallocSignalMemory) != hipSuccess) { flag_[s] = nullptr; flag_mode_ = false; (void)hipGetLastError(); break; }
            (void)hipMemset(flag_[s], 0, 8);
            (void)hipSetDevice(dev_[s]);
            if (hipMalloc((void**)&prod_[s], 256) != hipSuccess) { prod_[s] = nullptr; flag_mode_ = false; (void)hipGetLastError(); break; }
            (void)hipMemset(prod_[s], 0, 256);
        }
        for (size_t s = 0; s < dev_.size(); ++s) { (void)hipSetDevice(dev_[s]); (void)hipDeviceSynchronize(); }
        // First contact (the flag form has never met two physical devices in this project's history): one row of known values through every boundary —
        // peer-mapped stores, system-scope sequence word, the consumer stream's wait — before a user's eval depends on it.  A boundary that does not
        // deliver within two seconds, or delivers other bytes, sends the whole pipeline back to the copy + event form (CT_AMD_HANDOFF=flag does not
        // override a failed check).
        if (flag_mode_ && !handoff_self_check()) {
            fprintf(stderr, "ctransformers_amd: the in-stream hand-off (peer stores + stream wait) failed its self-check; stages hand over by copy + event\n");
            flag_mode_ = false;
        }
        // A stage that WAITS for a hand-off has the runtime's polling wave on its device while it waits: one CU less than the grid of the fused
        // QKV + attention launch and of the shared score rows needs (they assume every workgroup resident: kernels_qa9.h) — their sweeps would time out
        // and the eval be replayed (Engine::resident_timeout).  Decided here, once: consumer stages of the flag form run the forms that need no residency;
        // stage 0 (no wait on its device) keeps them.
        for (size_t s = 1; flag_mode_ && s < st_.size(); ++s)
            if (dev_[s] != dev_[s - 1]) st_[s]->disable_resident_forms();
    }
#endif
    // Tokens per micro-batch of a prompt.  A (stage, micro-batch) unit of a 7B costs 1.96 / 2.6 / 4.0 ms x 2 / stages at 32 / 64 / 128
    // tokens (bench.py --gpus 2 on one GPU, `tok_s_by_micro_batch`: fewer passes over the weights with larger ones), a 128-token
    // prompt takes (micro-batches + stages - 1) units: 64 tokens win up to four stages, 32 beyond.  CT_AMD_PP_MB overrides.
    micro_batch_ = st_.size() <= 4 ? 64 : 32;
    const char* mb = getenv("CT_AMD_PP_MB");
    if (mb && atoi(mb) > 0) micro_batch_ = atoi(mb);
    return true;
}

#ifndef CT_EMU
bool Pipeline::handoff_self_check() {
    const int E = st_[0]->hparams().n_embd;
    std::vector<float> pat((size_t)E), back((size_t)E);
    const char* dbg_fail = getenv("CT_AMD_DBG_HANDOFF_FAIL");   // tests: the check runs and then reports a mismatch
    for (size_t s = 0; s + 1 < st_.size(); ++s) {
        if (shares_stream((int)s)) continue;   // (stream order: no flag on this boundary)
        for (int i = 0; i < E; ++i) pat[(size_t)i] = (float)((i * 2654435761u + 97u * (unsigned)s) & 0xFFFFFF) * (1.0f / 4096.0f) - 1024.0f;
        if (hipSetDevice(dev_[s]) != hipSuccess || hipMemcpy(st_[s]->xio(), pat.data(), (size_t)E * 4, hipMemcpyHostToDevice) != hipSuccess) return false;
        if (hipSetDevice(dev_[s + 1]) != hipSuccess || hipMemset(st_[s + 1]->xio(), 0, (size_t)E * 4) != hipSuccess) return false;
        ++issued_[s];
        // producer first, as eval_stages queues them (two streams of ONE device may share a hardware queue: a wait queued ahead of the kernel it waits
        // for would never be released); then the consumer's stream waits for the sequence number and brings the row back
        (void)hipSetDevice(dev_[s]);
        hipLaunchKernelGGL(handoff_rows_kernel, dim3(1), dim3(256), 0, st_[s]->stream(), (const float4*)st_[s]->xio(), (float4*)st_[s + 1]->xio(), E / 4, flag_[s], prod_[s],
                           (int*)nullptr, 0, 0, 0, 0, (const int*)nullptr);
        (void)hipSetDevice(dev_[s + 1]);
        if (hipStreamWaitValue32(st_[s + 1]->stream(), flag_[s], issued_[s], hipStreamWaitValueGte, 0xFFFFFFFFu) != hipSuccess) return false;
        if (hipMemcpyAsync(back.data(), st_[s + 1]->xio(), (size_t)E * 4, hipMemcpyDeviceToHost, st_[s + 1]->stream()) != hipSuccess) return false;
        const auto t0 = std::chrono::steady_clock::now();
        bool done = false;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.0) {
            if (hipStreamQuery(st_[s + 1]->stream()) == hipSuccess) { done = true; break; }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        (void)hipGetLastError();
        if (!done) {   // release the waiting stream by hand, then give up on the form
            (void)hipMemcpy(flag_[s], &issued_[s], 4, hipMemcpyHostToDevice);
            (void)hipStreamSynchronize(st_[s + 1]->stream());
            return false;
        }
        if (memcmp(pat.data(), back.data(), (size_t)E * 4) != 0) return false;
    }
    return !(dbg_fail && *dbg_fail == '1');
}
#endif

#define PIPE_OK(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            err = std::string(#expr) + " failed: " + hipGetErrorString(e_);                                 \
            return false;                                                                                    \
        }                                                                                                    \
    } while (0)

bool Pipeline::eval(const int* tokens, int n, int n_past, std::string& err, int batch) {
    if (st_.size() == 1) return st_[0]->eval(tokens, n, n_past, err, batch);
    if (n <= 0) return true;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (eval_stages(tokens, n, n_past, err, batch)) return true;
        drain_after_failure();
        // A stage whose resident launch forms gave up (Engine::resident_timeout: the device is shared) has switched them off; eval_stages stopped at the
        // first such stage, so every stage is asked — their words would otherwise fail the next evals one after another — and the request is evaluated
        // once more (every KV row it wrote is written again).  The reference never fails an eval for this reason (models/llm.h:40-54).
        bool lost = false;
        for (auto& st : st_) lost = st->resident_timeout() || lost;
        if (!lost || attempt == 1) return false;
#ifndef CT_EMU
        if (step_graph_) { (void)hipGraphExecDestroy(step_graph_); step_graph_ = nullptr; }
#endif
        err.clear();
    }
    return false;
}

// a host-side failure in the middle of a request
void Pipeline::drain_after_failure() {
    std::lock_guard<std::recursive_mutex> legacy(capture_mutex());   // (the releases below are copies on the legacy stream)
    // a host-side failure in the middle of a request: the stages already fed keep running and writing into the next stage's hand-off
    // buffer and KV cache — drain every stream before the caller sees the error, so that a retry does not overlap stale work
#ifndef CT_EMU
    if (flag_mode_) {   // a stage may be waiting for a hand-off that was never queued: release it (the request is lost anyway), keep the sequence in step
        for (size_t s = 0; s + 1 < st_.size(); ++s) {
            (void)hipSetDevice(dev_[s]);
            (void)hipStreamSynchronize(st_[s]->stream());
            const unsigned v[2] = {0u, issued_[s]};
            (void)hipMemcpy(prod_[s], v, 8, hipMemcpyHostToDevice);
            (void)hipSetDevice(dev_[s + 1]);
            (void)hipMemcpy(flag_[s], &issued_[s], 4, hipMemcpyHostToDevice);
        }
    }
#endif
    for (size_t s = 0; s < st_.size(); ++s) {
        (void)hipSetDevice(dev_[s]);
        (void)hipStreamSynchronize(st_[s]->stream());
    }
}

// A decode step of stages that share one stream: see step_graph_ (pipeline.h).  taken = false: the caller goes the per-stage way.
bool Pipeline::eval_one_graph(const int* tokens, int n_past, std::string& err, int batch, bool& taken) {
    taken = false;
#ifndef CT_EMU
    const int S = (int)st_.size(), E = st_[0]->hparams().n_embd;
    if (step_graph_off_ || flag_mode_) return true;
    for (int s = 0; s < S; ++s) {
        if (!st_[s]->uses_graphs()) return true;
        if (s + 1 < S && (!shares_stream(s) || !direct_[s])) return true;
    }
    {   // CT_AMD_PP_ONE_GRAPH=0: one graph per stage (the A/B partner); read per call so that a test can switch it between handles
        const char* og = getenv("CT_AMD_PP_ONE_GRAPH");
        if (og && *og == '0') return true;
    }
    taken = true;
    hipStream_t stream = st_[0]->stream();
    PIPE_OK(hipSetDevice(dev_[0]));
    if ((int)issue_us_.size() != S) issue_us_.assign(S, 0.0);
    ++issue_evals_;
    const auto t_issue = std::chrono::steady_clock::now();
    for (int s = 0; s < S; ++s)
        if (!st_[s]->req_begin(tokens, 1, n_past, batch, err, s == 0)) return false;
    if (!step_graph_) {
        hipGraph_t g = nullptr;
        std::unique_lock<std::recursive_mutex> cap(capture_mutex());
        PIPE_OK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        bool ok = true;
        for (int s = 0; s < S && ok; ++s) {
            ok = st_[s]->capture_step(s == S - 1, err);
            if (ok && s + 1 < S)
                hipLaunchKernelGGL(handoff_rows_kernel, dim3((unsigned)std::max(1, std::min(256, (E / 4 + 255) / 256))), dim3(256), 0, stream, (const float4*)st_[s]->xio(),
                                   (float4*)st_[s + 1]->xio(), E / 4, (unsigned*)nullptr, (unsigned*)nullptr, st_[s + 1]->state_dev(), 0, 0, 0, 0, (const int*)st_[s]->state_dev());
        }
        hipError_t e = hipStreamEndCapture(stream, &g);
        cap.unlock();
        if (!ok || e != hipSuccess) {
            if (g) (void)hipGraphDestroy(g);
            if (ok) err = std::string("hipStreamEndCapture (pipeline step) failed: ") + hipGetErrorString(e);
            step_graph_off_ = true;   // the per-stage way from now on
            return false;
        }
        hipError_t ei = hipGraphInstantiate(&step_graph_, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ei != hipSuccess) { step_graph_ = nullptr; step_graph_off_ = true; err = std::string("hipGraphInstantiate (pipeline step) failed: ") + hipGetErrorString(ei); return false; }
    }
    PIPE_OK(hipGraphLaunch(step_graph_, stream));
    issue_us_[0] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_issue).count();
    if (!st_[S - 1]->req_logits(err)) return false;
    for (int s = S - 1; s >= 0; --s) {
        if (!st_[s]->req_wait(1, n_past, err)) {   // (a stage changed its launch forms: the step is captured again)
            (void)hipGraphExecDestroy(step_graph_);
            step_graph_ = nullptr;
            return false;
        }
    }
#else
    (void)tokens; (void)n_past; (void)err; (void)batch;
#endif
    return true;
}

bool Pipeline::eval_stages(const int* tokens, int n, int n_past, std::string& err, int batch) {
    const int S = (int)st_.size(), E = st_[0]->hparams().n_embd;
    if (n == 1) {
        bool taken = false;
        const bool ok = eval_one_graph(tokens, n_past, err, batch, taken);
        if (taken || !ok) return ok;
    }
    // (each stage's cursor + token ids go out right before its first range: stage 0 starts on the GPU while the host is still queueing the others)
    // micro-batches: every stage gets its ranges in order; stage s + 1's stream waits for the event behind stage s's copy
    // (stages that all share one stream cannot overlap: micro-batches would only add passes over the weights)
    bool one_stream = true;
    for (int s = 0; s + 1 < S; ++s) one_stream = one_stream && shares_stream(s);
    const int mb = n == 1 ? 1 : (one_stream && !getenv("CT_AMD_PP_MB") ? n : std::max(2, micro_batch_));
    const int n_mb = (n + mb - 1) / mb;
    for (int s = 0; s + 1 < S && !flag_mode_; ++s) {
        PIPE_OK(hipSetDevice(dev_[s]));
        while ((int)ev_[s].size() < n_mb) {
            hipEvent_t e;
            PIPE_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ev_[s].push_back(e);
        }
    }
    if ((int)issue_us_.size() != S) issue_us_.assign(S, 0.0);
    ++issue_evals_;
    for (int k = 0; k < n_mb; ++k) {
        const int c0 = k * mb, nt = std::min(mb, n - c0);
        const bool last_mb = k == n_mb - 1;
        for (int s = 0; s < S; ++s) {
            Engine& st = *st_[s];
            const auto t_issue = std::chrono::steady_clock::now();
            struct Acc { double& d; std::chrono::steady_clock::time_point t0; ~Acc() { d += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); } } acc{issue_us_[s], t_issue};
            PIPE_OK(hipSetDevice(dev_[s]));
            // (a stage behind a direct boundary gets its cursor from the hand-off kernel of the stage before it; stages > 0 use no token ids)
            bool cur_by_kernel = false;
#ifndef CT_EMU
            cur_by_kernel = s > 0 && (flag_mode_ || direct_[s - 1]);
#endif
            if (k == 0 && !st.req_begin(tokens, n, n_past, batch, err, !cur_by_kernel)) return false;
            int* cur_next = nullptr;
            const int* cq = st.req_cursor();
#ifndef CT_EMU
            if (k == 0 && s + 1 < S && (flag_mode_ || direct_[s])) cur_next = st_[s + 1]->state_dev();
#endif
            (void)cur_next; (void)cq;
#ifndef CT_EMU
            if (flag_mode_) {
                // stage s's stream waits (its command processor polls) for the sequence number the (k + 1)-th ... hand-off of boundary s - 1 publishes
                if (s > 0 && !shares_stream(s - 1)) PIPE_OK(hipStreamWaitValue32(st.stream(), flag_[s - 1], issued_[s - 1], hipStreamWaitValueGte, 0xFFFFFFFFu));
                if (!st.req_range(c0, nt, last_mb, err)) return false;
                if (s + 1 < S) {
                    const size_t off = (size_t)c0 * E;
                    const int n4 = nt * E / 4;
                    const int gx = std::max(1, std::min(256, (n4 + 255) / 256));
                    ++issued_[s];
                    hipLaunchKernelGGL(handoff_rows_kernel, dim3((unsigned)gx), dim3(256), 0, st.stream(), (const float4*)(st.xio() + off),
                                       (float4*)(st_[s + 1]->xio() + off), n4, flag_[s], prod_[s], cur_next, cq[0], cq[1], cq[2], cq[3], (const int*)nullptr);
                }
                continue;
            }
#endif
            if (s > 0 && !shares_stream(s - 1)) PIPE_OK(hipStreamWaitEvent(st.stream(), ev_[s - 1][k], 0));   // (a shared stream orders the stages by itself)
            if (!st.req_range(c0, nt, last_mb, err)) return false;
            if (s + 1 < S) {
                const size_t off = (size_t)c0 * E;
#ifndef CT_EMU
                if (direct_[s]) {   // the rows go straight into the next stage's buffer from a kernel (no copy engine: ~3 us less per hop)
                    const int n4 = nt * E / 4;
                    hipLaunchKernelGGL(handoff_rows_kernel, dim3((unsigned)std::max(1, std::min(256, (n4 + 255) / 256))), dim3(256), 0, st.stream(),
                                       (const float4*)(st.xio() + off), (float4*)(st_[s + 1]->xio() + off), n4, (unsigned*)nullptr, (unsigned*)nullptr, cur_next, cq[0], cq[1], cq[2], cq[3], (const int*)nullptr);
                } else
#endif
                PIPE_OK(hipMemcpyPeerAsync(st_[s + 1]->xio() + off, dev_[s + 1], st.xio() + off, dev_[s], (size_t)nt * E * sizeof(float), st.stream()));
                if (!shares_stream(s)) PIPE_OK(hipEventRecord(ev_[s][k], st.stream()));
            }
        }
    }
    if (!st_[S - 1]->req_logits(err)) return false;
    for (int s = S - 1; s >= 0; --s)   // the last stage's stream drains last in time: wait for it first, the others are then idle
        if (!st_[s]->req_wait(n, n_past, err)) return false;
    return true;
}

}  // namespace ctamd
