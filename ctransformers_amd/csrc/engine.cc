#include "engine.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include <thread>
#include <atomic>
#include <chrono>
#include <unistd.h>

#include "gguf_reader.h"
#include "kernels_v9.h"
#include "kernels_attn9.h"
#include "kernels_qa9.h"
#include "kernels_q32.h"
#include "kernels_pf.h"
#include "kernels_pg.h"
#include "kernels_raw32.h"
#include "kernels_mm8.h"

namespace ctamd {

#define HIP_OK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            err = std::string(#expr) + " failed: " + hipGetErrorString(e_);                                 \
            return false;                                                                                    \
        }                                                                                                    \
    } while (0)

// Prompt chunks: the order-free kernels (kernels_mm8.h) unless CT_AMD_PREFILL=exact asks for the bit-identical chunk kernels
constexpr bool kPrefillFastDefault = false;

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

std::recursive_mutex& capture_mutex() {
    static std::recursive_mutex m;
    return m;
}

Engine::~Engine() { free_all(); }

bool Engine::adopt_stream(hipStream_t s) {
    if (!s || s == stream_) return true;
    if (stream_) {
        if (hipStreamSynchronize(stream_) != hipSuccess) return false;
        if (stream_owned_) (void)hipStreamDestroy(stream_);
    }
    stream_ = s;
    stream_owned_ = false;
    return true;
}

void Engine::free_all() {
    release_staged();
    if (stream_ || !dev_allocs_.empty()) (void)hipSetDevice(device_);
    for (void* p : dev_allocs_) (void)hipFree(p);
    dev_allocs_.clear();
    if (h_logits_) (void)hipHostFree(h_logits_);   // h_emb_ lives in the same pinned block
    if (h_scalars_) (void)hipHostFree(h_scalars_);
    h_logits_ = h_emb_ = nullptr;
    h_scalars_ = nullptr;
#ifndef CT_EMU
    if (stream_) (void)hipStreamSynchronize(stream_);   // a speculative continuation step may still be running
    spec_inflight_ = false;
    if (graph_step_) (void)hipGraphExecDestroy(graph_step_);
    if (graph_step_head_) (void)hipGraphExecDestroy(graph_step_head_);
    graph_step_ = graph_step_head_ = nullptr;
    for (int b = 0; b < 2; ++b) {
        if (graph_cont_[b]) (void)hipGraphExecDestroy(graph_cont_[b]);
        if (ev_step_[b]) (void)hipEventDestroy(ev_step_[b]);
        graph_cont_[b] = nullptr;
        ev_step_[b] = nullptr;
    }
    for (auto& kv : chunk_graphs_) (void)hipGraphExecDestroy(kv.second.exec);
    chunk_graphs_.clear();
    for (hipGraphExec_t g : retired_graphs_) (void)hipGraphExecDestroy(g);
    retired_graphs_.clear();
#endif
    if (stream_ && stream_owned_) (void)hipStreamDestroy(stream_);
    stream_ = nullptr;
}

template <class T> static bool dev_alloc(std::vector<void*>& pool, T** out, size_t n_elems, std::string& err) {
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(n_elems * sizeof(T), 256);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        err = std::string("hipMalloc(") + std::to_string(bytes) + ") failed: " + hipGetErrorString(e);
        return false;
    }
    pool.push_back(p);
    *out = (T*)p;
    return true;
}

#include "engine_load.h"

// ---------------------------------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------------------------------
// Dynamic-LDS opt-in of a kernel (more than 64 KB per workgroup), once per kernel AND device: the attribute belongs to the code
// object loaded on the current device, and the in-process pipeline (pipeline.cc) launches the same kernels on several devices.
static int current_device() {
    int d = 0;
#ifndef CT_EMU
    (void)hipGetDevice(&d);
#endif
    return d >= 0 && d < 16 ? d : 0;
}
#define CT_OPTIN_ONCE(fn, bytes) \
    do { static bool done_[16] = {}; const int dv_ = current_device(); if (!done_[dv_]) { done_[dv_] = true; (void)CT_SMEM_OPTIN(fn, bytes); } } while (0)

static int chip_cus() {   // CUs of the current device (cached per device)
    static int n_cu[16] = {};
    const int d = current_device();
#ifdef CT_EMU
    n_cu[d] = 0;   // the emulated chip's CU count may change between tests of one process (CT_EMU_CUS)
#endif
    if (n_cu[d] == 0) {
        int n = 0;
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d);
        n_cu[d] = n > 0 ? n : 256;
    }
    return n_cu[d];
}

// K-quant decode mat-vec (kernels_v9.h): every job a K-quant matrix with a LAYOUT_L9 arena (gate/up: ONE job, the fused matrix).
// Units are row pairs; the jobs' units are concatenated, type group A first.
static bool kq_can(const MatvecArgs& a) {
    if (a.njobs < 1 || a.njobs > 3 || a.K > 32768) return false;
    if (a.gateup) return a.njobs == 1 && a.job[0].w.r9 && (a.job[0].w.layout == LAYOUT_R2C4 || a.job[0].w.layout == LAYOUT_L9);
    const bool b32 = is_block32(a.job[0].w.type);
    for (int j = 0; j < a.njobs; ++j) {
        const int t = a.job[j].w.type;
        if (!a.job[j].w.r9 || !(is_kquant(t) || is_block32(t))) return false;
        if (is_block32(t) != b32 || (b32 && t != a.job[0].w.type)) return false;   // the 32-block types: one type per launch
    }
    return true;
}

static long long g_kq_launches = 0;   // test hook (ctamd_kq_launches): K-quant decode mat-vec launches (kernels_v9.h) of this process
long long kq_launches() { return g_kq_launches; }
static long long g_pg_launches = 0;   // test hook (ctamd_pg_launches): chunk launches on the f16 matrix cores (kernels_pg.h)
long long pg_launches() { return g_pg_launches; }
static long long g_mm8_launches = 0;  // test hook (ctamd_mm8_launches): chunk launches of the order-free kernels (kernels_mm8.h)
long long mm8_launches() { return g_mm8_launches; }
static long long g_attn_free_launches = 0;   // test hook (ctamd_attn_free_launches): order-free decode attention launches (kernels_attn9.h)
long long attn_free_launches() { return g_attn_free_launches; }

// Two-type launch: how many of a workgroup's sixteen waves walk type group B.  A launch ends with its slowest wave, and a wave's time goes with the
// bytes of its units: the split with the smallest LARGEST per-wave byte count, ties to the split nearest the groups' byte shares.  (Round 5: the split
// by byte shares alone gave the 7B's QKV launch 9 + 7 waves — one V wave per workgroup with two Q6_K units, 13.4 KB against 9.2 KB for every other
// wave, and that wave ended the mat-vec phase of the fused launch 2 500 cycles after the rest; 8 + 8 leaves no wave above 9.2 KB.)
static int kq_split_waves(int units_a, double unit_bytes_a, int units_b, double unit_bytes_b, int slots) {
    const double share = 16.0 * units_b * unit_bytes_b / (units_a * unit_bytes_a + units_b * unit_bytes_b);
    int best = 1;
    double best_cost = 0.0;
    for (int nwb = 1; nwb <= 15; ++nwb) {
        const int nwa = 16 - nwb;
        const double ca = (double)((units_a + slots * nwa - 1) / (slots * nwa)) * unit_bytes_a, cb = (double)((units_b + slots * nwb - 1) / (slots * nwb)) * unit_bytes_b;
        const double cost = std::max(ca, cb);
        if (nwb == 1 || cost < best_cost || (cost == best_cost && fabs(nwb - share) < fabs(best - share))) { best = nwb; best_cost = cost; }
    }
    return best;
}

// unit bookkeeping of a generation-9 launch: the jobs' units concatenated (type group A first), the groups' arenas, the wave split
static bool kq_prepare(MatvecArgs& a, int& tb_out, std::string& err, int slots = 0) {   // slots: workgroups of the launch (0: the plain launch's grid)
    for (int j = 0; j < a.njobs; ++j) {
        if (!a.job[j].w.r9) { err = "mat-vec: K-quant matrix without a LAYOUT_L9 arena"; return false; }
    }
    const int ta = a.job[0].w.type;
    int tb = 0, item0 = 0, na = 0;
    const int spu = l9_spu(ta, a.K);
    for (int j = 0; j < a.njobs; ++j) {
        const int tj = a.job[j].w.type;
        const int units = a.gateup ? a.job[j].w.M : (a.job[j].w.M + 1) / 2;
        a.job[j].pair0 = item0;
        item0 += units;
        if (tj == ta && tb == 0) na += units;
        else if ((tb == 0 && tj == GT_Q6_K) || tj == tb) tb = tj;
        else { err = "unsupported weight-type mix in one launch"; return false; }
    }
    a.n_pairs = item0;
    a.n_groupA = na;
    // the units of a type group are one contiguous stream: job j + 1 of a group starts where job j ends (upload_r2c4 arenas)
    a.baseA = a.job[0].w.r9;
    a.baseB = nullptr;
    for (int j = 0; j < a.njobs; ++j) {
        const bool first_b = tb != 0 && a.job[j].pair0 == na;
        if (first_b) { a.baseB = a.job[j].w.r9; continue; }
        if (j == 0) continue;
        const int uj = a.job[j].pair0 - a.job[j - 1].pair0;
        if (a.job[j].w.r9 != a.job[j - 1].w.r9 + (size_t)uj * spu * l9_record_bytes(a.job[j - 1].w.type)) {
            err = "mat-vec jobs of one weight type are not contiguous in memory";
            return false;
        }
    }
    a.nwA = 16;
    if (tb != 0) a.nwA = 16 - kq_split_waves(na, (double)spu * l9_record_bytes(ta), item0 - na, (double)spu * l9_record_bytes(tb), slots > 0 ? slots : std::max(1, std::min(chip_cus(), item0)));
    tb_out = tb;
    return true;
}

static bool launch_matvec_kq(MatvecArgs& a, hipStream_t s, std::string& err) {
    ++g_kq_launches;
    int tb = 0;
    if (!kq_prepare(a, tb, err)) return false;
    const int ta = a.job[0].w.type, na = a.n_groupA;
    const bool ln = a.pro == PRO_LAYERNORM;
    if (ln && tb != 0) { err = "LayerNorm prologue with a mixed-type launch"; return false; }
    // one workgroup per CU; more only where a wave would otherwise own more than kV9MaxUnits units (its results wait in LDS
    // for its epilogue pass) — no real model shape gets there on 256 CUs, the 4-CU emulated chip of the tests does
    int gx = std::max(1, std::min(chip_cus(), a.n_pairs));
    {
        const int nb_units = a.n_pairs - na, nwb = 16 - a.nwA;
        if (a.nwA > 0) gx = std::max(gx, (na + a.nwA * kV9MaxUnits - 1) / (a.nwA * kV9MaxUnits));
        if (nb_units > 0) gx = std::max(gx, (nb_units + nwb * kV9MaxUnits - 1) / (nwb * kV9MaxUnits));
    }
    // (Measured and NOT taken, round 4: an 8-wave workgroup with a ring of 11 / 8 / 7 slots for launches in which a wave owns one unit of many
    // records — ffn_down, 2048 row pairs of eleven records — so that the whole unit is in flight: 10.8 us against 8.8 us per launch.  A CU
    // with eight streaming waves is served at about 8.6 B/cycle whatever their requests in flight, with sixteen at 10.7: DESIGN.md 5.)
    const dim3 grid((unsigned)gx), block(1024);
    a.grid_out = gx;   // (Engine::launch_pick: the head launch's workgroups = its key slots)
    if (a.pick_ws && gx > 2048) { err = "head launch with more than 2048 workgroups"; return false; }
    {
        if (a.emb_out && (tb != 0 || a.K > 16384)) { err = "emb_out on a mixed-type or wide launch"; return false; }
        if (tb != 0 && a.K > 16384) { err = "mixed-type launch with K > 16384"; return false; }
        // Ring slots per wave (records in flight): THREE for the single-type K-quant launches, four for the two-type (QKV) launch and the
        // 32-block types.  Measured late in round 4 (profiles/r04_decode_ablations.txt, alternating on one box, three boxes): 2 -> 736,
        // 3 -> 746-750, 4 -> 740-744 tok/s, 5 (125 registers, no spills) -> 728-734; the two-type launch with three: 742-745 against 747-750;
        // Q8_0 (config 3) with three: 535 against 539.
        // More requests in flight per wave lengthen a launch on this memory system; gate+up gains most (11.9 -> 11.2 us).
#define V9L(MK, TAV, TBV, LNV, EMBV) do { \
        auto kfn = matvec_v9_kernel<MK, TAV, TBV, LNV, EMBV, 16, ((TAV == GT_Q4_K || TAV == GT_Q5_K || TAV == GT_Q6_K) ? (TBV == 0 ? V9_NS_K : V9_NS_K2) : V9_NS_B)>; \
        constexpr size_t smem = sizeof(SmemV9<MK>); \
        CT_OPTIN_ONCE(kfn, smem); \
        CT_LAUNCH_DYN(kfn, grid, block, smem, s, a.x, a.norm_w, a.K, a.pro, a); } while (0)
        // (Round 5, measured and NOT taken: eight-wave workgroups for a launch whose units reach only waves 0..7 — the 7B's attention output projection —
        // 786.7 against 793.0 tok/s, alternating on one box: the idle waves' share of the prologue is worth more than their launch costs.)
#define V9T(MK, TAV) do { \
        if (a.emb_out) { if (ln) V9L(16384, TAV, 0, true, true); else V9L(16384, TAV, 0, false, true); } \
        else if (ln) V9L(MK, TAV, 0, true, false); \
        else if (tb != 0) V9L(16384, TAV, GT_Q6_K, false, false); \
        else V9L(MK, TAV, 0, false, false); } while (0)
#define V9(MK) do { \
        if (ta == GT_Q4_K) V9T(MK, GT_Q4_K); \
        else if (ta == GT_Q5_K) V9T(MK, GT_Q5_K); \
        else if (ta == GT_Q8_0) { if (a.emb_out) V9L(16384, GT_Q8_0, 0, false, true); else V9L(MK, GT_Q8_0, 0, false, false); } \
        else if (ta == GT_Q4_0) { if (a.emb_out) V9L(16384, GT_Q4_0, 0, false, true); else V9L(MK, GT_Q4_0, 0, false, false); } \
        else if (a.emb_out) { if (ln) V9L(16384, GT_Q6_K, 0, true, true); else V9L(16384, GT_Q6_K, 0, false, true); } \
        else if (ln) V9L(MK, GT_Q6_K, 0, true, false); \
        else V9L(MK, GT_Q6_K, 0, false, false); } while (0)
        if (a.K <= 16384) V9(16384); else V9(32768);
#undef V9
#undef V9T
#undef V9L
    }
    return true;
}

// One mat-vec launch.
//   K-quant jobs (LAYOUT_L9 arenas)    -> generation 9 (kernels_v9.h), work items are row pairs
//   LAYOUT_G4 (Q8_0 / Q4_0)            -> systolic 32-block kernel (kernels_q32.h), work items are 8-row tiles
static bool launch_matvec_one(MatvecArgs& a, hipStream_t s, std::string& err) {
    if (kq_can(a)) return launch_matvec_kq(a, s, err);
    err = "mat-vec: this launch shape has no kernel (a matrix without a LAYOUT_L9 arena, or weight types that cannot share a launch)";
    return false;
}

// A launch site with several matrices (QKV) whose weight types one kernel launch cannot take together — reference files mix freely
// (llama.cpp:4785-4850: Q4_K_S has attn_v in Q5_K beside Q4_K q/k; a Q8_0 tensor may sit beside K-quants) — is issued as one launch
// per group of jobs that CAN share a launch: K-quant jobs whose arenas are contiguous and whose types are all equal or "X.. then
// Q6_K.." (the two-type kernel), or Q8_0 / Q4_0 jobs of one type.  Every group recomputes the (cheap) prologue; the jobs' epilogues
// are independent, so the results are those of the single launch.
// A site whose matrices stay in file layout (F16 / F32: kernels_f16.h; Q4_1 / Q5_0 / Q5_1: kernels_raw32.h): one dot-product launch per
// matrix into the site's scratch rows, then the epilogue launch.
static bool launch_matvec_raw(MatvecArgs& a, hipStream_t s, std::string& err) {
    if (!a.f16_tmp) { err = "file-layout mat-vec without its scratch rows"; return false; }
    if (a.K % 32 || a.K > 32768) { err = "file-layout rows must be whole 32-element steps, at most 32768 elements"; return false; }
    int off = 0;
    for (int j = 0; j < a.njobs; ++j) {
        const DevMat& w = a.job[j].w;
        if (!(w.type == GT_F16 || w.type == GT_F32 || is_raw32(w.type)) || !w.raw) { err = "a launch site mixes file-layout matrices (F32, F16, Q4_1, Q5_0, Q5_1) with other weight types"; return false; }
        if (w.type == GT_F16 || w.type == GT_F32) {
            // 256-thread workgroups, 64 rows per pass: 1024-thread ones measured 75 against 125 tok/s on the 7B F16 file
            const int gx = std::max(1, std::min((w.M + 63) / 64, 8 * chip_cus()));
            if (w.type == GT_F16) {
                auto kfn = matvec_f16_kernel<256, false>;
                CT_OPTIN_ONCE(kfn, (size_t)64 * 1024);
                CT_LAUNCH_DYN(kfn, dim3((unsigned)gx), dim3(256), (size_t)a.K * 2, s, a.x, a.norm_w, a.norm_b, a.K, a.pro, a.eps, (const void*)w.raw, w.M, a.f16_tmp + off);
            } else {
                if (a.K > 16384) { err = "F32 rows of more than 16384 elements are not supported"; return false; }
                auto kfn = matvec_f16_kernel<256, true>;
                CT_OPTIN_ONCE(kfn, (size_t)64 * 1024);
                CT_LAUNCH_DYN(kfn, dim3((unsigned)gx), dim3(256), (size_t)a.K * 4, s, a.x, a.norm_w, a.norm_b, a.K, a.pro, a.eps, (const void*)w.raw, w.M, a.f16_tmp + off);
            }
        } else {   // 16 rows per pass of a 256-thread workgroup (128 / 512 threads measured 170 / 218 against 225 tok/s on the 7B Q4_1 file:
                   // profiles/r03_raw32_q41_q50_q51.txt); LDS: K quant bytes + 8 bytes per block
            const int gx = std::max(1, std::min((w.M + 15) / 16, 8 * chip_cus()));
            const size_t lds = (size_t)a.K + (size_t)(a.K / 32) * 8;
#define CT_RAW32(T)                                                                                                                        \
    {                                                                                                                                      \
        auto kfn = matvec_raw32_kernel<T, 256>;                                                                                            \
        CT_OPTIN_ONCE(kfn, (size_t)64 * 1024);                                                                                             \
        CT_LAUNCH_DYN(kfn, dim3((unsigned)gx), dim3(256), lds, s, a.x, a.norm_w, a.norm_b, a.K, a.pro, a.eps, (const uint8_t*)w.raw, w.M, a.f16_tmp + off); \
    }
            if (w.type == GT_Q4_1) CT_RAW32(GT_Q4_1)
            else if (w.type == GT_Q5_0) CT_RAW32(GT_Q5_0)
            else CT_RAW32(GT_Q5_1)
#undef CT_RAW32
        }
        off += w.M;
    }
    const int n_rows = a.gateup ? a.job[0].w.M : off;
    CT_LAUNCH(f16_epilogue_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), s, a, (const float*)a.f16_tmp, n_rows);
    return true;
}

static bool launch_matvec(MatvecArgs& a, hipStream_t s, std::string& err) {
    for (int j = 0; j < a.njobs; ++j)
        if (a.job[j].w.type == GT_F16 || a.job[j].w.type == GT_F32 || is_raw32(a.job[j].w.type)) return launch_matvec_raw(a, s, err);
    if (a.gateup || a.njobs <= 1) return launch_matvec_one(a, s, err);
    int i = 0;
    while (i < a.njobs) {
        int j = i + 1;
        const DevMat& w0 = a.job[i].w;
        const bool kq0 = is_kquant(w0.type) && w0.r9;
        const bool b0 = is_block32(w0.type) && w0.r9;
        int tb = 0;
        while (j < a.njobs) {
            const DevMat& wp = a.job[j - 1].w;
            const DevMat& wj = a.job[j].w;
            if (kq0) {
                if (!(is_kquant(wj.type) && wj.r9)) break;
                if (wj.r9 != wp.r9 + (size_t)((wp.M + 1) / 2) * l9_spu(wp.type, a.K) * l9_record_bytes(wp.type)) break;   // not the same arena run
                if (wj.type != wp.type) {
                    if (tb != 0 || wj.type != GT_Q6_K || wp.type != w0.type || a.K > 16384) break;   // only "X.. then Q6_K.." shares a launch
                    tb = wj.type;
                }
            } else if (b0) {
                if (wj.type != w0.type || !wj.r9 || wj.r9 != wp.r9 + (size_t)((wp.M + 1) / 2) * l9_spu(wp.type, a.K) * l9_record_bytes(wp.type)) break;
            } else {
                break;
            }
            ++j;
        }
        if (i == 0 && j == a.njobs) return launch_matvec_one(a, s, err);
        MatvecArgs g = a;
        g.njobs = j - i;
        for (int k = 0; k < g.njobs; ++k) g.job[k] = a.job[i + k];
        for (int k = g.njobs; k < 3; ++k) g.job[k] = MatJob();
        if (!launch_matvec_one(g, s, err)) return false;
        i = j;
    }
    return true;
}

static void set_jobs(MatvecArgs& a, std::initializer_list<std::pair<const DevMat*, int>> jobs) {
    int j = 0, pair0 = 0;
    for (auto& it : jobs) {
        a.job[j].w = *it.first;
        a.job[j].epi = it.second;
        a.job[j].pair0 = pair0;
        pair0 += (it.first->M + 1) / 2;
        ++j;
    }
    a.njobs = j;
    a.n_pairs = pair0;
}

// Debug aid (CT_AMD_DUMP=<dir>): after every launch of a token step, synchronise and write the scratch buffers to
// <dir>/t<eval#>_l<layer>_<site>.bin so two builds (HIP vs the CPU emulation of the same sources) can be diffed.
void Engine::debug_dump(const char* site, int layer) {
    if (!dump_dir_) return;
    (void)hipStreamSynchronize(stream_);
    const int E = hp_.n_embd, F = hp_.n_ff;
    std::vector<float> buf((size_t)E * 2 + F + (size_t)hp_.n_head * n_ctx_ + E);
    size_t o = 0;
    (void)hipMemcpy(&buf[o], x_, (size_t)E * 4, hipMemcpyDeviceToHost); o += E;
    (void)hipMemcpy(&buf[o], attn_out_, (size_t)E * 4, hipMemcpyDeviceToHost); o += E;
    (void)hipMemcpy(&buf[o], h_, (size_t)F * 4, hipMemcpyDeviceToHost); o += F;
    (void)hipMemcpy(&buf[o], scores_, (size_t)hp_.n_head * n_ctx_ * 4, hipMemcpyDeviceToHost); o += (size_t)hp_.n_head * n_ctx_;
    (void)hipMemcpy(&buf[o], q_f16_, (size_t)E * 2, hipMemcpyDeviceToHost);
    char name[512];
    snprintf(name, sizeof(name), "%s/t%04d_l%02d_%s.bin", dump_dir_, dump_seq_, layer, site);
    FILE* f = fopen(name, "wb");
    if (f) { fwrite(buf.data(), 4, buf.size(), f); fclose(f); }
    if (!strcmp(site, "2attn")) {  // also the layer's K/V cache
        const int G = hp_.n_embd_gqa();
        std::vector<uint16_t> kv((size_t)n_ctx_ * G + (size_t)v_stride_ * G);
        (void)hipMemcpy(kv.data(), kcache_ + (size_t)layer * n_ctx_ * G, (size_t)n_ctx_ * G * 2, hipMemcpyDeviceToHost);
        (void)hipMemcpy(kv.data() + (size_t)n_ctx_ * G, vcache_ + (size_t)layer * v_stride_ * G, (size_t)v_stride_ * G * 2, hipMemcpyDeviceToHost);
        snprintf(name, sizeof(name), "%s/t%04d_l%02d_kv.bin", dump_dir_, dump_seq_, layer);
        f = fopen(name, "wb");
        if (f) { fwrite(kv.data(), 2, kv.size(), f); fclose(f); }
    }
}

void Engine::apply_trace(MatvecArgs& a, const char* site) {
    if (trace_site_ && !strcmp(site, trace_site_)) {
        a.dbg |= 32;
        a.dbg_sink = (float*)trace_buf_;
    }
}

bool Engine::run_matvec(MatvecArgs& a, std::string& err) {
    if (!a.f16_tmp) a.f16_tmp = f16_tmp_;   // file-layout matrices (F16 / Q4_1 / Q5_x head behind chunked K-quant layers): every step function gets them
    return launch_matvec(a, stream_, err);
}

// The head launch of the generation-9 mat-vec picks the greedy token itself and, where the token step advanced the cursor in its last
// ffn_down launch (`cont`), prepares the next step of a greedy chain (kernels_v9.h:v9_pick_store + kernels.h:pick_cont_kernel).  Legacy graphs sample on the host
// (models/llm.h), file-layout heads (F16, Q4_1, ...) keep the argmax launch.
bool Engine::head_folds() const {
    return fold_on_ && l1_ == hp_.n_layer && !hp_.legacy() && output_.r9 && (is_kquant(output_.type) || is_block32(output_.type)) && hp_.n_embd <= 16384;
}
void Engine::set_head_fold(MatvecArgs& a, bool cont) {
    head_cont_ = false;
    if (!head_folds() || !a.emb_out) return;
    a.pick_ws = pick_ws_;
    head_cont_ = cont && l0_ == 0 && tok_embd_.raw && fold_on_ == 1 && ggml_row_bytes(tok_embd_.type, hp_.n_embd) <= (size_t)64 * 1024;
}
// behind a folding head launch: the pick's second half (+ the continuation of a greedy chain where the head said so)
void Engine::launch_pick(const MatvecArgs& head) {
    if (!head_folds() || !head.pick_ws || head.grid_out <= 0) return;
    head_folded_ = true;   // this handle's head launches leave keys: req_logits launches no argmax
    const bool c = head_cont_;
    const size_t rb = c ? (size_t)ggml_row_bytes(tok_embd_.type, hp_.n_embd) + 16 : 16;
    CT_OPTIN_ONCE(pick_cont_kernel, (size_t)80 * 1024);
    CT_LAUNCH_DYN(pick_cont_kernel, dim3(1), dim3(1024), rb, stream_, (const unsigned long long*)pick_ws_, 16 * head.grid_out, d_argmax_, pick_host_, c ? d_state_ : (int*)nullptr,
              c ? x_ : (float*)nullptr, (const uint8_t*)tok_embd_.raw, tok_embd_.type, hp_.n_embd);
}

// One fused attention launch for the current token over this layer's fp16 KV cache (kernels_exact.h).
// The fused QKV + attention launch of a token step (kernels_qa9.h) takes a layer whose q / k / v matrices can share ONE generation-9
// launch (K-quants in one arena run: q and k of one type — Q4_K or Q5_K —, v the same or Q6_K), heads of 64 or 128, the ring form of
// contexts up to 1024, and a device on which the attention grid (n_head x channel groups <= CUs, one 1024-thread workgroup per CU) is
// resident as a whole: the workgroups of a KV head wait for each other's rows (kernels_qa9.h: co-residency).
bool Engine::qa_can(const Layer& L) const {
    if (!fuse_qa_ || hp_.legacy() || !xq_) return false;
    if (hp_.falcon() && !falcon_fold_) return false;   // (falcon: the three row ranges of attn_qkv as jobs, NEOX pairs as row pairs: engine_load.h)
    const int hd = hp_.head_dim(), E = hp_.n_embd;
    if (!(hd == 128 || hd == 64) || E > 16384 || n_ctx_ > 1024 || hp_.n_head % hp_.n_head_kv) return false;
    if (hp_.falcon() && hd != 64) return false;        // (the LayerNorm instantiations: head size 64, every falcon model's)
    const DevMat *q = hp_.falcon() ? &L.wq_v : &L.wq, *k = hp_.falcon() ? &L.wk_v : &L.wk, *v = hp_.falcon() ? &L.wv_v : &L.wv;
    if (!q->r9 || !k->r9 || !v->r9 || q->type != k->type || !(q->type == GT_Q4_K || q->type == GT_Q5_K)) return false;
    if (!(v->type == q->type || v->type == GT_Q6_K)) return false;
    if (k->r9 != q->r9 + (size_t)((q->M + 1) / 2) * l9_spu(q->type, E) * l9_record_bytes(q->type)) return false;
    if (v->type == q->type && v->r9 != k->r9 + (size_t)((k->M + 1) / 2) * l9_spu(k->type, E) * l9_record_bytes(k->type)) return false;
#ifdef CT_EMU
    const int cus = 256;   // the test build runs the two phases as two passes: no residency to check, the product's grid shapes
#else
    const int cus = chip_cus();
#endif
    int ng = hd / 16;
    while (ng > hd / 64 && hp_.n_head * ng > cus) ng >>= 1;
    const int pvw = hd / ng / 16;
    if (hp_.n_head * ng > cus || !(pvw == 1 || (pvw == 2 && (hd == 128 || hp_.falcon())))) return false;
    // at most two units per wave (kernels_qa9.h:QaItems): the group's (rep + 2) * hd / 2 units over its rep * ng workgroups of sixteen waves —
    // with a second weight type for v, the waves are split by bytes and either part may be as small as one wave per workgroup
    const int rep = hp_.n_head / hp_.n_head_kv, NG = rep * ng;
    if ((rep & (rep - 1)) || (ng & (ng - 1))) return false;   // the workgroup map is shifts and masks
    if (v->type == q->type) { if ((rep + 2) * (hd / 2) > 2 * 16 * NG) return false; }
    else {   // kq_prepare's split of the sixteen waves by bytes
        const int ua = (q->M + 1) / 2 + (k->M + 1) / 2, ub = (v->M + 1) / 2, spu = l9_spu(q->type, E);
        const int nwb = kq_split_waves(ua, (double)spu * l9_record_bytes(q->type), ub, (double)spu * l9_record_bytes(v->type), hp_.n_head * ng);
        if ((rep + 1) * (hd / 2) > 2 * (16 - nwb) * NG || hd / 2 > 2 * nwb * NG) return false;
    }
    return true;
}

bool Engine::launch_qkv_attn(MatvecArgs& a, uint16_t* kc, uint16_t* vc, int il, std::string& err) {
    const int hd = hp_.head_dim();
#ifdef CT_EMU
    const int cus = 256;
#else
    const int cus = chip_cus();
#endif
    int tb = 0;
    {
        int ng0 = hd / 16;
        while (ng0 > hd / 64 && hp_.n_head * ng0 > cus) ng0 >>= 1;
        if (!kq_prepare(a, tb, err, hp_.n_head * ng0)) return false;
        const int rep = hp_.n_head / hp_.n_head_kv, NG = rep * ng0;
        const bool ok = tb == 0 ? (rep + 2) * (hd / 2) <= 2 * 16 * NG : ((rep + 1) * (hd / 2) <= 2 * a.nwA * NG && hd / 2 <= 2 * (16 - a.nwA) * NG);
        if (!ok) { err = "fused QKV + attention launch: more than two units per wave"; return false; }
    }
    const int ta = a.job[0].w.type;
    AttnArgsX ax = AttnArgsX();
    fill_attn_args(ax, kc, vc, 0);
    int ng = hd / 16;
    while (ng > hd / 64 && hp_.n_head * ng > cus) ng >>= 1;
    const int pvw = hd / ng / 16;
    QaArgs qa;
    qa.xq = xq_; qa.epoch = (const unsigned*)(d_state_ + 4 + n_ctx_); qa.err = qa_err_; qa.layer = il & 255; qa.ng = ng; qa.phase = 0;
    qa.ng_sh = 0; while ((1 << qa.ng_sh) < ng) ++qa.ng_sh;
    qa.rep_sh = 0; while ((1 << qa.rep_sh) < hp_.n_head / hp_.n_head_kv) ++qa.rep_sh;
    qa.out = ax.out; qa.exp_tab = ax.exp_tab; qa.trace = ax.trace; qa.kq_scale = ax.kq_scale; qa.n_head = ax.n_head; qa.n_head_kv = ax.n_head_kv;
    static const int exp_phase1 = env_int("CT_AMD_QA_PHASE1", 0);   // experiment: the fused launch's mat-vec phase only, the attention as its own launch
    if (exp_phase1) qa.phase = 1;
    const size_t al = 15;
    const size_t smem = ((sizeof(SmemV9<16384>) + al) & ~al) + ((sizeof(QaSmem) + al) & ~al) + (size_t)((n_ctx_ + 63) & ~63) * 4;
    const dim3 grid((unsigned)(hp_.n_head * ng)), block(1024);
    ++g_kq_launches;
    ++qa_launches_;
#ifdef CT_EMU
#define QAL(TAV, TBV, HDV, NWVV, ...) do { \
        auto kfn = qkv_attn9_kernel<TAV, TBV, HDV, NWVV, (TBV == 0 ? V9_NS_K : V9_NS_K2), ##__VA_ARGS__>; \
        qa.phase = 1; CT_LAUNCH_DYN(kfn, grid, block, smem, stream_, a.x, a.norm_w, a.K, a.pro, a, qa); \
        qa.phase = 2; CT_LAUNCH_DYN(kfn, grid, block, smem, stream_, a.x, a.norm_w, a.K, a.pro, a, qa); } while (0)
#else
#define QAL(TAV, TBV, HDV, NWVV, ...) do { \
        auto kfn = qkv_attn9_kernel<TAV, TBV, HDV, NWVV, (TBV == 0 ? V9_NS_K : V9_NS_K2), ##__VA_ARGS__>; \
        CT_OPTIN_ONCE(kfn, (size_t)150 * 1024); \
        CT_LAUNCH_DYN(kfn, grid, block, smem, stream_, a.x, a.norm_w, a.K, a.pro, a, qa); } while (0)
#endif
#define QAT(TAV, TBV) do { \
        if (hd == 64) QAL(TAV, TBV, 64, 14); \
        else if (pvw == 1) QAL(TAV, TBV, 128, 14); \
        else QAL(TAV, TBV, 128, 12); } while (0)   /* 16 - (channels per workgroup) / 8 score waves (kernels_qa9.h) */
    if (a.pro == PRO_LAYERNORM) {   // falcon: one weight type, head size 64, 16 or 32 channels per workgroup
        if (tb != 0 || hd != 64) { err = "fused QKV + attention launch: LayerNorm form with a second weight type or a head size other than 64"; return false; }
        if (ta == GT_Q4_K) { if (pvw == 1) QAL(GT_Q4_K, 0, 64, 14, true); else QAL(GT_Q4_K, 0, 64, 12, true); }
        else { if (pvw == 1) QAL(GT_Q5_K, 0, 64, 14, true); else QAL(GT_Q5_K, 0, 64, 12, true); }
    } else
    if (ta == GT_Q4_K) { if (tb) QAT(GT_Q4_K, GT_Q6_K); else QAT(GT_Q4_K, 0); }
    else { if (tb) QAT(GT_Q5_K, GT_Q6_K); else QAT(GT_Q5_K, 0); }
#undef QAT
#undef QAL
    if (exp_phase1) launch_attention(kc, vc);
    return true;
}

void Engine::fill_attn_args(AttnArgsX& ax, uint16_t* kc, uint16_t* vc, int nt) {
    const int hd = hp_.head_dim();
    ax.q_f16 = nt ? q_f16_b_ : q_f16_; ax.kcache = kc; ax.vcache = vc; ax.out = nt ? attn_out_b_ : attn_out_; ax.pos = d_state_ + 1;
    ax.q_stride = hp_.n_embd; ax.out_stride = hp_.n_embd;
    ax.exp_tab = exp_tab_; ax.n_total = d_state_ + 2; ax.n_head = hp_.n_head; ax.n_head_kv = hp_.n_head_kv; ax.head_dim = hd;
    ax.n_embd_gqa = hp_.n_embd_gqa(); ax.n_ctx = n_ctx_; ax.v_stride = v_stride_;
    // llama.cpp:2263 / :2596 write 1.0f / sqrtf(float(n_embd) / n_head); the legacy loaders (mpt.cc:460-462, gpt2.cc:540-543) write
    // 1.0f / sqrt(float(n_embd) / n_head): ::sqrt is the double function there, the double quotient is rounded to float once by
    // ggml_new_f32 — an ulp apart from the sqrtf form at head sizes such as 96 or 112 (equal at 64 and 128)
    ax.kq_scale = hp_.legacy() ? (float)(1.0 / sqrt((double)((float)hp_.n_embd / (float)hp_.n_head)))
                               : 1.0f / sqrtf((float)hp_.n_embd / (float)hp_.n_head);
    ax.alibi = alibi_;
    if (trace_site_ && (!strcmp(trace_site_, "attn") || !strcmp(trace_site_, "qa"))) ax.trace = trace_buf_;
}

void Engine::launch_attention(uint16_t* kc, uint16_t* vc, int nt) {
    const int hd = hp_.head_dim();
    AttnArgsX ax = AttnArgsX();
    fill_attn_args(ax, kc, vc, nt);
    if (nt > 0 && fast_pf_ && !alibi_ && (hd == 128 || hd == 64) && env_int("CT_AMD_ATTN_MM", 1) != 0) {
        // order-free prompt chunks: K.Q and V.P on the f16 matrix cores (kernels_mm8.h:attn_mm_kernel), any context length
        const dim3 gm((unsigned)hp_.n_head, (unsigned)((nt + 31) / 32));
        const size_t sm = (size_t)8 * 32 * 12 + (size_t)8 * 16 * 64 * 4;
        if (hd == 128) { auto kfn = attn_mm_kernel<128>; CT_LAUNCH_DYN(kfn, gm, dim3(512), sm, stream_, ax, nt); }
        else { auto kfn = attn_mm_kernel<64>; CT_LAUNCH_DYN(kfn, gm, dim3(512), sm, stream_, ax, nt); }
        return;
    }
    const dim3 ag((unsigned)hp_.n_head, (unsigned)(hd / 64), (unsigned)std::max(1, nt));   // nt > 0: the tokens of a prompt chunk
    const size_t smem = (size_t)((n_ctx_ + 63) & ~63) * 4;   // the probability row
#define ATTN(NTV, HDV, ALLV, GRID) do { \
        auto kfn = attn_fused_exact_kernel<NTV, HDV, ALLV>; \
        CT_OPTIN_ONCE(kfn, (size_t)kMaxCtxFused * 4); \
        CT_LAUNCH_DYN(kfn, GRID, dim3(NTV), smem, stream_, ax); } while (0)
    if (n_ctx_ > kMaxCtxFused) {   // llama / falcon contexts whose probability row exceeds LDS: the row in global memory, one workgroup per head
        ax.scores = scores_;
        const dim3 agp((unsigned)hp_.n_head, 1u, 1u);
#define ATTN_GP(HDV) do { \
        auto kfn = attn_fused_exact_kernel<512, HDV, true, false, true>; \
        CT_LAUNCH_DYN(kfn, agp, dim3(512), (size_t)64, stream_, ax); } while (0)
        if (hd == 128) ATTN_GP(128); else if (hd == 64) ATTN_GP(64); else if (hd == 192) ATTN_GP(192); else ATTN_GP(256);
#undef ATTN_GP
        return;
    }
    if (alibi_) {   // MPT: the fused kernel with the ALiBi term (one workgroup per (head, token) for chunks); head sizes checked at load
#define ATTN_ALIBI(NTV, HDV, ALLV, GRID) do { \
        auto kfn = attn_fused_exact_kernel<NTV, HDV, ALLV, true>; \
        CT_OPTIN_ONCE(kfn, (size_t)kMaxCtxFused * 4); \
        CT_LAUNCH_DYN(kfn, GRID, dim3(NTV), smem, stream_, ax); } while (0)
        const dim3 ag1((unsigned)hp_.n_head, 1u, (unsigned)std::max(1, nt));
        if (hd == 112) {   // MPT-30B heads: three 32-element steps + the scalar tail of 16; all channels of a head in one workgroup
            if (nt > 0) ATTN_ALIBI(256, 112, true, ag1); else ATTN_ALIBI(512, 112, true, ag1);
        } else if (nt > 0) { if (hd == 128) ATTN_ALIBI(256, 128, true, ag1); else ATTN_ALIBI(256, 64, true, ag1); }
        else { if (hd == 128) ATTN_ALIBI(512, 128, false, ag); else ATTN_ALIBI(512, 64, false, ag); }
#undef ATTN_ALIBI
        return;
    }
    if (nt > 0 && (hd == 128 || hd == 64) && chunk_below_128_ && n_ctx_ >= 128 && env_int("CT_AMD_ATTN_TILE", 1) != 0) {
        // every position of this chunk is below 128: K / V of a head go through LDS once per 16 tokens
        const dim3 gt((unsigned)hp_.n_head, (unsigned)((nt + 15) / 16));
        const size_t sm = (size_t)128 * (hd * 2 + 64) + (size_t)hd * 320 + (size_t)16 * 160 * 4;
        static const char* pgt = getenv("CT_AMD_PG_TRACE");   // measurement only: "attn" = in-kernel stamps of this launch
        const bool tr = pgt && !strcmp(pgt, "attn") && nt > 64;
        if (tr) ax.trace = trace_buf_;
        if (hd == 128) { auto kfn = attn_chunk_tile_kernel<128>; CT_OPTIN_ONCE(kfn, (size_t)96 * 1024); CT_LAUNCH_DYN(kfn, gt, dim3(1024), sm, stream_, ax, nt); }
        else { auto kfn = attn_chunk_tile_kernel<64>; CT_OPTIN_ONCE(kfn, (size_t)96 * 1024); CT_LAUNCH_DYN(kfn, gt, dim3(1024), sm, stream_, ax, nt); }
        if (tr) {
            (void)hipStreamSynchronize(stream_);
            unsigned long long hbuf[256];
            (void)hipMemcpy(hbuf, trace_buf_, sizeof hbuf, hipMemcpyDeviceToHost);
            fprintf(stderr, "attn_trace:");
            for (int w : {0, 5, 15}) fprintf(stderr, " [wave %d n_kv %llu: load %llu, scores %llu, softmax %llu, pv %llu]", w, hbuf[16 * w + 5], hbuf[16 * w + 1] - hbuf[16 * w],
                                              hbuf[16 * w + 2] - hbuf[16 * w + 1], hbuf[16 * w + 3] - hbuf[16 * w + 2], hbuf[16 * w + 4] - hbuf[16 * w + 3]);
            fprintf(stderr, "\n");
        }
        return;
    }
    if (nt > 0 && (hd == 128 || hd == 64) && env_int("CT_AMD_ATTN_LONG", 1) != 0) {
        // prompt chunk beyond position 128: K / V tiles of 64 positions through LDS, fetched once per 16 (8) tokens of a head; the
        // tokens' probability rows live in LDS, which bounds the context this kernel takes (2048 with 16 tokens, 4096 with 8)
        const int row = (n_ctx_ + 63) & ~63;
        const size_t tile_bytes = hd == 128 ? (size_t)128 * 192 : (size_t)64 * 192;
        const size_t cap = (size_t)160 * 1024;
        const int ntok = tile_bytes + (size_t)16 * row * 4 <= cap ? 16 : (tile_bytes + (size_t)8 * row * 4 <= cap ? 8 : 0);
        if (ntok) {
            const dim3 gl((unsigned)hp_.n_head, (unsigned)((nt + ntok - 1) / ntok));
            const size_t sm = tile_bytes + (size_t)ntok * row * 4;
#define ATTNL(KERN, HDV, NTV) do { \
                auto kfn = KERN<HDV, NTV>; \
                CT_OPTIN_ONCE(kfn, cap); \
                CT_LAUNCH_DYN(kfn, gl, dim3(NTV * 64), sm, stream_, ax, nt, row); } while (0)
            // (eight rows: the form with three tiles in flight, kernels_exact.h:attn_chunk_long8_kernel)
            if (hd == 128) { if (ntok == 16) ATTNL(attn_chunk_long_kernel, 128, 16); else ATTNL(attn_chunk_long8_kernel, 128, 8); }
            else { if (ntok == 16) ATTNL(attn_chunk_long_kernel, 64, 16); else ATTNL(attn_chunk_long8_kernel, 64, 8); }
#undef ATTNL
            return;
        }
    }
    if (nt > 0 && (hd == 128 || hd == 64) && n_ctx_ <= 4096 && env_int("CT_AMD_ATTN_WAVE", 1) != 0) {
        // prompt chunk, contexts whose probability rows fit LDS eight (four) at a time: one WAVE per (head, token)
        const int row = (n_ctx_ + 63) & ~63;
        const bool w8 = (size_t)8 * row * 4 <= (size_t)64 * 1024;
        const size_t sm = (size_t)(w8 ? 8 : 4) * row * 4;
        const dim3 gw((unsigned)hp_.n_head, (unsigned)((nt + (w8 ? 8 : 4) - 1) / (w8 ? 8 : 4)));
#define ATTNW(HDV, WPBV) do { \
            auto kfn = attn_chunk_wave_kernel<HDV, WPBV>; \
            CT_OPTIN_ONCE(kfn, (size_t)64 * 1024); \
            CT_LAUNCH_DYN(kfn, gw, dim3(WPBV * 64), sm, stream_, ax, nt, row); } while (0)
        if (hd == 128) { if (w8) ATTNW(128, 8); else ATTNW(128, 4); }
        else { if (w8) ATTNW(64, 8); else ATTNW(64, 4); }
#undef ATTNW
        return;
    }
    if (nt > 0 && (hd == 128 || hd == 64)) {
        // prompt chunk: n_head x nt workgroups, each latency-bound — 256-thread workgroups let three of them share a CU
        // (the arithmetic does not depend on the workgroup size: scores, softmax and V*P are per position / per channel),
        // all channels of a head in one workgroup (ALLCH)
        const dim3 ag1((unsigned)hp_.n_head, 1u, (unsigned)nt);
        if (hd == 128) ATTN(256, 128, true, ag1); else ATTN(256, 64, true, ag1);
        return;
    }
    static const int attn_gen = env_int("CT_AMD_ATTN_GEN", 9);   // 7: attn_fused_exact_kernel (A/B partner)
    if (nt == 0 && attn_gen != 7 && (hd == 128 || hd == 64)) {
        // decode, generation 9 (kernels_attn9.h): a workgroup per (head, group of output channels); the groups per head are what
        // makes the grid about one workgroup per CU — every group recomputes the head's score row, so no more of them than that
        int ng = hd / 16;
        while (ng > hd / 64 && hp_.n_head * ng > chip_cus()) ng >>= 1;   // at most four V*P waves beside the seven score waves: the kernel is built for <= 768 threads
        const int pv_waves = hd / ng / 16;
        const dim3 g9((unsigned)(hp_.n_head * ng)), b9((unsigned)(448 + 64 * pv_waves));
#define ATTN9(HDV, PBV, VBV) do { \
        auto kfn = attn_decode9_kernel<HDV, PBV, VBV>; \
        CT_OPTIN_ONCE(kfn, (size_t)kMaxCtxFused * 4); \
        CT_LAUNCH_DYN(kfn, g9, b9, smem, stream_, ax.pos - 1, ng, ax.n_head, ax.n_head_kv, ax); } while (0)
        // ring depth of the K / V requests; seven score waves with two K-row slots each measured best at contexts <= 1024 (3 / 4 / 5 score
        // waves, four slots: 0-4 % slower per token on the 7B, profiles/r03_attn9_score_waves_ab.txt)
        const bool deep = n_ctx_ > 1024;
        if (deep) {
            // Long-context form (kernels_attn9.h): four K-row slots, sixteen V chunks, FOUR score waves — one per SIMD, beside the V*P
            // waves: with seven, six of them share three SIMDs and finish late (the score phase is VALU work); measured at 2001 positions
            // 16.2 against 16.7 us (7B) and 16.4 against 17.4 us (70B widths), profiles/r03_attn9_ring_fix.txt.  At contexts <= 1024
            // the forms measure the same per token (717-727 tok/s); the shallow one stays there.
            const dim3 b9d((unsigned)(64 * (4 + pv_waves)));
#define ATTN9D(HDV, SHV) do { \
            auto kfn = attn_decode9_kernel<HDV, 4, 16, 4, 512, SHV>; \
            CT_OPTIN_ONCE(kfn, (size_t)kMaxCtxFused * 4); \
            CT_LAUNCH_DYN(kfn, g9, b9d, smem, stream_, ax.pos - 1, ng, ax.n_head, ax.n_head_kv, ax); } while (0)
            // one score row per head, shared by its channel groups (kernels_attn9.h: SHARE) where the whole grid is resident at once
            bool share = attn_share_ && xs_ && ng >= 2 && hp_.n_head * ng <= chip_cus();
#ifdef CT_EMU
            share = false;   // (the test build runs workgroups one after the other: a gather would wait for ever)
#endif
            if (share) { ax.xs = xs_; ax.epoch = (const unsigned*)(d_state_ + 4 + n_ctx_); ax.err = qa_err_; ax.layer = cur_layer_ & 255; }
            if (attn_free_) {   // CT_AMD_DECODE_ATTN=fast: every wave a score wave and a V*P wave, the V*P steps of a channel split over the waves
                ++g_attn_free_launches;
#define ATTN9F(HDV, SHV) do { \
                auto kfn = attn_decode9_free_kernel<HDV, SHV>; \
                CT_OPTIN_ONCE(kfn, (size_t)kMaxCtxFused * 4); \
                CT_LAUNCH_DYN(kfn, g9, dim3(512), smem, stream_, ax.pos - 1, ng, ax.n_head, ax.n_head_kv, ax); } while (0)
                if (hd == 128) { if (share) ATTN9F(128, true); else ATTN9F(128, false); }
                else { if (share) ATTN9F(64, true); else ATTN9F(64, false); }
#undef ATTN9FV
#undef ATTN9FV
#undef ATTN9F
                return;
            }
            if (share) {
                if (hd == 128) ATTN9D(128, true); else ATTN9D(64, true);
            } else if (hd == 128) ATTN9D(128, false); else ATTN9D(64, false);
#undef ATTN9D
            return;
        }
        if (hd == 128) ATTN9(128, 2, 4); else ATTN9(64, 2, 4);
#undef ATTN9
        return;
    }
    if (hd == 128) ATTN(512, 128, false, ag);
    else if (hd == 64) ATTN(512, 64, false, ag);
    else if (hd == 192) ATTN(512, 192, false, ag);
    else ATTN(512, 256, false, ag);
#undef ATTN
}

// One mat-vec site of a prompt chunk on the f16 matrix cores (kernels_pg.h): stage images of the nt activation rows in the
// layout(s) the site's weight types read, then one launch per weight type over the LAYOUT_R2C4 records of the decode path.
bool Engine::pg_matvec(MatvecArgs& m, const float* x, int ldx, int nt, int ld_out, int ld_res, std::string& err) {
    // Tokens per workgroup: 32 amortise the weight unpack over two matrix products — unless the launch then has fewer workgroups than
    // the chip has CUs (Wo, ffn_down of a 7B: 32 x 4; the Q6_K attn_v beside Q4_K q / k): 16-token groups double the workgroups and every
    // CU gets one.  The two image layouts (Q4_K / Q5_K launches, Q6_K launch) choose separately: their launches are separate.
    auto tg_of = [&](bool q6) {
        int tg = nt > 16 ? 32 : 16;
        if (tg == 32) {
            int items = 0;
            for (int j = 0; j < m.njobs; ++j)
                if ((m.job[j].w.type == GT_Q6_K) == q6) items += m.gateup ? (m.job[j].w.M + 7) / 8 : (m.job[j].w.M + 15) / 16;
            if (((items + kPgWaves - 1) / kPgWaves) * ((nt + 31) / 32) < chip_cus() * (8 / kPgWaves)) tg = 16;
        }
        if (pg_force_tg_ == 16 || pg_force_tg_ == 32) tg = pg_force_tg_;
        return tg;
    };
    const int tg45 = tg_of(false), tg6 = tg_of(true), nb = m.K / 256;
    if ((size_t)((nt + tg45 - 1) / tg45) * nb * pg_stage_bytes(tg45) > acts_h_half_ || (size_t)((nt + tg6 - 1) / tg6) * nb * pg_stage_bytes(tg6) > acts_h_half_) {
        err = "stage images exceed their buffer"; return false;
    }
    bool has45 = false, has6 = false;
    for (int j = 0; j < m.njobs; ++j) (m.job[j].w.type == GT_Q6_K ? has6 : has45) = true;
    uint8_t* img45 = has45 ? acts_h_ : nullptr;
    uint8_t* img6 = has6 ? acts_h_ + acts_h_half_ : nullptr;
    static const int pgq_remap = env_int("CT_AMD_PGQ_REMAP", 1);
    const int qn = pgq_remap ? nt : -1;
    const dim3 qg((unsigned)(pgq_remap ? 8 * ((nt + 7) / 8) : nt)), qb(1024);
    if (m.pro == PRO_LAYERNORM) {   // falcon: n_embd-long inputs only
        if (m.K <= 4096) CT_LAUNCH((pg_quantize_kernel<4096, true>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, img45, img6, tg45, tg6, m.norm_b, qn);
        else CT_LAUNCH((pg_quantize_kernel<12288, true>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, img45, img6, tg45, tg6, m.norm_b, qn);
    } else if (m.K <= 4096) CT_LAUNCH((pg_quantize_kernel<4096, false>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, img45, img6, tg45, tg6, (const float*)nullptr, qn);
    else if (m.K <= 12288) CT_LAUNCH((pg_quantize_kernel<12288, false>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, img45, img6, tg45, tg6, (const float*)nullptr, qn);
    else CT_LAUNCH((pg_quantize_kernel<32768, false>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, img45, img6, tg45, tg6, (const float*)nullptr, qn);
    constexpr int NW = kPgWaves;
    for (const int ty : {GT_Q4_K, GT_Q5_K, GT_Q6_K}) {
        PgArgs a;
        a.m = m;
        a.acts = ty == GT_Q6_K ? img6 : img45;
        const int tg = ty == GT_Q6_K ? tg6 : tg45, groups = (nt + tg - 1) / tg;
        a.n_tok = nt; a.ld_out = ld_out; a.ld_res = ld_res; a.ld_q = hp_.n_embd;
        int nj = 0, item0 = 0;
        for (int j = 0; j < m.njobs; ++j) {
            if (m.job[j].w.type != ty) continue;
            a.m.job[nj] = m.job[j];
            a.m.job[nj].pair0 = item0;
            item0 += m.gateup ? (m.job[j].w.M + 7) / 8 : (m.job[j].w.M + 15) / 16;   // 8 row pairs per item
            ++nj;
        }
        if (nj == 0) continue;
        a.m.njobs = nj;
        a.n_items = item0;
        const int gx = ((item0 + NW - 1) / NW + 7) / 8 * 8;
        const dim3 grid((unsigned)gx, (unsigned)groups), block(NW * 64);
        const size_t smem = 3 * (size_t)pg_stage_bytes(tg);   // three stage buffers (kernels_pg.h)
#define PG(TYV, TGV, GUV) do { \
            auto kfn = matmul_pg_kernel<TYV, TGV, NW, GUV>; \
            CT_OPTIN_ONCE(kfn, 3 * (size_t)PgStage<TGV>::BYTES); ++g_pg_launches; \
            CT_LAUNCH_DYN(kfn, grid, block, smem, stream_, a.acts, a.m.K, a.n_items, a); } while (0)
#define PG_T(TYV) do { if (tg == 16) { if (m.gateup) PG(TYV, 16, true); else PG(TYV, 16, false); } \
                       else { if (m.gateup) PG(TYV, 32, true); else PG(TYV, 32, false); } } while (0)
        static const char* pg_trace = getenv("CT_AMD_PG_TRACE");   // measurement only: "gate_up" / "qkv" / "wo" / "down": in-kernel stamps of that site
        if (pg_trace && ty == GT_Q4_K && tg == 32 && pg_trace_site_ && !strcmp(pg_trace, pg_trace_site_) && nt > 32) {
            a.m.dbg |= 32; a.m.dbg_sink = (float*)trace_buf_;
            if (m.gateup) { auto kfn = matmul_pg_kernel<GT_Q4_K, 32, NW, true, true>; CT_OPTIN_ONCE(kfn, 3 * (size_t)PgStage<32>::BYTES + 1024); CT_LAUNCH_DYN(kfn, grid, block, smem + 1024, stream_, a.acts, a.m.K, a.n_items, a); }
            else { auto kfn = matmul_pg_kernel<GT_Q4_K, 32, NW, false, true>; CT_OPTIN_ONCE(kfn, 3 * (size_t)PgStage<32>::BYTES + 1024); CT_LAUNCH_DYN(kfn, grid, block, smem + 1024, stream_, a.acts, a.m.K, a.n_items, a); }
            HIP_OK(hipStreamSynchronize(stream_));
            unsigned long long h[120];
            HIP_OK(hipMemcpy(h, trace_buf_, sizeof h, hipMemcpyDeviceToHost));
            fprintf(stderr, "pg_trace %s: start->prologue %llu, total %llu;", pg_trace, h[1] - h[0], h[2] - h[0]);
            for (int b = 0; b < 16 && 8 + 6 * b + 5 < 120; ++b)
                fprintf(stderr, " [b%d issue %llu, l0-3 %llu, midwait %llu, l4-7 %llu, endbar %llu]", b, h[8 + 6 * b + 1] - h[8 + 6 * b], h[8 + 6 * b + 2] - h[8 + 6 * b + 1],
                        h[8 + 6 * b + 3] - h[8 + 6 * b + 2], h[8 + 6 * b + 4] - h[8 + 6 * b + 3], h[8 + 6 * b + 5] - h[8 + 6 * b + 4]);
            fprintf(stderr, "\n");
            continue;
        }
        if (ty == GT_Q4_K) PG_T(GT_Q4_K); else if (ty == GT_Q5_K) PG_T(GT_Q5_K); else PG_T(GT_Q6_K);
#undef PG_T
#undef PG
    }
    return true;
}

// One mat-vec site of a prompt chunk in the order-free form (kernels_mm8.h): the nt activation rows as Q8_K / Q8_0 units in operand order, then one launch per
// weight type over the LAYOUT_M8 records of the site's jobs.
bool Engine::mm8_can(const MatvecArgs& m) const {
    if (!fast_pf_ || !acts8_ || m.njobs < 1) return false;
    const bool b32 = is_block32(m.job[0].w.type);
    for (int j = 0; j < m.njobs; ++j) {
        const DevMat& w = m.job[j].w;
        if (!w.m8 || is_block32(w.type) != b32 || (b32 && w.type != m.job[0].w.type)) return false;
        if (m.gateup ? w.M % 16 != 0 : w.M % 32 != 0) return false;   // whole row tiles only
        const int e = m.job[j].epi;
        if (!(e == EPI_STORE || e == EPI_ADD || e == EPI_ROPE_Q || e == EPI_ROPE_K || e == EPI_V || e == EPI_SILU_MUL || e == EPI_GELU || e == EPI_ADD2)) return false;
    }
    if (m.gateup && (m.njobs != 1 || m.job[0].epi != EPI_SILU_MUL)) return false;
    if ((size_t)m8_steps(m.job[0].w.type, m.K) * ((pf_cap_ + 31) / 32) * kMm8Unit > acts8_bytes_) return false;
    return !b32 ? (m.K % 256 == 0 && m.K <= 32768) : (m.K % 32 == 0 && m.K <= 32768);
}

bool Engine::mm8_matvec(MatvecArgs& m, const float* x, int ldx, int nt, int ld_out, int ld_res, std::string& err) {
    const int ty0 = m.job[0].w.type;
    const bool b32 = is_block32(ty0);
    const int ntt = (nt + 31) / 32, ns = m8_steps(ty0, m.K);
    const dim3 qg((unsigned)(8 * ((nt + 7) / 8))), qb(1024);
    if (b32) {
        if (m.K <= 12288) CT_LAUNCH((mm8_quantize_q80_kernel<12288>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, acts8_, ntt, m.norm_b, nt);
        else CT_LAUNCH((mm8_quantize_q80_kernel<32768>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, acts8_, ntt, m.norm_b, nt);
    } else if (m.pro == PRO_LAYERNORM) {
        if (m.K <= 4096) CT_LAUNCH((mm8_quantize_q8k_kernel<4096, true>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, acts8_, ntt, m.norm_b, nt);
        else if (m.K <= 12288) CT_LAUNCH((mm8_quantize_q8k_kernel<12288, true>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, acts8_, ntt, m.norm_b, nt);
        else { err = "LayerNorm prologue on rows above 12288"; return false; }
    } else if (m.K <= 4096) CT_LAUNCH((mm8_quantize_q8k_kernel<4096, false>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, acts8_, ntt, (const float*)nullptr, nt);
    else if (m.K <= 12288) CT_LAUNCH((mm8_quantize_q8k_kernel<12288, false>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, acts8_, ntt, (const float*)nullptr, nt);
    else CT_LAUNCH((mm8_quantize_q8k_kernel<32768, false>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, acts8_, ntt, (const float*)nullptr, nt);
    for (const int ty : {GT_Q4_K, GT_Q5_K, GT_Q6_K, GT_Q8_0, GT_Q4_0}) {
        Mm8Args a;
        a.m = m;
        a.acts = acts8_;
        a.n_tok = nt; a.ntt = ntt; a.nb = ns; a.ld_out = ld_out; a.ld_res = ld_res; a.ld_q = hp_.n_embd;
        int nj = 0, tile0 = 0;
        for (int j = 0; j < m.njobs; ++j) {
            if (m.job[j].w.type != ty) continue;
            a.m.job[nj] = m.job[j];
            a.m.job[nj].pair0 = tile0;
            tile0 += m.gateup ? m.job[j].w.M / 16 : m.job[j].w.M / 32;
            ++nj;
        }
        if (nj == 0) continue;
        for (int j = nj; j < 3; ++j) a.m.job[j] = MatJob();
        a.m.njobs = nj;
        a.n_tiles = tile0;
        // Shape of the launch: NTT token tiles per wave (they share the unpacked weight operands of a K-step) x KS K-slices per workgroup (8 / KS row
        // tiles, which share the staged activation units).  One workgroup per CU at a time (LDS, registers).  Cost model from in-kernel stamps (7B shapes):
        // a step = the matrix work of a SIMD's two waves + the part of the step's vector-memory issue that does not hide behind it (the CU takes ~45 bytes
        // of requests per cycle: the stage copy grows with KS, the weight records with 1 / NTT); a workgroup = its steps + ~6 000 cycles of prologue and
        // the K-slices' reduction; a launch = rounds of workgroups over the CUs.
        static const int shapes[4][2] = {{2, 1}, {2, 2}, {2, 4}, {1, 8}};
        int best = -1;
        double best_cost = 0.0;
        const size_t unit = (size_t)mm8_unit_bytes(ty);
        for (int si = 0; si < 4; ++si) {
            const int NTTv = shapes[si][0], KSv = shapes[si][1], RW = kMm8Waves / KSv;
            if (mm8_force_ntt_ && (NTTv != mm8_force_ntt_ || KSv != mm8_force_ks_)) continue;
            if (!mm8_force_ntt_ && NTTv > 1 && ntt < 2) continue;   // token tiles that do not exist
            const long long wgs = (long long)((tile0 + RW - 1) / RW) * ((ntt + NTTv - 1) / NTTv);
            const double rounds = (double)((wgs + chip_cus() - 1) / chip_cus());
            const double tile_cycles = ty == GT_Q6_K ? 1200.0 : (is_block32(ty) ? 800.0 : 620.0);
            const double issue = ((double)KSv * NTTv * unit + (double)kMm8Waves * m8_record_bytes(ty)) / 45.0;
            const double step = 2.0 * NTTv * tile_cycles + 0.5 * issue;
            const double cost = rounds * (((ns + KSv - 1) / KSv) * step + 6000.0);
            if (best < 0 || cost < best_cost) { best = si; best_cost = cost; }
        }
        if (best < 0) { err = "CT_AMD_MM8_SHAPE names no instantiated launch shape (2,1 / 2,2 / 2,4 / 1,8)"; return false; }
        const int NTTv = shapes[best][0], KSv = shapes[best][1], RW = kMm8Waves / KSv;
        const dim3 grid((unsigned)((tile0 + RW - 1) / RW), (unsigned)((ntt + NTTv - 1) / NTTv)), block(512);
        const size_t smem = std::max((size_t)2 * KSv * NTTv * unit, (size_t)kMm8Waves * NTTv * 4096);
        ++g_mm8_launches;
#define MM8L(TYV, NV, KV) do { \
            auto kfn = mm8_kernel<TYV, NV, KV>; \
            CT_OPTIN_ONCE(kfn, (size_t)160 * 1024); \
            CT_LAUNCH_DYN(kfn, grid, block, smem, stream_, a.acts, a.nb, a.n_tiles, a); } while (0)
#define MM8T(TYV) do { \
            if (NTTv == 2 && KSv == 1) MM8L(TYV, 2, 1); else if (NTTv == 2 && KSv == 2) MM8L(TYV, 2, 2); \
            else if (NTTv == 2) MM8L(TYV, 2, 4); else MM8L(TYV, 1, 8); } while (0)
        if (ty == GT_Q4_K) MM8T(GT_Q4_K); else if (ty == GT_Q5_K) MM8T(GT_Q5_K); else if (ty == GT_Q6_K) MM8T(GT_Q6_K); else if (ty == GT_Q8_0) MM8T(GT_Q8_0); else MM8T(GT_Q4_0);
#undef MM8T
#undef MM8L
    }
    return true;
}

// One mat-vec site of a prompt chunk: activation images of the nt rows, then the token-batched kernel(s) — kernels_pg.h for
// K-quant weights (f16 matrix cores), kernels_pf.h for Q8_0 / Q4_0 (4x4x4 int8 matrix cores; dot4 for rows above 16384).
bool Engine::pf_matvec(MatvecArgs& m, const float* x, int ldx, int nt, int ld_out, int ld_res, const char* site, double bytes,
                       std::string& err) {
    if (!site_on(site)) return true;
    if (!m.gateup && m.njobs > 1) {   // K-quant and Q8_0 / Q4_0 matrices (or the two 32-block types) at one site: one pass per family
        auto fam = [&](int j) { return m.job[j].w.layout == LAYOUT_G4 ? m.job[j].w.type : -1; };
        bool mixed = false;
        for (int j = 1; j < m.njobs; ++j) mixed = mixed || fam(j) != fam(0);
        if (mixed) {
            int i = 0;
            while (i < m.njobs) {
                int j = i + 1;
                while (j < m.njobs && fam(j) == fam(i)) ++j;
                MatvecArgs g = m;
                g.njobs = j - i;
                int pair0 = 0;
                for (int k = 0; k < g.njobs; ++k) { g.job[k] = m.job[i + k]; g.job[k].pair0 = pair0; pair0 += (g.job[k].w.M + 1) / 2; }
                for (int k = g.njobs; k < 3; ++k) g.job[k] = MatJob();
                g.n_pairs = pair0;
                if (!pf_matvec(g, x, ldx, nt, ld_out, ld_res, site, bytes * (j - i) / m.njobs, err)) return false;
                i = j;
            }
            return true;
        }
    }
    static const char* mm8_sites = getenv("CT_AMD_MM8_SITES");   // experiments: only these sites ("qkv,wo,gate_up,down") take the order-free kernels
    if (mm8_can(m) && (!mm8_sites || strstr(mm8_sites, site))) {
        prof_begin(site, "mm8", bytes);
        const bool ok = mm8_matvec(m, x, ldx, nt, ld_out, ld_res, err);
        prof_end();
        return ok;
    }
    if (nt > kPfChunk) { err = "prompt chunk above 128 tokens at a site without the order-free kernels"; return false; }
    prof_begin(site, "matvec_pf", bytes);
    const dim3 qg((unsigned)nt), qb(1024);
    if (m.job[0].w.layout == LAYOUT_G4) {   // Q8_0 / Q4_0 weights: Q8_0 activation images, kernels_pf.h
        const int Kp = (m.K + 127) & ~127;   // rows in whole groups of four blocks (zero blocks behind a row's end: kernels_pf.h)
        const int aw32 = pf_act_words_q32(Kp);
        // Tokens per workgroup of the matrix-core form: 16 where their images fit the CU's LDS (K <= 9000; gate + up keeps SiLU(gate) in LDS
        // too), 12 at K = 11008, 8 up to K = 16384 — a CU then loads the weights once per 16 / 12 tokens instead of once per 8 (measured on
        // config 3, profiles/r04_q80_chunk_ablations.txt: 8 -> 5 210, 16 / 12 -> 5 610, 32 on eight waves -> 5 410 prompt tok/s).  Rows whose 8
        // images do not fit (K > 16384) take the dot4 form with 4.
        const size_t lds_cap = (size_t)158 * 1024;
        int ntg = 0;
        for (int c : {4, 3, 2}) {
            if (m.gateup && c != 4 && c != 2) continue;
            const size_t need = (size_t)4 * c * aw32 * 4 + (m.gateup ? (size_t)16 * 8 * c * 4 * 4 : 0);
            if (need <= lds_cap && (c == 2 || nt > 4 * (c == 3 ? 2 : c / 2))) { ntg = c; break; }   // short chunks keep more workgroups
        }
        if (ntg != 0) {   // experiments: CT_AMD_PFM_NTG forces a smaller group where that form exists
            static const int forced = env_int("CT_AMD_PFM_NTG", 0);
            const int f = forced ? forced : ntg;
            if (f >= 2 && f <= ntg && (f == 2 || f == 4 || (!m.gateup && f == 3))) ntg = f;
        }
        const int tb = ntg ? 4 * ntg : kPfTokens / 2;
        const bool mfma = ntg != 0;                 // lane sums on v_mfma_i32_4x4x4_16b_i8 (one image per token)
        const int paired = mfma ? 0 : 1;
        if (m.K <= 12288) CT_LAUNCH((pf_quantize_q80_kernel<12288>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, acts_, aw32, m.norm_b, paired);
        else CT_LAUNCH((pf_quantize_q80_kernel<32768>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, acts_, aw32, m.norm_b, paired);
        PfArgs a;
        a.m = m;
        a.m.K = Kp;
        a.acts = acts_; a.act_words = aw32; a.n_tok = nt;
        a.ld_out = ld_out; a.ld_res = ld_res; a.ld_q = hp_.n_embd;
        int item0 = 0;
        for (int j = 0; j < m.njobs; ++j) {
            a.m.job[j].pair0 = m.gateup ? 0 : item0;
            item0 += (m.job[j].w.M + 7) / 8;
        }
        a.m.n_pairs = m.gateup ? (m.job[0].w.M + 7) / 8 : item0;
        const int groups = (nt + tb - 1) / tb;
        const int gx = std::max(1, std::min(chip_cus() / groups, a.m.n_pairs));
        const dim3 grid((unsigned)gx, (unsigned)groups), block(1024);
        const size_t smem = (size_t)tb * aw32 * 4 + (mfma && m.gateup ? (size_t)16 * 8 * ntg * 4 * 4 : 0);
#define PF32(TBV, GUV) do { \
            auto kfn = matvec_pf_kernel<TBV, GUV>; \
            CT_OPTIN_ONCE(kfn, (size_t)150 * 1024); \
            CT_LAUNCH_DYN(kfn, grid, block, smem, stream_, a); } while (0)
#define PFM(GUV, NV) do { \
            auto kfn = matvec_pfm_kernel<GUV, NV>; \
            CT_OPTIN_ONCE(kfn, lds_cap); \
            CT_LAUNCH_DYN(kfn, grid, block, smem, stream_, a); } while (0)
        if (mfma) {
            if (m.gateup) { if (ntg == 4) PFM(true, 4); else PFM(true, 2); }
            else if (ntg == 4) PFM(false, 4); else if (ntg == 3) PFM(false, 3); else PFM(false, 2);
        } else { if (m.gateup) PF32(kPfTokens / 2, true); else PF32(kPfTokens / 2, false); }
#undef PFM
#undef PF32
        prof_end();
        return true;
    }
    bool pg = acts_h_ != nullptr;
    for (int j = 0; j < m.njobs; ++j) pg = pg && is_kquant(m.job[j].w.type) && m.job[j].w.r2;
    if (pg && m.gateup) pg = m.njobs == 1 && m.job[0].w.layout == LAYOUT_R2C4;
    if (!pg) { err = "prompt chunk: this launch shape has no kernel"; return false; }
    pg_trace_site_ = site;
    const bool ok = pg_matvec(m, x, ldx, nt, ld_out, ld_res, err);
    prof_end();
    return ok;
}

#include "engine_steps_llama.h"
#include "engine_steps_falcon.h"
#include "engine_steps_legacy.h"

// A whole-model handle launches nothing in chunk_step that depends on c0 (the cursor lives in d_state_), so the ~300 launches
// of a chunk shape seen before are replayed from a graph: the first use of a shape runs eagerly (it also performs the
// one-time dynamic-LDS opt-ins), the second captures.
bool Engine::run_chunk(int c0, int nt, bool want_logits, std::string& err) {
    // attn_chunk_tile_kernel applies when every position of the chunk lies below 128 AND so does the end of the reference batch its tokens belong to: the
    // V*P dot of a token runs over the positions of its whole batch (masked ones with probability zero, llama.cpp:2373-2378), and the kernel keeps 128
    // positions of V and 160-float probability rows in LDS.  (Round 6: the second condition was missing — a request longer than 128 tokens evaluated as ONE
    // reference batch, batch_size > 128, zeroed probabilities past its row's end in the first chunk and differed from the reference build.)
    {
        const int bs = h_scalars_[3], idx_last = c0 + nt - 1;
        const int end = bs > 0 ? std::min(req_n_, (idx_last / bs + 1) * bs) : req_n_;
        chunk_below_128_ = req_past_ + c0 + nt <= 128 && req_past_ + end <= 128;   // positions of this chunk: [n_past + c0, n_past + c0 + nt)
    }
#ifndef CT_EMU
    if (use_graph_ && !prof_ && !only_site_) {
        // the attention kernel depends on the flag; a pipeline stage (not the whole model) copies its rows from / to the hand-off
        // buffer at offset c0, so its graphs are per (c0, shape) — a request's micro-batches start at the same offsets every time
        const bool whole = l0_ == 0 && l1_ == hp_.n_layer;
        const long long key = ((long long)(whole ? 0 : c0 + 1) << 24) | (long long)(4 * nt + 2 * (chunk_below_128_ ? 1 : 0) + (want_logits ? 1 : 0));
        auto it = chunk_graphs_.find(key);
        if (it == chunk_graphs_.end() && chunk_seen_[key]++ >= 1) {
            hipGraph_t g = nullptr;
            std::unique_lock<std::recursive_mutex> cap(capture_mutex());
            HIP_OK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
            const bool ok = chunk_step(c0, nt, want_logits, err);
            const hipError_t e = hipStreamEndCapture(stream_, &g);
            cap.unlock();
            if (!ok) { if (g) (void)hipGraphDestroy(g); return false; }
            if (e != hipSuccess) { err = std::string("hipStreamEndCapture (chunk) failed: ") + hipGetErrorString(e); return false; }
            hipGraphExec_t ex = nullptr;
            const hipError_t ei = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (ei != hipSuccess) { err = std::string("hipGraphInstantiate (chunk) failed: ") + hipGetErrorString(ei); return false; }
            // a pipeline stage keys its graphs by micro-batch offset too (n_ctx / micro-batch offsets x shapes): bound what stays
            // instantiated — past the cap the LEAST RECENTLY USED graph goes (a long prompt walks its offsets in ascending order pass after
            // pass: evicting the lowest key would destroy and recapture the same graphs on alternating passes).  Launches of the evicted
            // graph may still be in flight on stream_: it is only retired here and destroyed behind the next stream sync (req_wait).
            if (chunk_graphs_.size() >= kMaxChunkGraphs) {
                auto old = chunk_graphs_.begin();
                for (auto jt = chunk_graphs_.begin(); jt != chunk_graphs_.end(); ++jt)
                    if (jt->second.last_use < old->second.last_use) old = jt;
                retired_graphs_.push_back(old->second.exec);
                chunk_seen_.erase(old->first);
                chunk_graphs_.erase(old);
            }
            it = chunk_graphs_.emplace(key, ChunkGraph{ex, 0ull}).first;
        }
        if (it != chunk_graphs_.end()) {
            it->second.last_use = ++chunk_use_clock_;
            HIP_OK(hipGraphLaunch(it->second.exec, stream_));
            return true;
        }
    }
#endif
    return chunk_step(c0, nt, want_logits, err);
}

// One token step is ~6 launches per layer; replaying it from a hipGraph removes the host launch cost (eager goes
// host-bound below ~3 us per kernel — guide "graph-replay-floor").  Two graphs: with and without the lm_head tail.
bool Engine::ensure_graphs(std::string& err) {
#ifndef CT_EMU
    if (graph_step_) return true;
    for (int head = 0; head < 2; ++head) {
        hipGraph_t g = nullptr;
        std::unique_lock<std::recursive_mutex> cap(capture_mutex());
        HIP_OK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
        const bool ok = token_step(head == 1, err);
        hipError_t e = hipStreamEndCapture(stream_, &g);
        cap.unlock();
        if (!ok) { if (g) (void)hipGraphDestroy(g); return false; }
        if (e != hipSuccess) { err = std::string("hipStreamEndCapture failed: ") + hipGetErrorString(e); return false; }
        hipGraphExec_t ex = nullptr;
        HIP_OK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        HIP_OK(hipGraphDestroy(g));
        (head ? graph_step_head_ : graph_step_) = ex;
    }
    // continuation steps of a greedy chain: no embedding launch (the previous head launch left the row, the token id and the cursor),
    // outputs into pair member b
    if (spec_possible()) {
        for (int b = 0; b < 2; ++b) {
            hipGraph_t g = nullptr;
            select_out(b);
            cont_mode_ = true;
            std::unique_lock<std::recursive_mutex> cap(capture_mutex());
            HIP_OK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
            const bool ok = token_step(true, err);
            hipError_t e = hipStreamEndCapture(stream_, &g);
            cap.unlock();
            cont_mode_ = false;
            select_out(0);
            if (!ok) { if (g) (void)hipGraphDestroy(g); return false; }
            if (e != hipSuccess) { err = std::string("hipStreamEndCapture (continuation) failed: ") + hipGetErrorString(e); return false; }
            if (!head_cont_) {   // this graph's head does not prepare the next step (the cursor was not advanced before it): no chains
                (void)hipGraphDestroy(g);
                if (graph_cont_[0]) { (void)hipGraphExecDestroy(graph_cont_[0]); graph_cont_[0] = nullptr; }
                break;
            }
            HIP_OK(hipGraphInstantiate(&graph_cont_[b], g, nullptr, nullptr, 0));
            HIP_OK(hipGraphDestroy(g));
            HIP_OK(hipEventCreateWithFlags(&ev_step_[b], hipEventDisableTiming));
        }
    }
#endif
    (void)err;
    return true;
}

#ifndef CT_EMU
// A whole-model llama / falcon handle whose head launch folds the pick and whose token step advances the cursor before the head.
bool Engine::spec_possible() const {
    return spec_on_ && use_graph_ && l0_ == 0 && head_folds() && fold_on_ == 1 && !dump_dir_ && !hp_.legacy() && tok_embd_.raw;
}

bool Engine::launch_spec(int buf, int pos, std::string& err) {
    HIP_OK(hipGraphLaunch(graph_cont_[buf], stream_));
    HIP_OK(hipEventRecord(ev_step_[buf], stream_));
    spec_inflight_ = true;
    spec_buf_ = buf;
    spec_pos_ = pos;
    ++spec_launched_;
    return true;
}

bool Engine::drain_spec(std::string& err) {
    if (spec_inflight_ || ev_pending_) HIP_OK(hipStreamSynchronize(stream_));
    spec_inflight_ = false;
    ev_pending_ = false;
    return true;
}
#endif

void Engine::disable_resident_forms() {
    if (!fuse_qa_ && !attn_share_) return;
    (void)hipSetDevice(device_);
    (void)hipStreamSynchronize(stream_);   // nothing of the graphs below (a queued continuation step included) is in flight any more
    fuse_qa_ = false;
    attn_share_ = false;
#ifndef CT_EMU
    if (graph_step_) { (void)hipGraphExecDestroy(graph_step_); graph_step_ = nullptr; }
    if (graph_step_head_) { (void)hipGraphExecDestroy(graph_step_head_); graph_step_head_ = nullptr; }
    for (int b = 0; b < 2; ++b) {
        if (graph_cont_[b]) { (void)hipGraphExecDestroy(graph_cont_[b]); graph_cont_[b] = nullptr; }
        if (ev_step_[b]) { (void)hipEventDestroy(ev_step_[b]); ev_step_[b] = nullptr; }
    }
    spec_inflight_ = false;
    ev_pending_ = false;
#endif
    h_scalars_[n_ctx_ + 14] = 0;   // (a step that was still running when the word was read may have raised it again)
}

bool Engine::resident_timeout() {
    if (!h_scalars_ || h_scalars_[n_ctx_ + 14] == 0) return false;
    const bool had = fuse_qa_ || attn_share_;
    disable_resident_forms();
    h_scalars_[n_ctx_ + 14] = 0;
    if (had) {
        static bool said = false;
        if (!said) {
            said = true;
            fprintf(stderr, "ctransformers_amd: a launch that needs its whole grid resident gave up (another process or handle on this GPU?); "
                            "this handle continues with the forms that need no residency, the request is evaluated again\n");
        }
    }
    return true;
}

bool Engine::eval(const int* tokens, int n, int n_past, std::string& err, int batch) {
    if (l0_ != 0 || l1_ != hp_.n_layer) { err = "this handle is a pipeline stage: use eval_stage"; return false; }
#ifndef CT_EMU
    const bool armed = greedy_armed_;
    greedy_armed_ = false;   // one eval per sample(): a second eval without a pick in between is not a greedy chain
    if (spec_inflight_) {
        spec_inflight_ = false;
        if (n == 1 && tokens && tokens[0] == last_pick_ && n_past == spec_pos_ && armed) {
            // the queued continuation step IS this eval: its outputs become the committed ones; the next one goes behind it
            HIP_OK(hipSetDevice(device_));
            const int b = spec_buf_;
            ++spec_hits_;
            if (n_past + 1 < n_ctx_ && !launch_spec(b ^ 1, n_past + 1, err)) return false;   // the guess after this one, before waiting
            HIP_OK(hipEventSynchronize(ev_step_[b]));
            HIP_OK(hipGetLastError());
            if (!resident_timeout()) {
                cur_buf_ = b;
                have_logits_ = true;
                outputs_on_host_ = false;
                last_token_ = tokens[0];
                last_pos_ = n_past;
                last_pick_ = h_scalars_[n_ctx_ + 12 + b];
                return true;
            }
            // the queued step (or the one behind it) gave up on residency: its outputs are not valid — this token is evaluated again below, the normal way
            --spec_hits_;
            ++resident_replays_;
        }
        // not what was guessed: the queued step finishes first (stream order); what it wrote is a KV position nobody has evaluated
        // and the other output buffer
    }
    spec_want_ = armed && n == 1 && n_past + 1 < n_ctx_ && spec_possible();
#endif
    const long long replays0 = resident_replays_;
    if (eval_stage(tokens, n, n_past, nullptr, nullptr, err, batch)) return true;
    if (resident_replays_ == replays0) return false;   // some other failure
    // req_wait switched the handle to the forms that need no residency and drained the stream: the same request once more (every KV row it wrote is
    // written again; a continuation step is not queued this time)
#ifndef CT_EMU
    spec_want_ = false;
#endif
    err.clear();
    return eval_stage(tokens, n, n_past, nullptr, nullptr, err, batch);
}

bool Engine::req_begin(const int* tokens, int n, int n_past, int batch, std::string& err, bool upload) {
    if (n <= 0) { err = "empty request"; return false; }
    if (n_past < 0 || n_past + n > n_ctx_) { err = "eval past the context window"; return false; }
    if (l0_ == 0 && !tokens) { err = "first stage needs token ids"; return false; }
    HIP_OK(hipSetDevice(device_));
    for (int i = 0; i < n; ++i) {
        const int tk = (l0_ == 0 && tokens) ? tokens[i] : 0;
        if (tk < 0 || tk >= hp_.n_vocab) { err = "token id out of range"; return false; }
        h_scalars_[4 + i] = tk;
    }
    h_scalars_[0] = 0;           // step
    h_scalars_[1] = n_past;      // position of the first token of this request
    h_scalars_[2] = n_past + n;  // end of the eval; with [3] the attention kernels derive the reference batch each token belongs to
    h_scalars_[3] = batch > 0 && batch < n ? batch : 0;   // 0: the reference runs these n tokens as ONE batch
    if (env_int("CT_AMD_DBG_ONE_BATCH", 0)) h_scalars_[3] = 0;   // tests of the tests: ignore the batch structure on purpose
    if (upload) HIP_OK(hipMemcpyAsync(d_state_, &h_scalars_[0], (size_t)(4 + n) * 4, hipMemcpyHostToDevice, stream_));   // cursor + token ids
    req_n_ = n;
    req_past_ = n_past;
    return true;
}

bool Engine::req_range(int c0, int nt, bool last_of_request, std::string& err) {
    if (nt <= 0) return true;
    if (c0 < 0 || c0 + nt > req_n_) { err = "range outside the request"; return false; }
    HIP_OK(hipSetDevice(device_));
    const int n = c0 + nt;
    int done = c0;
    if (pf_ok_ && nt >= pf_min_ && !dump_dir_) {   // prompt chunks: kPfChunk tokens per pass over the weights
        while (n - done >= pf_min_) {
            const int k = std::min(pf_chunk_, n - done);
            if (!run_chunk(done, k, last_of_request && done + k == n, err)) return false;
            done += k;
            chunk_tokens_ += k;
        }
    }
#ifndef CT_EMU
    if (use_graph_) {
        if (done < n && !ensure_graphs(err)) return false;
        for (int i = done; i < n; ++i) HIP_OK(hipGraphLaunch(last_of_request && i == n - 1 ? graph_step_head_ : graph_step_, stream_));
    } else
#endif
    {
        for (int i = done; i < n; ++i)
            if (!token_step(last_of_request && i == n - 1, err)) return false;
    }
    return true;
}

bool Engine::req_logits(std::string& err) {
    if (l1_ != hp_.n_layer) { err = "logits live on the last stage"; return false; }
    HIP_OK(hipSetDevice(device_));
    if (!head_folded_) {   // (a folding head launch — every head launch of a handle folds or none does — wrote the pick into the pinned word itself)
        CT_LAUNCH(argmax_first_kernel, dim3(1), dim3(1024), stream_, (const float*)d_logits_, hp_.n_vocab, d_argmax_);
        HIP_OK(hipMemcpyAsync(&h_scalars_[n_ctx_ + 12], d_argmax_, 4, hipMemcpyDeviceToHost, stream_));
    }
    outputs_on_host_ = false;
    cur_buf_ = 0;
#ifndef CT_EMU
    if (spec_want_) {   // a greedy chain: wait for THIS step's event, the guessed next step runs behind it
        spec_want_ = false;
        if (!ensure_graphs(err)) return false;
        if (graph_cont_[1]) {
            HIP_OK(hipEventRecord(ev_step_[0], stream_));
            ev_pending_ = true;
            if (!launch_spec(1, req_past_ + req_n_, err)) return false;
        }
    }
#endif
    return true;
}

void Engine::fetch_outputs() {
    if (outputs_on_host_ || !have_logits_) return;
    (void)hipSetDevice(device_);
    // (on the handle's own stream: a copy on the legacy stream would collide with a capture another thread has open, engine.h:capture_mutex)
    (void)hipMemcpyAsync(h_logits_, d_logits2_[cur_buf_], ((size_t)hp_.n_vocab + hp_.n_embd) * 4, hipMemcpyDeviceToHost, stream_);
    (void)hipStreamSynchronize(stream_);
    outputs_on_host_ = true;
}

bool Engine::req_wait(int n, int n_past, std::string& err) {
    HIP_OK(hipSetDevice(device_));
#ifndef CT_EMU
    if (ev_pending_) {   // a speculative step is queued behind this eval: wait for the eval, not for the stream
        ev_pending_ = false;
        HIP_OK(hipEventSynchronize(ev_step_[0]));
        HIP_OK(hipGetLastError());
    } else {
        HIP_OK(hipStreamSynchronize(stream_));
        HIP_OK(hipGetLastError());
        for (hipGraphExec_t g : retired_graphs_) (void)hipGraphExecDestroy(g);   // evicted chunk graphs: nothing of them is in flight now
        retired_graphs_.clear();
    }
#else
    HIP_OK(hipStreamSynchronize(stream_));
#endif
    if (dbg_qa_timeout_ > 0 && (fuse_qa_ || attn_share_) && --dbg_qa_timeout_ == 0) h_scalars_[n_ctx_ + 14] = 1;   // tests: as if a sweep had given up
    if (resident_timeout()) {
        ++resident_replays_;
        err = "a decode launch whose workgroups hand data to each other (fused QKV + attention, shared score rows) timed out waiting for workgroups of "
              "its own grid (is the device shared?): the request is evaluated again with the forms that need no residency";
        return false;
    }
    have_logits_ = l1_ == hp_.n_layer;
    last_token_ = h_scalars_[4 + n - 1];
    last_pos_ = n_past + n - 1;
    last_pick_ = h_scalars_[n_ctx_ + 12];
    return true;
}

bool Engine::eval_stage(const int* tokens, int n, int n_past, const float* x_in_dev, float* x_out_dev, std::string& err, int batch) {
    if (n <= 0) return true;
    if (l0_ > 0 && !x_in_dev) { err = "stage with layer_begin > 0 needs x_in"; return false; }
    if (l1_ < hp_.n_layer && !x_out_dev) { err = "stage with layer_end < n_layer needs x_out"; return false; }
    if (!req_begin(tokens, n, n_past, batch, err)) return false;
    const size_t xbytes = (size_t)n * hp_.n_embd * sizeof(float);
    if (l0_ > 0) HIP_OK(hipMemcpyAsync(xio_, x_in_dev, xbytes, hipMemcpyDeviceToDevice, stream_));
    if (!req_range(0, n, true, err)) return false;
    if (l1_ == hp_.n_layer) {
        if (!req_logits(err)) return false;
    } else {
        HIP_OK(hipMemcpyAsync(x_out_dev, xio_, xbytes, hipMemcpyDeviceToDevice, stream_));
    }
    return req_wait(n, n_past, err);
}

void Engine::prof_begin(const char* site, const char* kernel, double bytes) {
#ifndef CT_EMU
    if (!prof_) return;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, stream_);
    prof_->push_back(ProfRec{site, kernel, bytes, (void*)e0, (void*)e1});
#else
    (void)site; (void)kernel; (void)bytes;
#endif
}
void Engine::prof_end() {
#ifndef CT_EMU
    if (!prof_) return;
    (void)hipEventRecord((hipEvent_t)prof_->back().e1, stream_);
#endif
}

bool Engine::trace_site(const char* site, unsigned long long* out, int n, std::string& err) {
#ifndef CT_EMU
    HIP_OK(hipSetDevice(device_));
    if (!drain_spec(err)) return false;
    if (last_pos_ < 0) { err = "nothing evaluated yet"; return false; }
    h_scalars_[0] = 0; h_scalars_[1] = last_pos_; h_scalars_[2] = last_pos_ + 1; h_scalars_[4] = last_token_;
    HIP_OK(hipMemcpyAsync(d_tokens_, &h_scalars_[4], 4, hipMemcpyHostToDevice, stream_));
    HIP_OK(hipMemcpyAsync(d_state_, &h_scalars_[0], 12, hipMemcpyHostToDevice, stream_));
    trace_site_ = site;
    const bool ok = token_step(true, err);
    trace_site_ = nullptr;
    if (!ok) return false;
    HIP_OK(hipStreamSynchronize(stream_));
    HIP_OK(hipMemcpy(out, trace_buf_, (size_t)std::min(n, 512) * 8, hipMemcpyDeviceToHost));
    return true;
#else
    (void)site; (void)out; (void)n; err = "needs the HIP build"; return false;
#endif
}

int Engine::read_stamps(unsigned long long* out, int max) {
    if (!stamps_) return 0;
    (void)hipSetDevice(device_);
    (void)hipStreamSynchronize(stream_);
    unsigned long long n = 0;
    (void)hipMemcpy(&n, stamps_, 8, hipMemcpyDeviceToHost);
    const int k = (int)std::min<unsigned long long>(n, (unsigned long long)max);
    (void)hipMemcpy(out, stamps_ + 1, (size_t)k * 8, hipMemcpyDeviceToHost);
    (void)hipMemset(stamps_, 0, 8);
    return k;
}

int Engine::debug_read_kv(int layer, uint16_t* k, uint16_t* v) {
    if (layer < l0_ || layer >= l1_ || !kcache_ || !vcache_) return -1;
    (void)hipSetDevice(device_);
    (void)hipStreamSynchronize(stream_);
    const int G = hp_.n_embd_gqa();
    (void)hipMemcpy(k, kcache_ + (size_t)(layer - l0_) * n_ctx_ * G, (size_t)n_ctx_ * G * 2, hipMemcpyDeviceToHost);
    (void)hipMemcpy(v, vcache_ + (size_t)(layer - l0_) * v_stride_ * G, (size_t)v_stride_ * G * 2, hipMemcpyDeviceToHost);
    return v_stride_;
}

// Test hook: the attention output rows of the last chunk launched (last layer of this stage), n_tok x n_embd floats.
int Engine::debug_read_attn_out(float* dst, int n_tok) {
    if (n_tok == 0 && attn_out_) {   // the last token step's row (its last layer's attention output)
        (void)hipSetDevice(device_);
        (void)hipStreamSynchronize(stream_);
        (void)hipMemcpy(dst, attn_out_, (size_t)hp_.n_embd * 4, hipMemcpyDeviceToHost);
        return hp_.n_embd;
    }
    if (!attn_out_b_ || n_tok < 1 || n_tok > pf_cap_) return -1;
    (void)hipSetDevice(device_);
    (void)hipStreamSynchronize(stream_);
    (void)hipMemcpy(dst, attn_out_b_, (size_t)n_tok * hp_.n_embd * 4, hipMemcpyDeviceToHost);
    return hp_.n_embd;
}

bool Engine::decode_burst(int n, double* us_per_token, std::string& err) {
#ifndef CT_EMU
    HIP_OK(hipSetDevice(device_));
    if (last_pos_ < 0 || !use_graph_) { err = "decode_burst: nothing evaluated yet, or graphs are off"; return false; }
    if (last_pos_ + 1 + n >= n_ctx_ || n < 1 || n > 100) { err = "decode_burst: would run past the context"; return false; }
    if (!drain_spec(err)) return false;
    if (!ensure_graphs(err)) return false;
    // the cursor as the next eval would set it; the token ids are whatever the last request left in d_tokens_ (timing only) — with
    // continuation graphs the burst IS the greedy chain (each head launch hands the next step its token)
    h_scalars_[0] = 0; h_scalars_[1] = last_pos_ + 1; h_scalars_[2] = last_pos_ + 1 + (graph_cont_[0] ? 1 : n); h_scalars_[3] = 0;
    HIP_OK(hipMemcpyAsync(d_state_, &h_scalars_[0], 16, hipMemcpyHostToDevice, stream_));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipGraphLaunch(graph_step_head_, stream_));   // one untimed step: the burst starts with the GPU busy
    HIP_OK(hipEventRecord(e0, stream_));
    for (int i = 1; i < n; ++i) HIP_OK(hipGraphLaunch(graph_cont_[0] ? graph_cont_[i & 1] : graph_step_head_, stream_));
    HIP_OK(hipEventRecord(e1, stream_));
    HIP_OK(hipStreamSynchronize(stream_));
    float ms = 0.0f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *us_per_token = (double)ms * 1e3 / (double)(n - 1);
    return true;
#else
    (void)n; (void)us_per_token; err = "needs the HIP build"; return false;
#endif
}

bool Engine::profile_decode(int iters, std::vector<LaunchStat>& out, std::string& err) {
    out.clear();
#ifndef CT_EMU
    HIP_OK(hipSetDevice(device_));
    if (!drain_spec(err)) return false;
    if (last_pos_ < 0) { err = "profile_decode: nothing evaluated yet"; return false; }
    std::vector<ProfRec> recs;
    for (int it = 0; it < iters; ++it) {
        h_scalars_[0] = 0;
        h_scalars_[1] = last_pos_;
        h_scalars_[2] = last_pos_ + 1;
        h_scalars_[4] = last_token_;
        HIP_OK(hipMemcpyAsync(d_tokens_, &h_scalars_[4], 4, hipMemcpyHostToDevice, stream_));
        HIP_OK(hipMemcpyAsync(d_state_, &h_scalars_[0], 12, hipMemcpyHostToDevice, stream_));
        prof_ = &recs;
        const bool ok = token_step(true, err);
        prof_ = nullptr;
        if (!ok) return false;
        HIP_OK(hipStreamSynchronize(stream_));
    }
    // Second view, "sweep": for each launch site, the kernels of ALL layers back to back between ONE pair of events
    // (different weights every launch, so HBM-cold like the real step; no per-launch event, so the ~6.6 us eager
    // event floor is paid once per sweep, not per kernel).  This is the per-launch cost inside a graph replay.
    static const char* kSweep[] = {"qkv", "attn_fused", "wo", "gate_up", "down", "lm_head"};
    struct Sweep { const char* site; float ms; int launches; };
    std::vector<Sweep> sweeps;
    for (const char* site : kSweep) {
        hipEvent_t e0, e1;
        HIP_OK(hipEventCreate(&e0));
        HIP_OK(hipEventCreate(&e1));
        float ms_tot = 0.0f;
        int n_tot = 0;
        for (int it = 0; it < iters; ++it) {
            h_scalars_[0] = 0; h_scalars_[1] = last_pos_; h_scalars_[2] = last_pos_ + 1; h_scalars_[4] = last_token_;
            HIP_OK(hipMemcpyAsync(d_tokens_, &h_scalars_[4], 4, hipMemcpyHostToDevice, stream_));
            HIP_OK(hipMemcpyAsync(d_state_, &h_scalars_[0], 12, hipMemcpyHostToDevice, stream_));
            if (!strcmp(site, "lm_head")) {   // one launch per step: evict its weights from the memory-side cache first
                only_site_ = "gate_up";
                const bool okf = token_step(true, err);
                only_site_ = nullptr;
                if (!okf) return false;
            }
            only_site_ = site;
            site_launches_ = 0;
            HIP_OK(hipEventRecord(e0, stream_));
            const bool ok = token_step(true, err);
            HIP_OK(hipEventRecord(e1, stream_));
            only_site_ = nullptr;
            if (!ok) return false;
            HIP_OK(hipStreamSynchronize(stream_));
            float ms = 0.0f;
            HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            ms_tot += ms;
            n_tot += site_launches_;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (n_tot > 0) sweeps.push_back(Sweep{site, ms_tot, n_tot});
    }
    for (auto& r : recs) {
        float ms = 0.0f;
        HIP_OK(hipEventElapsedTime(&ms, (hipEvent_t)r.e0, (hipEvent_t)r.e1));
        (void)hipEventDestroy((hipEvent_t)r.e0);
        (void)hipEventDestroy((hipEvent_t)r.e1);
        bool found = false;
        for (auto& o : out)
            if (!strcmp(o.site, r.site)) { o.ms += ms; o.bytes += r.bytes; o.launches++; found = true; break; }
        if (!found) out.push_back(LaunchStat{r.site, r.kernel, r.bytes, (double)ms, 1});
    }
    static std::vector<std::string> names;   // storage behind the "<site>@sweep" labels
    names.clear();
    names.reserve(sweeps.size());
    for (auto& w : sweeps) {
        double bytes = 0.0;
        for (auto& o : out)
            if (!strcmp(o.site, w.site)) bytes = o.bytes / o.launches * w.launches;
        names.push_back(std::string(w.site) + "@sweep");
        out.push_back(LaunchStat{names.back().c_str(), "sweep", bytes, (double)w.ms, w.launches});
    }
    return true;
#else
    (void)iters;
    err = "profiling needs the HIP build";
    return false;
#endif
}


}  // namespace ctamd
