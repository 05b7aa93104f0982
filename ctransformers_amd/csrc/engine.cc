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
#include "kernels_q32.h"
#include "kernels_pf.h"
#include "kernels_pg.h"

namespace ctamd {

#define HIP_OK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            err = std::string(#expr) + " failed: " + hipGetErrorString(e_);                                 \
            return false;                                                                                    \
        }                                                                                                    \
    } while (0)

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

Engine::~Engine() { free_all(); }

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
    if (graph_step_) (void)hipGraphExecDestroy(graph_step_);
    if (graph_step_head_) (void)hipGraphExecDestroy(graph_step_head_);
    graph_step_ = graph_step_head_ = nullptr;
    for (auto& kv : chunk_graphs_) (void)hipGraphExecDestroy(kv.second);
    chunk_graphs_.clear();
#endif
    if (stream_) (void)hipStreamDestroy(stream_);
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

// ---------------------------------------------------------------------------------------------------------------------
// Load-time repack of file-layout blocks into per-row byte planes (see quant.h).  Values are untouched.
// ---------------------------------------------------------------------------------------------------------------------
static void parallel_rows(int M, const std::function<void(int, int)>& fn) {
    const int nt = std::max(1, std::min<int>(16, (int)std::thread::hardware_concurrency()));
    if (M < 4 * nt) { fn(0, M); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) {
        const int r0 = (int)((long long)M * t / nt), r1 = (int)((long long)M * (t + 1) / nt);
        th.emplace_back([=, &fn] { fn(r0, r1); });
    }
    for (auto& t : th) t.join();
}

// One file-layout K-quant block -> slot `r` (0..7) of a LAYOUT_R2C4 record (quant.h); the 6-bit scale/min
// field of Q4_K / Q5_K headers is re-encoded as four 24-bit groups (reference packing: k_quants.c:306-314).
CT_HD static inline void place_kblock(int type, uint8_t* rp, int r, const uint8_t* blk) {
    uint8_t hdr[16];
    if (type != GT_Q6_K) {
        memcpy(hdr, blk, 16);
        const uint8_t* q = blk + 4;
        uint8_t sc[8], mn[8];
        for (int jj = 0; jj < 8; ++jj) {
            if (jj < 4) { sc[jj] = q[jj] & 63; mn[jj] = q[jj + 4] & 63; }
            else { sc[jj] = (q[jj + 4] & 0xF) | ((q[jj - 4] >> 6) << 4); mn[jj] = (q[jj + 4] >> 4) | ((q[jj] >> 6) << 4); }
        }
        for (int cc = 0; cc < 4; ++cc) {
            const uint32_t g24 = sc[2 * cc] | (sc[2 * cc + 1] << 6) | (mn[2 * cc] << 12) | (mn[2 * cc + 1] << 18);
            hdr[4 + 3 * cc] = g24 & 0xFF; hdr[5 + 3 * cc] = (g24 >> 8) & 0xFF; hdr[6 + 3 * cc] = (g24 >> 16) & 0xFF;
        }
    }
    if (type == GT_Q4_K) {
        memcpy(rp + r * 16, hdr, 16);
        memcpy(rp + 128 + r * 128, blk + 16, 128);
    } else if (type == GT_Q5_K) {
        memcpy(rp + r * 16, hdr, 16);
        memcpy(rp + 128 + r * 32, blk + 16, 32);
        memcpy(rp + 384 + r * 128, blk + 48, 128);
    } else {  // GT_Q6_K: scales and d as in the file, the quants unpacked-ready (quant.h:r2c4_record_bytes)
        memcpy(rp + r * 2, blk + 208, 2);
        memcpy(rp + 16 + r * 16, blk + 192, 16);
        uint8_t* q = rp + 144 + r * 256;
        for (int p = 0; p < 4; ++p)
            for (int l = 0; l < 8; ++l) {
                uint32_t W, H;
                memcpy(&W, blk + 4 * (8 * p + l), 4);                    // ql bytes 32p + 4l ..: low nibbles vector va, high nibbles vb
                memcpy(&H, blk + 128 + 4 * (8 * (p >> 1) + l), 4);       // qh bytes 32(p >> 1) + 4l ..
                H >>= 2 * (p & 1);
                auto q6 = [&](int wsh, int hsh) { return ((W >> wsh) & 0xFu) | (((H >> hsh) & 3u) << 4); };
                const uint32_t A = (q6(0, 0) << 3) | (q6(16, 16) << 19) | (q6(8, 8) << 9) | (q6(24, 24) << 25);
                const uint32_t B = (q6(4, 4) << 3) | (q6(20, 20) << 19) | (q6(12, 12) << 9) | (q6(28, 28) << 25);
                memcpy(q + (p * 8 + l) * 8, &A, 4);
                memcpy(q + (p * 8 + l) * 8 + 4, &B, 4);
            }
    }
}

// One file-layout K-quant block -> slot `r` = 4 * row + c of a LAYOUT_L9 record (quant.h): the bytes each lane
// (32 * row + 4 * l + c) of the decode wave consumes are contiguous, the 6-bit scales / mins of Q4_K / Q5_K sit word-aligned.
CT_HD static inline void place_kblock9(int type, uint8_t* rp, int r, const uint8_t* blk) {
    const int row = r >> 2, c = r & 3;
    if (type != GT_Q6_K) {
        const uint8_t* q = blk + 4;
        uint32_t sc[8], mn[8];
        for (int jj = 0; jj < 8; ++jj) {
            if (jj < 4) { sc[jj] = q[jj] & 63; mn[jj] = q[jj + 4] & 63; }
            else { sc[jj] = (q[jj + 4] & 0xF) | ((q[jj - 4] >> 6) << 4); mn[jj] = (q[jj + 4] >> 4) | ((q[jj] >> 6) << 4); }
        }
        const uint32_t W1 = sc[0] | (sc[1] << 6) | (sc[2] << 12) | (sc[3] << 18) | (sc[4] << 24) | ((mn[7] & 3u) << 30);
        const uint32_t W2 = ((mn[7] >> 2) & 3u) | (sc[5] << 2) | (sc[6] << 8) | (sc[7] << 14) | (mn[5] << 20) | (mn[6] << 26);
        const uint32_t W3 = ((mn[7] >> 4) & 3u) | (mn[0] << 2) | (mn[1] << 8) | (mn[2] << 14) | (mn[3] << 20) | (mn[4] << 26);
        uint8_t* hdr = rp + (type == GT_Q4_K ? 1024 : 1280) + r * 16;
        memcpy(hdr, blk, 4);
        memcpy(hdr + 4, &W1, 4); memcpy(hdr + 8, &W2, 4); memcpy(hdr + 12, &W3, 4);
        const uint8_t* qs = blk + (type == GT_Q4_K ? 16 : 48);
        for (int l = 0; l < 8; ++l) {
            const int lane = 32 * row + 4 * l + c;
            for (int j = 0; j < 4; ++j) memcpy(rp + lane * 16 + 4 * j, qs + 32 * j + 4 * l, 4);
            if (type == GT_Q5_K) memcpy(rp + 1024 + lane * 4, blk + 16 + 4 * l, 4);
        }
    } else {  // GT_Q6_K: ql[128] | qh[64] | scales[16] | d
        for (int l = 0; l < 8; ++l) {
            const int lane = 32 * row + 4 * l + c;
            for (int n = 0; n < 2; ++n) {
                memcpy(rp + lane * 16 + 8 * n, blk + 64 * n + 4 * l, 4);
                memcpy(rp + lane * 16 + 8 * n + 4, blk + 64 * n + 32 + 4 * l, 4);
                memcpy(rp + 1024 + lane * 8 + 4 * n, blk + 128 + 32 * n + 4 * l, 4);
            }
        }
        for (int v = 0; v < 8; ++v) { rp[1536 + r * 16 + v] = blk[192 + 2 * v]; rp[1536 + r * 16 + 8 + v] = blk[192 + 2 * v + 1]; }
        memcpy(rp + 1664 + r * 2, blk + 208, 2);
    }
}

// One file-layout Q8_0 / Q4_0 block -> its place in a LAYOUT_L9 record (quant.h): row (0 / 1) of the pair, block i (0..15) of the record.
CT_HD static inline void place_block32_l9(int type, uint8_t* rp, int row, int i, const uint8_t* blk) {
    const int t = i >> 2, c = i & 3;
    if (type == GT_Q8_0) {
        for (int l = 0; l < 8; ++l) memcpy(rp + (size_t)(32 * row + 4 * l + c) * 16 + 4 * t, blk + 2 + 4 * l, 4);
        memcpy(rp + 1024 + (row * 4 + c) * 8 + 2 * t, blk, 2);
    } else {
        for (int l = 0; l < 4; ++l) memcpy(rp + (size_t)((row * 4 + l) * 4 + c) * 16 + 4 * t, blk + 2 + 4 * l, 4);
        memcpy(rp + 512 + (row * 4 + c) * 8 + 2 * t, blk, 2);
    }
}

// GPU placement of the 32-block types: one thread per (unit, record, row, block) slot of the arena (cleared first).
__global__ void __launch_bounds__(256) repack_l9b_kernel(int type, const uint8_t* __restrict__ sa, const uint8_t* __restrict__ sb,
                                                         uint8_t* __restrict__ dst, int M, int nb, int n_units) {
    const int spu = (nb + 15) / 16, bb = ggml_block_bytes(type), rec = l9_record_bytes(type);
    const long long n = (long long)n_units * spu * 32;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int bi = (int)(i & 15), rr = (int)((i >> 4) & 1);
        const long long us = i >> 5;
        const int s = (int)(us % spu), u = (int)(us / spu);
        const int row = sb ? u : 2 * u + rr, b = 16 * s + bi;
        if (row >= M || b >= nb) continue;
        place_block32_l9(type, dst + (size_t)us * rec, rr, bi, ((sb && rr) ? sb : sa) + ((size_t)row * nb + b) * bb);
    }
}

// The same placement on the GPU: one thread per block slot of the arena, reading the tensor in FILE layout from the staged copy of
// the model file (stage_file).  `sb` != null: fused gate/up (unit u = row u of sa and of sb), else unit u = rows 2u, 2u + 1 of sa.
// L9: place_kblock9 (decode arena) instead of place_kblock (prompt-chunk arena).
template <int TYPE, bool L9 = false>
__global__ void __launch_bounds__(256) repack_r2c4_kernel(const uint8_t* __restrict__ sa, const uint8_t* __restrict__ sb,
                                                          uint8_t* __restrict__ dst, int M, int nb, int n_units) {
    constexpr int type = TYPE;
    const int spu = (nb + 3) / 4, bb = ggml_block_bytes(type), rec = L9 ? l9_record_bytes(type) : r2c4_record_bytes(type);
    const long long n = (long long)n_units * spu * 8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int slot = (int)(i & 7), rr = slot >> 2, cc = slot & 3;
        const long long us = i >> 3;
        const int s = (int)(us % spu), u = (int)(us / spu);
        const int row = sb ? u : 2 * u + rr, b = 4 * s + cc;
        if (row >= M || b >= nb) continue;   // zero slot (the arena is cleared first)
        const uint8_t* blk = ((sb && rr) ? sb : sa) + ((size_t)row * nb + b) * bb;
        if constexpr (L9) place_kblock9(type, dst + (size_t)us * rec, slot, blk);
        else place_kblock(type, dst + (size_t)us * rec, slot, blk);
    }
}

// LAYOUT_G4 (Q8_0 / Q4_0, quant.h) from the staged file copy: one thread per (tile, group of four blocks, row, block).
__global__ void __launch_bounds__(256) repack_g4_kernel(int q8, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int M, int nb) {
    const int ng = nb / 4, bb = q8 ? 34 : 18, rec = q8 ? 1088 : 576, dbase = q8 ? 1024 : 512, nl = q8 ? 8 : 4;
    const long long n = (long long)((M + 7) / 8) * ng * 32;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int bi = (int)(i & 3), r = (int)((i >> 2) & 7);
        const long long tg = i >> 5;
        const int g = (int)(tg % ng), tl = (int)(tg / ng), row = tl * 8 + r;
        if (row >= M) continue;   // zero rows (the buffer is cleared first)
        const uint8_t* blk = src + ((size_t)row * nb + (size_t)g * 4 + bi) * bb;
        uint8_t* rp = dst + (size_t)tg * rec;
        memcpy(rp + dbase + r * 8 + bi * 2, blk, 2);
        for (int l = 0; l < nl; ++l) memcpy(rp + (r * nl + l) * 16 + bi * 4, blk + 2 + 4 * l, 4);
    }
}

// Load pipeline, stage 1: the byte range of the mapping that holds the tensors in `need` -> device memory, unchanged.  pread() by
// worker threads straight into pinned slots (no page faults on the mapping, no pageable bounce inside the runtime), one async copy
// per slot; reading slot k + 1 overlaps the copy of slot k.
bool Engine::stage_file(const GgufFile& f, const std::vector<const GgufTensor*>& need, std::string& err) {
    file_lo_ = file_hi_ = nullptr;
    for (const GgufTensor* t : need) {
        if (!t) continue;
        if (!file_lo_ || t->data < file_lo_) file_lo_ = t->data;
        if (!file_hi_ || t->data + t->nbytes > file_hi_) file_hi_ = t->data + t->nbytes;
    }
    if (!file_lo_) return true;
    const size_t total = (size_t)(file_hi_ - file_lo_);
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc((void**)&dev_file_, total + 256);
    if (e != hipSuccess) {
        // no room for the file-layout copy beside the repacked arenas: the host repack path (pageable reads of the mapping, host-side
        // placement, one copy per arena) takes over — slower to load, same result
        (void)hipGetLastError();
        dev_file_ = nullptr;
        return true;
    }
#ifdef CT_EMU
    memcpy(dev_file_, file_lo_, total);
#else
    constexpr int NSLOT = 4;
    constexpr size_t SLOT = (size_t)64 << 20;
    uint8_t* pin[NSLOT] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[NSLOT] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t cs = nullptr;
    bool ok = true;
    std::string why;
    auto fail = [&](const char* what, hipError_t code) { ok = false; why = std::string(what) + " failed: " + hipGetErrorString(code); };
    hipError_t rc = hipStreamCreate(&cs);
    if (rc != hipSuccess) fail("hipStreamCreate", rc);
    for (int k = 0; k < NSLOT && ok; ++k) {
        if ((rc = hipHostMalloc((void**)&pin[k], SLOT)) != hipSuccess) fail("hipHostMalloc", rc);
        else if ((rc = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming)) != hipSuccess) fail("hipEventCreateWithFlags", rc);
    }
    const off_t base = (off_t)(file_lo_ - f.map_base());
    const int fd = f.fd();
    int k = 0;
    for (size_t off = 0; off < total && ok; off += SLOT, k = (k + 1) % NSLOT) {
        const size_t len = std::min(SLOT, total - off);
        if (off >= NSLOT * SLOT && (rc = hipEventSynchronize(ev[k])) != hipSuccess) { fail("hipEventSynchronize", rc); break; }   // the copy that last used this slot
        uint8_t* dstp = pin[k];
        std::atomic<bool> good(true);
        parallel_rows((int)((len + (1 << 20) - 1) >> 20), [&](int m0, int m1) {   // 1 MB pieces
            for (int mi = m0; mi < m1; ++mi) {
                size_t o = (size_t)mi << 20;
                const size_t end = std::min(len, o + ((size_t)1 << 20));
                while (o < end) {
                    const ssize_t r = pread(fd, dstp + o, end - o, base + (off_t)(off + o));
                    if (r <= 0) { good = false; return; }
                    o += (size_t)r;
                }
            }
        });
        if (!good) { ok = false; why = "reading the model file failed"; break; }
        if ((rc = hipMemcpyAsync(dev_file_ + off, dstp, len, hipMemcpyHostToDevice, cs)) != hipSuccess) { fail("hipMemcpyAsync", rc); break; }
        if ((rc = hipEventRecord(ev[k], cs)) != hipSuccess) { fail("hipEventRecord", rc); break; }
    }
    if (cs) {
        rc = hipStreamSynchronize(cs);
        if (ok && rc != hipSuccess) fail("hipStreamSynchronize", rc);
    }
    for (int q = 0; q < NSLOT; ++q) {   // every exit path releases the pinned slots, the events and the copy stream
        if (pin[q]) (void)hipHostFree(pin[q]);
        if (ev[q]) (void)hipEventDestroy(ev[q]);
    }
    if (cs) (void)hipStreamDestroy(cs);
    if (!ok) { err = why; return false; }
#endif
    load_stage_s_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return true;
}

const uint8_t* Engine::staged(const GgufTensor* t) const { return dev_file_ ? dev_file_ + (t->data - file_lo_) : nullptr; }

void Engine::release_staged() {
    if (dev_file_) { hipFree(dev_file_); dev_file_ = nullptr; }
}

// LAYOUT_L9 (decode mat-vec, kernels_v9.h) and LAYOUT_R2C4 (prompt chunks, kernels_pg.h) copies.  The matrices of `parts` are placed back to back in ONE device
// allocation, in order: a launch walks the row pairs of all its jobs of one weight type as a single contiguous unit space
// (attn_q | attn_k | attn_v).  `fuse` (two parts of the same type and shape): ONE fused gate/up matrix, unit u = (row u of
// parts[0], row u of parts[1]), described by parts[0].second; else unit u of a part = its rows (2u, 2u + 1).
bool Engine::upload_r2c4(const std::vector<std::pair<const GgufTensor*, DevMat*>>& parts, bool fuse, std::string& err) {
    struct Plan { const GgufTensor* ta; const GgufTensor* tb; DevMat* m; size_t off, bytes, off9; int type, K, M, nb, n_units; };
    std::vector<Plan> plan;
    size_t total = 0, total9 = 0;
    for (size_t i = 0; i < parts.size(); i += fuse ? 2 : 1) {
        const GgufTensor* ta = parts[i].first;
        const GgufTensor* tb = fuse ? parts[i + 1].first : nullptr;
        if (!ta || (fuse && !tb)) { err = "missing tensor for the R2C4 layout"; return false; }
        if (!is_kquant(ta->type)) { err = "tensor " + ta->name + ": R2C4 layout is for K-quants"; return false; }
        if (tb && (tb->type != ta->type || tb->ne[0] != ta->ne[0] || tb->ne[1] != ta->ne[1])) { err = "gate/up tensors differ in type or shape"; return false; }
        Plan p;
        p.ta = ta; p.tb = tb; p.m = parts[i].second; p.type = ta->type; p.K = (int)ta->ne[0]; p.M = (int)ta->ne[1]; p.nb = p.K / 256;
        if (p.K > 32768) { err = "tensor " + ta->name + ": rows longer than 32768 are not supported yet"; return false; }
        p.n_units = tb ? p.M : (p.M + 1) / 2;
        p.off = total;
        p.bytes = (size_t)p.n_units * ((p.nb + 3) / 4) * r2c4_record_bytes(p.type);
        total += p.bytes;
        p.off9 = total9;
        total9 += (size_t)p.n_units * ((p.nb + 3) / 4) * l9_record_bytes(p.type);
        plan.push_back(p);
    }
    // Two arenas of the same geometry: LAYOUT_R2C4 for the prompt-chunk kernels (kernels_pg.h), LAYOUT_L9 for the decode mat-vec
    // (kernels_v9.h).  HBM is sized for it (the 70B Q5_K_M model: 2 x 48.6 GB of 288); a token step reads the L9 arena only.
    if (dev_file_) {   // tensors already on the device in file layout: repack there
        uint8_t* d = nullptr;
        uint8_t* d9 = nullptr;
        if (!dev_alloc(dev_allocs_, &d, total + 4096, err)) return false;
        if (!dev_alloc(dev_allocs_, &d9, total9 + 4096, err)) return false;
        HIP_OK(hipMemsetAsync(d, 0, total + 4096, stream_));
        HIP_OK(hipMemsetAsync(d9, 0, total9 + 4096, stream_));
        for (const Plan& p : plan) {
            const long long n = (long long)p.n_units * ((p.nb + 3) / 4) * 8;
            const unsigned gx = (unsigned)std::min<long long>((n + 255) / 256, 65535LL * 16);
            const uint8_t* sa = staged(p.ta);
            const uint8_t* sb = p.tb ? staged(p.tb) : nullptr;
            if (p.type == GT_Q4_K) {
                CT_LAUNCH((repack_r2c4_kernel<GT_Q4_K>), dim3(gx), dim3(256), stream_, sa, sb, d + p.off, p.M, p.nb, p.n_units);
                CT_LAUNCH((repack_r2c4_kernel<GT_Q4_K, true>), dim3(gx), dim3(256), stream_, sa, sb, d9 + p.off9, p.M, p.nb, p.n_units);
            } else if (p.type == GT_Q5_K) {
                CT_LAUNCH((repack_r2c4_kernel<GT_Q5_K>), dim3(gx), dim3(256), stream_, sa, sb, d + p.off, p.M, p.nb, p.n_units);
                CT_LAUNCH((repack_r2c4_kernel<GT_Q5_K, true>), dim3(gx), dim3(256), stream_, sa, sb, d9 + p.off9, p.M, p.nb, p.n_units);
            } else {
                CT_LAUNCH((repack_r2c4_kernel<GT_Q6_K>), dim3(gx), dim3(256), stream_, sa, sb, d + p.off, p.M, p.nb, p.n_units);
                CT_LAUNCH((repack_r2c4_kernel<GT_Q6_K, true>), dim3(gx), dim3(256), stream_, sa, sb, d9 + p.off9, p.M, p.nb, p.n_units);
            }
        }
        for (const Plan& p : plan) {
            p.m->r2 = d + p.off;
            p.m->r9 = d9 + p.off9;
            if (p.tb) { p.m->type = p.type; p.m->K = p.K; p.m->M = p.M; p.m->nb = p.nb; p.m->layout = LAYOUT_R2C4; p.m->bytes = p.ta->nbytes + p.tb->nbytes; }
        }
        return true;
    }
    std::vector<uint8_t> st(total, 0), st9(total9, 0);
    for (const Plan& p : plan) {
        const int type = p.type, nb = p.nb, M = p.M, bb = ggml_block_bytes(type), rec = r2c4_record_bytes(type), spu = (nb + 3) / 4;
        const uint8_t* sa = p.ta->data;
        const uint8_t* sb = p.tb ? p.tb->data : nullptr;
        uint8_t* dst = st.data() + p.off;
        uint8_t* dst9 = st9.data() + p.off9;
        const int rec9 = l9_record_bytes(type);
        parallel_rows(p.n_units, [&](int u0, int u1) {
            for (int u = u0; u < u1; ++u)
                for (int s = 0; s < spu; ++s) {
                    uint8_t* rp = dst + ((size_t)u * spu + s) * rec;
                    uint8_t* rp9 = dst9 + ((size_t)u * spu + s) * rec9;
                    for (int rr = 0; rr < 2; ++rr) {
                        const int row = sb ? u : 2 * u + rr;
                        if (row >= M) continue;
                        const uint8_t* src = (sb && rr) ? sb : sa;
                        for (int cc = 0; cc < 4; ++cc) {
                            const int b = 4 * s + cc;
                            if (b < nb) {
                                place_kblock(type, rp, 4 * rr + cc, src + ((size_t)row * nb + b) * bb);
                                place_kblock9(type, rp9, 4 * rr + cc, src + ((size_t)row * nb + b) * bb);
                            }
                        }
                    }
                }
        });
    }
    uint8_t* d = nullptr;
    uint8_t* d9 = nullptr;
    // + 4 KB: the prompt-chunk kernels request up to two block slots past a row's last block (kernels_pg.h), i.e. past the arena's
    // last record for its last unit
    if (!dev_alloc(dev_allocs_, &d, total + 4096, err)) return false;
    if (!dev_alloc(dev_allocs_, &d9, total9 + 4096, err)) return false;
    HIP_OK(hipMemset(d + total, 0, 4096));
    HIP_OK(hipMemset(d9 + total9, 0, 4096));
    HIP_OK(hipMemcpy(d, st.data(), total, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d9, st9.data(), total9, hipMemcpyHostToDevice));
    for (const Plan& p : plan) {
        p.m->r2 = d + p.off;
        p.m->r9 = d9 + p.off9;
        if (p.tb) {   // the fused matrix is a DevMat of its own
            p.m->type = p.type; p.m->K = p.K; p.m->M = p.M; p.m->nb = p.nb; p.m->layout = LAYOUT_R2C4; p.m->bytes = p.ta->nbytes + p.tb->nbytes;
        }
    }
    return true;
}

// LAYOUT_L9 arena of Q8_0 / Q4_0 matrices for the decode mat-vec (kernels_v9.h); same contract as upload_r2c4 (the matrices of
// `parts` back to back in one allocation; `fuse`: one fused gate/up matrix described by parts[0].second).  The LAYOUT_G4 copy the
// prompt-chunk kernels read is made by upload_matrix.
bool Engine::upload_l9b(const std::vector<std::pair<const GgufTensor*, DevMat*>>& parts, bool fuse, std::string& err) {
    struct Plan { const GgufTensor* ta; const GgufTensor* tb; DevMat* m; size_t off; int type, K, M, nb, n_units; };
    std::vector<Plan> plan;
    size_t total = 0;
    for (size_t i = 0; i < parts.size(); i += fuse ? 2 : 1) {
        const GgufTensor* ta = parts[i].first;
        const GgufTensor* tb = fuse ? parts[i + 1].first : nullptr;
        if (!ta || (fuse && !tb)) { err = "missing tensor for the L9 layout"; return false; }
        if (!is_block32(ta->type)) { err = "tensor " + ta->name + ": not a 32-element block type"; return false; }
        if (tb && (tb->type != ta->type || tb->ne[0] != ta->ne[0] || tb->ne[1] != ta->ne[1])) { err = "gate/up tensors differ in type or shape"; return false; }
        Plan p;
        p.ta = ta; p.tb = tb; p.m = parts[i].second; p.type = ta->type; p.K = (int)ta->ne[0]; p.M = (int)ta->ne[1]; p.nb = p.K / 32;
        if (p.K % 32 || p.K > 32768) { err = "tensor " + ta->name + ": rows of " + std::to_string(p.K) + " elements are not supported"; return false; }
        p.n_units = tb ? p.M : (p.M + 1) / 2;
        p.off = total;
        total += (size_t)p.n_units * l9_spu(p.type, p.K) * l9_record_bytes(p.type);
        plan.push_back(p);
    }
    uint8_t* d9 = nullptr;
    if (!dev_alloc(dev_allocs_, &d9, total + 4096, err)) return false;
    if (dev_file_) {
        HIP_OK(hipMemsetAsync(d9, 0, total + 4096, stream_));
        for (const Plan& p : plan) {
            const long long n = (long long)p.n_units * l9_spu(p.type, p.K) * 32;
            const unsigned gx = (unsigned)std::min<long long>((n + 255) / 256, 65535LL * 16);
            CT_LAUNCH(repack_l9b_kernel, dim3(gx), dim3(256), stream_, p.type, staged(p.ta), p.tb ? staged(p.tb) : (const uint8_t*)nullptr, d9 + p.off, p.M, p.nb, p.n_units);
        }
    } else {
        std::vector<uint8_t> st(total, 0);
        for (const Plan& p : plan) {
            const int bb = ggml_block_bytes(p.type), rec = l9_record_bytes(p.type), spu = l9_spu(p.type, p.K);
            const uint8_t* sa = p.ta->data;
            const uint8_t* sb = p.tb ? p.tb->data : nullptr;
            uint8_t* dst = st.data() + p.off;
            const int M = p.M, nb = p.nb, type = p.type;
            parallel_rows(p.n_units, [&](int u0, int u1) {
                for (int u = u0; u < u1; ++u)
                    for (int rr = 0; rr < 2; ++rr) {
                        const int row = sb ? u : 2 * u + rr;
                        if (row >= M) continue;
                        const uint8_t* src = (sb && rr) ? sb : sa;
                        for (int b = 0; b < nb; ++b)
                            place_block32_l9(type, dst + ((size_t)u * spu + (b >> 4)) * rec, rr, b & 15, src + ((size_t)row * nb + b) * bb);
                    }
            });
        }
        HIP_OK(hipMemset(d9 + total, 0, 4096));
        HIP_OK(hipMemcpy(d9, st.data(), total, hipMemcpyHostToDevice));
    }
    for (const Plan& p : plan) {
        p.m->r9 = d9 + p.off;
        if (p.tb) { p.m->type = p.type; p.m->K = p.K; p.m->M = p.M; p.m->nb = p.nb; p.m->layout = LAYOUT_L9; p.m->bytes = p.ta->nbytes + p.tb->nbytes; }
    }
    return true;
}

bool Engine::upload_matrix(const GgufTensor* t, DevMat& m, bool keep_raw, std::string& err) {
    m.type = t->type;
    m.K = (int)t->ne[0];
    m.M = (int)t->ne[1];
    const int be = ggml_block_elems(t->type), bb = ggml_block_bytes(t->type);
    if (!(is_kquant(t->type) || t->type == GT_Q8_0 || t->type == GT_Q4_0)) {
        err = "tensor " + t->name + ": weight type " + std::to_string(t->type) + " has no mat-vec kernel yet";
        return false;
    }
    m.nb = m.K / be;
    m.bytes = t->nbytes;
    const int nb = m.nb, M = m.M;
    if (is_kquant(t->type)) {
        if (m.K > 32768) { err = "tensor " + t->name + ": rows longer than 32768 are not supported yet"; return false; }
        m.layout = LAYOUT_R2C4;   // the records live in an arena several matrices may share: upload_r2c4 places them
        return true;
    }
    {   // Q8_0 / Q4_0
        // the 32-block kernels (kernels_q32.h) take rows in groups of four blocks and at most 12288 elements: real Falcon-7B
        // (n_embd 4544) and the ffn_down rows of Llama-13B/70B Q4_0/Q8_0 files (13824, 28672) are outside that — said here, at
        // load, not by a failing launch later
        if (m.K % 32 || m.K > 32768) {
            err = "tensor " + t->name + ": Q8_0/Q4_0 rows of " + std::to_string(m.K) + " elements are not supported (need a multiple of 32, at most 32768)";
            return false;
        }
        if (m.K % 128) {
            // Rows that are not whole groups of four blocks (real Falcon-7B: n_embd 4544 = 142 blocks): the decode arena (LAYOUT_L9,
            // upload_l9b: a row's last record is padded with zero blocks, the prologue writes zero images with y.d = 0 for them) serves
            // them; the prompt-chunk kernels' LAYOUT_G4 copy does not exist, so such a handle evaluates prompts token by token
            // (alloc_state: pf_ok_ stays false) — the reference's results either way.
            m.layout = LAYOUT_L9;
            return true;
        }
        m.layout = LAYOUT_G4;
        const bool q8 = t->type == GT_Q8_0;
        const int n_tiles = (M + 7) / 8, ng = nb / 4, rec = q8 ? 1088 : 576, dbase = q8 ? 1024 : 512;
        if (dev_file_) {   // the tensor is already on the device in file layout (stage_file): repack there
            const size_t bytes = (size_t)n_tiles * ng * rec;
            uint8_t* d = nullptr;
            if (!dev_alloc(dev_allocs_, &d, bytes + 64, err)) return false;
            HIP_OK(hipMemsetAsync(d, 0, bytes + 64, stream_));
            const long long n = (long long)n_tiles * ng * 32;
            CT_LAUNCH(repack_g4_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 65535LL * 16)), dim3(256), stream_, q8 ? 1 : 0, staged(t), d, M, nb);
            m.p[0] = d;
            return true;
        }
        std::vector<uint8_t> st((size_t)n_tiles * ng * rec, 0);
        const uint8_t* src = t->data;
        parallel_rows(n_tiles, [&](int t0, int t1) {
            for (int tl = t0; tl < t1; ++tl)
                for (int g = 0; g < ng; ++g) {
                    uint8_t* rp = &st[((size_t)tl * ng + g) * rec];
                    for (int r = 0; r < 8; ++r) {
                        const int row = tl * 8 + r;
                        if (row >= M) continue;
                        for (int i = 0; i < 4; ++i) {
                            const uint8_t* blk = src + ((size_t)row * nb + (size_t)g * 4 + i) * bb;
                            memcpy(rp + dbase + r * 8 + i * 2, blk, 2);
                            if (q8) {
                                for (int l = 0; l < 8; ++l) memcpy(rp + (r * 8 + l) * 16 + i * 4, blk + 2 + 4 * l, 4);
                            } else {
                                for (int l = 0; l < 4; ++l) memcpy(rp + (r * 4 + l) * 16 + i * 4, blk + 2 + 4 * l, 4);
                            }
                        }
                    }
                }
        });
        uint8_t* d = nullptr;
        if (!dev_alloc(dev_allocs_, &d, st.size() + 64, err)) return false;
        HIP_OK(hipMemcpy(d, st.data(), st.size(), hipMemcpyHostToDevice));
        m.p[0] = d;
        return true;
    }
}

bool Engine::upload_f32(const GgufTensor* t, float** out, int n, std::string& err) {
    if (!t) { err = "missing f32 tensor"; return false; }
    if (t->type != GT_F32 || t->ne[0] != n) { err = "tensor " + t->name + " must be f32[" + std::to_string(n) + "]"; return false; }
    if (!dev_alloc(dev_allocs_, out, (size_t)n, err)) return false;
    if (dev_file_ && t->data >= file_lo_ && t->data + (size_t)n * 4 <= file_hi_) HIP_OK(hipMemcpyAsync(*out, staged(t), (size_t)n * 4, hipMemcpyDeviceToDevice, stream_));
    else HIP_OK(hipMemcpy(*out, t->data, (size_t)n * 4, hipMemcpyHostToDevice));
    return true;
}

// fp16 lookup tables with the reference's exact contents (ggml.c:4318-4332, built with the host libm like the
// reference does), and the RoPE cos/sin table with the reference's iterative theta (ggml.c:12482-12539).
bool Engine::build_tables(std::string& err) {
    std::vector<uint16_t> e(65536), s(65536), g(65536);
    for (int i = 0; i < 65536; ++i) {
        const float f = f16_bits_to_f32((uint16_t)i);
        e[i] = f32_to_f16_bits(expf(f));
        s[i] = f32_to_f16_bits(f / (1.0f + expf(-f)));
        // GELU table: the reference build contracts `1 + A*x*x` into one fma and nothing else (oracle/mirror.c:init_gelu_table)
        g[i] = f32_to_f16_bits((0.5f * f) * (1.0f + tanhf((0.79788456080286535587989211986876f * f) * fmaf(0.044715f * f, f, 1.0f))));
    }
    if (!dev_alloc(dev_allocs_, &exp_tab_, 65536, err) || !dev_alloc(dev_allocs_, &silu_tab_, 65536, err) ||
        !dev_alloc(dev_allocs_, &gelu_tab_, 65536, err))
        return false;
    HIP_OK(hipMemcpy(exp_tab_, e.data(), 65536 * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(silu_tab_, s.data(), 65536 * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(gelu_tab_, g.data(), 65536 * 2, hipMemcpyHostToDevice));

    const int hd = hp_.head_dim(), half = hd / 2;
    std::vector<float> cs((size_t)n_ctx_ * half * 2);
    const float theta_scale = powf(hp_.rope_freq_base, -2.0f / (float)hp_.n_rot);
    for (int p = 0; p < n_ctx_; ++p) {
        float theta = hp_.rope_freq_scale * (float)p;
        for (int i = 0; i < half; ++i) {
            cs[((size_t)p * half + i) * 2 + 0] = cosf(theta);
            cs[((size_t)p * half + i) * 2 + 1] = sinf(theta);
            theta *= theta_scale;
        }
    }
    if (!dev_alloc(dev_allocs_, &rope_cs_, cs.size(), err)) return false;
    HIP_OK(hipMemcpy(rope_cs_, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
    return true;
}

bool Engine::load(const std::string& path, int context_length, int gpu_layers, std::string& err, int layer_begin,
                  int layer_end, int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        err = "no HIP device visible: this library runs on MI355X only and has no CPU fallback";
        return false;
    }
    if (device < 0 || device >= ndev) { err = "HIP device ordinal out of range"; return false; }
    device_ = device;
    HIP_OK(hipSetDevice(device_));
    (void)gpu_layers;  // every layer lives on the GPU(s); the CPU/GPU split of the reference does not exist here

    GgufFile f;
    if (!f.open(path)) { err = f.error(); return false; }
    if (!f.get_str("general.architecture", hp_.arch)) { err = "general.architecture missing"; return false; }
    if (hp_.arch != "llama" && hp_.arch != "falcon") { err = "architecture '" + hp_.arch + "' is not supported yet (llama, falcon)"; return false; }
    const std::string a = hp_.arch + ".";
    uint32_t u;
    auto need = [&](const char* key, int& out) {
        if (!f.get_u32(a + key, u)) { err = "missing key " + a + key; return false; }
        out = (int)u;
        return true;
    };
    if (!need("context_length", hp_.n_ctx_train) || !need("embedding_length", hp_.n_embd) ||
        !need("attention.head_count", hp_.n_head) || !need("block_count", hp_.n_layer) ||
        !need("feed_forward_length", hp_.n_ff))
        return false;
    hp_.n_head_kv = hp_.n_head;
    if (f.get_u32(a + "attention.head_count_kv", u)) hp_.n_head_kv = (int)u;
    if (hp_.n_embd <= 0 || hp_.n_head <= 0 || hp_.n_head_kv <= 0 || hp_.n_layer <= 0 || hp_.n_ff <= 0 || hp_.n_embd % hp_.n_head != 0 ||
        hp_.n_head % hp_.n_head_kv != 0) {
        err = "inconsistent hyper-parameters (embedding_length / head_count / head_count_kv / block_count / feed_forward_length)";
        return false;
    }
    hp_.n_rot = hp_.n_embd / hp_.n_head;
    if (f.get_u32(a + "rope.dimension_count", u)) hp_.n_rot = (int)u;
    if (hp_.falcon()) {
        if (!f.get_f32(a + "attention.layer_norm_epsilon", hp_.rms_eps)) { err = "missing layer_norm epsilon"; return false; }
    } else if (!f.get_f32(a + "attention.layer_norm_rms_epsilon", hp_.rms_eps)) { err = "missing rms epsilon"; return false; }
    f.get_f32(a + "rope.freq_base", hp_.rope_freq_base);
    float rs = 1.0f;
    if (f.get_f32(a + "rope.scale_linear", rs) && rs != 0.0f) hp_.rope_freq_scale = 1.0f / rs;
    if (hp_.n_rot != hp_.head_dim()) { err = "rope.dimension_count must equal head_dim"; return false; }
    if (hp_.head_dim() % 64 || hp_.head_dim() > 256) { err = "unsupported head_dim (need a multiple of 64)"; return false; }
    if (!vocab_.load(f, err)) return false;
    hp_.n_vocab = vocab_.size();
    // reference default n_ctx = 512 unless context_length is passed (llama.cpp:5281, llama.cc:90-92)
    n_ctx_ = context_length > 0 ? context_length : 512;
    if (n_ctx_ > kMaxCtxFused) { err = "context_length above " + std::to_string(kMaxCtxFused) + " not supported yet"; return false; }

    HIP_OK(hipStreamCreate(&stream_));
    bool r2_auto = true;   // mat() also makes the matrix's own R2C4 copy (false: the caller places several matrices in one arena)
    const GgufTensor* t;
    auto mat = [&](const std::string& name, DevMat& m, int M, int K, bool raw = false) {
        t = f.tensor(name);
        if (!t) { err = "missing tensor " + name; return false; }
        if (t->ne[0] != K || t->ne[1] != M) { err = "bad shape for " + name; return false; }
        if (!upload_matrix(t, m, raw, err)) return false;
        if (is_kquant(t->type) && r2_auto && !upload_r2c4({{t, &m}}, false, err)) return false;
        if (is_block32(t->type) && r2_auto && !upload_l9b({{t, &m}}, false, err)) return false;
        weight_bytes_ += t->nbytes;
        return true;
    };
    const int E = hp_.n_embd, G = hp_.n_embd_gqa(), F = hp_.n_ff, V = hp_.n_vocab;
    l0_ = layer_begin < 0 ? 0 : layer_begin;
    l1_ = layer_end < 0 ? hp_.n_layer : layer_end;
    if (l0_ >= l1_ || l1_ > hp_.n_layer) { err = "bad pipeline stage layer range"; return false; }
    {   // Load pipeline.  The tensors of this handle's layers: all matrices K-quants -> stage the file range on the GPU and repack there.
        std::vector<const GgufTensor*> need;
        size_t kq_bytes = 0, other_bytes = 0;   // 2-D weight bytes that are K-quants (repacked on the GPU) / other types (host path)
        for (const GgufTensor& x : f.tensors()) {
            bool mine = false;
            if (x.name.compare(0, 4, "blk.") == 0) { const int li = atoi(x.name.c_str() + 4); mine = li >= l0_ && li < l1_; }
            else if (x.name.compare(0, 10, "token_embd") == 0) mine = l0_ == 0;
            else if (x.name.compare(0, 6, "output") == 0) mine = l1_ == hp_.n_layer;
            if (!mine) continue;
            need.push_back(&x);
            if (x.n_dims >= 2 && x.name.compare(0, 10, "token_embd") != 0) (is_kquant(x.type) ? kq_bytes : other_bytes) += x.nbytes;
        }
        // every weight type is repacked on the GPU (K-quants: repack_r2c4_kernel, Q8_0 / Q4_0: repack_g4_kernel); CT_AMD_GPU_REPACK=0: host
        // repack everywhere (A/B)
        (void)kq_bytes; (void)other_bytes;
        if (env_int("CT_AMD_GPU_REPACK", 1) != 0 && !stage_file(f, need, err)) return false;
    }
    t = f.tensor("token_embd.weight");
    if (!t || t->ne[0] != E || t->ne[1] != V) { err = "bad token_embd.weight"; return false; }
    if (l0_ == 0) {   // token_embd is only used by row lookup: keep the file layout, no planes
        tok_embd_.type = t->type; tok_embd_.K = E; tok_embd_.M = V;
        uint8_t* d = nullptr;
        if (!dev_alloc(dev_allocs_, &d, t->nbytes, err)) return false;
        if (dev_file_) HIP_OK(hipMemcpyAsync(d, staged(t), t->nbytes, hipMemcpyDeviceToDevice, stream_));
        else HIP_OK(hipMemcpy(d, t->data, t->nbytes, hipMemcpyHostToDevice));
        tok_embd_.raw = d;
    }
    layers_.resize(hp_.n_layer);
    if (hp_.falcon()) {
        for (int i = l0_; i < l1_; ++i) {
            const std::string p = "blk." + std::to_string(i) + ".";
            Layer& L = layers_[i];
            if (!upload_f32(f.tensor(p + "attn_norm.weight"), &L.attn_norm, E, err)) return false;
            if (!upload_f32(f.tensor(p + "attn_norm.bias"), &L.attn_norm_b, E, err)) return false;
            if (f.tensor(p + "attn_norm_2.weight")) {   // Falcon-40B style: separate norm for the attention input
                if (!upload_f32(f.tensor(p + "attn_norm_2.weight"), &L.attn_norm2, E, err)) return false;
                if (!upload_f32(f.tensor(p + "attn_norm_2.bias"), &L.attn_norm2_b, E, err)) return false;
            }
            if (!mat(p + "attn_qkv.weight", L.wqkv, E + 2 * G, E) || !mat(p + "attn_output.weight", L.wo, E, E) ||
                !mat(p + "ffn_up.weight", L.w_up, F, E) || !mat(p + "ffn_down.weight", L.w_down, E, F))
                return false;
        }
        if (l1_ == hp_.n_layer) {
            if (!upload_f32(f.tensor("output_norm.weight"), &output_norm_, E, err)) return false;
            if (!upload_f32(f.tensor("output_norm.bias"), &output_norm_b_, E, err)) return false;
            if (!mat("output.weight", output_, V, E)) return false;
        }
        if (!dev_alloc(dev_allocs_, &qkv_tmp_, (size_t)(E + 2 * G), err) || !dev_alloc(dev_allocs_, &attn_proj_, (size_t)E, err))
            return false;
    } else {
    for (int i = l0_; i < l1_; ++i) {
        const std::string p = "blk." + std::to_string(i) + ".";
        Layer& L = layers_[i];
        if (!upload_f32(f.tensor(p + "attn_norm.weight"), &L.attn_norm, E, err)) return false;
        if (!upload_f32(f.tensor(p + "ffn_norm.weight"), &L.ffn_norm, E, err)) return false;
        r2_auto = false;   // q | k | v share one arena, gate/up are fused
        if (!mat(p + "attn_q.weight", L.wq, E, E) || !mat(p + "attn_k.weight", L.wk, G, E) ||
            !mat(p + "attn_v.weight", L.wv, G, E))
            return false;
        if (!mat(p + "ffn_gate.weight", L.w_gate, F, E) || !mat(p + "ffn_up.weight", L.w_up, F, E)) return false;
        r2_auto = true;
        if (!mat(p + "attn_output.weight", L.wo, E, E) || !mat(p + "ffn_down.weight", L.w_down, E, F)) return false;
        if (L.w_gate.type != L.w_up.type) { err = "ffn_gate/ffn_up type mismatch in layer " + std::to_string(i); return false; }
        if (is_kquant(L.wq.type) && is_kquant(L.wk.type) && is_kquant(L.wv.type)) {
            if (!upload_r2c4({{f.tensor(p + "attn_q.weight"), &L.wq}, {f.tensor(p + "attn_k.weight"), &L.wk}, {f.tensor(p + "attn_v.weight"), &L.wv}}, false, err))
                return false;
        } else if (is_block32(L.wq.type) && L.wk.type == L.wq.type && L.wv.type == L.wq.type) {
            if (!upload_l9b({{f.tensor(p + "attn_q.weight"), &L.wq}, {f.tensor(p + "attn_k.weight"), &L.wk}, {f.tensor(p + "attn_v.weight"), &L.wv}}, false, err))
                return false;
        } else {   // mixed families or 32-block types: every matrix its own arena (the site then takes one launch per group, launch_matvec)
            const std::pair<const char*, DevMat*> qkv[3] = {{"attn_q.weight", &L.wq}, {"attn_k.weight", &L.wk}, {"attn_v.weight", &L.wv}};
            for (const auto& it : qkv) {
                if (is_kquant(it.second->type) && !upload_r2c4({{f.tensor(p + it.first), it.second}}, false, err)) return false;
                if (is_block32(it.second->type) && !upload_l9b({{f.tensor(p + it.first), it.second}}, false, err)) return false;
            }
        }
        if (is_kquant(L.w_gate.type) &&
            !upload_r2c4({{f.tensor(p + "ffn_gate.weight"), &L.w_gu}, {f.tensor(p + "ffn_up.weight"), &L.w_gu}}, true, err))
            return false;
        if (is_block32(L.w_gate.type) &&
            !upload_l9b({{f.tensor(p + "ffn_gate.weight"), &L.w_gu}, {f.tensor(p + "ffn_up.weight"), &L.w_gu}}, true, err))
            return false;
    }
    if (l1_ == hp_.n_layer) {
        if (!upload_f32(f.tensor("output_norm.weight"), &output_norm_, E, err)) return false;
        if (!mat("output.weight", output_, V, E)) return false;
    }
    }
    if (l0_ > 0 || l1_ < hp_.n_layer)
        if (!dev_alloc(dev_allocs_, &xio_, (size_t)n_ctx_ * E, err)) return false;

    if (!alloc_state(err)) return false;
    HIP_OK(hipDeviceSynchronize());
    release_staged();
    if (!warm_up(err)) return false;
    return true;
}

// First use of a kernel pays for loading the code object, the dynamic-LDS opt-ins and (decode) the graph capture: ~7 ms on a
// first 128-token prompt.  A whole-model handle pays it here, at load time, with a two-token chunk of token 0 and the capture
// of the token-step graphs.  Nothing of it is observable through the ABI: the logits stay "not evaluated" (size 0), the
// positions it touched in the KV cache are rewritten by the first real tokens that use them.
bool Engine::warm_up(std::string& err) {
#ifndef CT_EMU
    if (l0_ != 0 || l1_ != hp_.n_layer || dump_dir_ || env_int("CT_AMD_WARMUP", 1) == 0 || n_ctx_ < 4) return true;
    h_scalars_[0] = 0; h_scalars_[1] = 0; h_scalars_[2] = 2; h_scalars_[3] = 0; h_scalars_[4] = 0; h_scalars_[5] = 0;
    HIP_OK(hipMemcpyAsync(d_state_, &h_scalars_[0], 6 * 4, hipMemcpyHostToDevice, stream_));
    if (pf_ok_) { if (!chunk_step(0, 2, true, err)) return false; }
    else { if (!token_step(true, err)) return false; }
    if (use_graph_ && !ensure_graphs(err)) return false;
    HIP_OK(hipStreamSynchronize(stream_));
    HIP_OK(hipGetLastError());
#endif
    (void)err;
    return true;
}

bool Engine::alloc_state(std::string& err) {
    const int E = hp_.n_embd, G = hp_.n_embd_gqa(), F = hp_.n_ff, V = hp_.n_vocab;
    v_stride_ = (n_ctx_ + 31) / 32 * 32;  // V rows (one per channel) start 16-byte aligned
    const size_t k_elems = (size_t)(l1_ - l0_) * n_ctx_ * G, v_elems = (size_t)(l1_ - l0_) * v_stride_ * G;
    if (!dev_alloc(dev_allocs_, &kcache_, k_elems, err) || !dev_alloc(dev_allocs_, &vcache_, v_elems + 64, err)) return false;
    HIP_OK(hipMemset(kcache_, 0, k_elems * 2));
    HIP_OK(hipMemset(vcache_, 0, v_elems * 2));
    if (!dev_alloc(dev_allocs_, &x_, (size_t)E, err) || !dev_alloc(dev_allocs_, &attn_out_, (size_t)E, err) ||
        !dev_alloc(dev_allocs_, &h_, (size_t)F, err) || !dev_alloc(dev_allocs_, &q_f16_, (size_t)E, err) ||
        !dev_alloc(dev_allocs_, &scores_, (size_t)hp_.n_head * n_ctx_, err) ||
        !dev_alloc(dev_allocs_, &d_logits_, (size_t)V + E, err) ||   // [logits | final-norm embedding]: one D2H copy per eval
        !dev_alloc(dev_allocs_, &trace_buf_, 512, err) || !dev_alloc(dev_allocs_, &d_argmax_, 4, err) || !dev_alloc(dev_allocs_, &d_state_, (size_t)n_ctx_ + 4, err))   // [cursor(4) | tokens]: one H2D copy
        return false;
    d_emb_ = d_logits_ + V;
    d_tokens_ = d_state_ + 4;
    // prompt chunks (kernels_pg.h: K-quants; kernels_pf.h: Q8_0 / Q4_0): n_embd <= 12288, n_ff <= 32768
    pf_ok_ = E <= 12288 && F <= 32768 && env_int("CT_AMD_PF", 1) != 0;
    bool kq_model = false, mixed_model = false;
    {   // every layer matrix a K-quant (llama, falcon), or every one Q8_0 / Q4_0 of one type with K <= 32768
        int n_kq = 0, n_q32 = 0, n_all = 0, ty32 = -1, n_any32 = 0;
        for (int i = l0_; i < l1_; ++i) {
            const Layer& L = layers_[i];
            const std::initializer_list<const DevMat*> llama_mats = {&L.wq, &L.wk, &L.wv, &L.wo, &L.w_gate, &L.w_up, &L.w_down};
            const std::initializer_list<const DevMat*> fused_mats = {&L.wqkv, &L.wo, &L.w_up, &L.w_down};   // falcon, gpt2
            for (const DevMat* m : (hp_.falcon() || hp_.legacy()) ? fused_mats : llama_mats) {
                ++n_all;
                if (m->layout == LAYOUT_R2C4 && is_kquant(m->type)) ++n_kq;
                if (m->layout == LAYOUT_G4 && (ty32 < 0 || ty32 == m->type) && m->K <= 32768) { ++n_q32; ty32 = m->type; }
                if (m->layout == LAYOUT_G4 && m->K <= 32768) ++n_any32;
            }
        }
        // a llama file that mixes the families (a Q8_0 tensor beside K-quants): both kinds of activation images, one pass per family
        // at a site (pf_matvec)
        mixed_model = !hp_.falcon() && !hp_.legacy() && n_kq > 0 && n_any32 > 0 && n_kq + n_any32 == n_all;
        pf_ok_ = pf_ok_ && ((n_kq == n_all && !hp_.gpt2()) || n_q32 == n_all || mixed_model);
        kq_model = n_kq == n_all;
    }
    if (pf_ok_) {
        pf_min_ = std::max(2, env_int("CT_AMD_PF_MIN", 2));
        pf_chunk_ = std::max(pf_min_, std::min(kPfChunk, env_int("CT_AMD_PF_CHUNK", kPfChunk)));
        if (!dev_alloc(dev_allocs_, &xb_, (size_t)kPfChunk * E, err) || !dev_alloc(dev_allocs_, &attn_out_b_, (size_t)kPfChunk * E, err) ||
            !dev_alloc(dev_allocs_, &hb_, (size_t)kPfChunk * F, err) || !dev_alloc(dev_allocs_, &q_f16_b_, (size_t)kPfChunk * E, err))
            return false;
        pg_force_tg_ = env_int("CT_AMD_PG_TG", 0);
        if (!kq_model) {   // Q8_0 activation images of the Q8_0 / Q4_0 chunk kernel
            if (!dev_alloc(dev_allocs_, &acts_, (size_t)kPfChunk * pf_act_words_q32(std::max(E, F)), err)) return false;
        }
        if (kq_model || mixed_model) {   // 128 tokens of stage images per block and layout (kernels_pg.h PgStage)
            acts_h_half_ = (size_t)(std::max(E, F) / 256) * std::max((kPfChunk / 16) * PgStage<16>::BYTES, (kPfChunk / 32) * PgStage<32>::BYTES) + 4096;
            if (!dev_alloc(dev_allocs_, &acts_h_, 2 * acts_h_half_, err)) return false;
            HIP_OK(hipMemset(acts_h_, 0, 2 * acts_h_half_));   // token slots past the chunk's end are read (and their results dropped)
        }
        if (hp_.falcon() && (!dev_alloc(dev_allocs_, &qkv_tmp_b_, (size_t)kPfChunk * (E + 2 * G), err) ||
                             !dev_alloc(dev_allocs_, &attn_proj_b_, (size_t)kPfChunk * E, err)))
            return false;
        if (hp_.legacy() && !dev_alloc(dev_allocs_, &qkv_tmp_b_, (size_t)kPfChunk * 3 * E, err)) return false;
    }
    HIP_OK(hipHostMalloc(&h_logits_, ((size_t)V + E) * 4));
    h_emb_ = h_logits_ + V;
    HIP_OK(hipHostMalloc(&h_scalars_, ((size_t)n_ctx_ + 16) * 4));
    use_graph_ = env_int("CT_AMD_GRAPH", 1) != 0;
    dump_dir_ = getenv("CT_AMD_DUMP");
    if (dump_dir_ && !*dump_dir_) dump_dir_ = nullptr;
    if (dump_dir_) use_graph_ = false;
    memset(h_logits_, 0, (size_t)V * 4);
    memset(h_emb_, 0, (size_t)E * 4);
    if (!build_tables(err)) return false;
    return true;
}

// GPT-2 from the legacy GGML container (reference gpt2_model_load, models/llms/gpt2.cc:61-381).  `starcoder`: the reference's
// starcoder / gptbigcode loader (models/llms/starcoder.cc:62-421) reads the same container, tensor names and shapes (its K/V stay
// expanded to n_head heads, :162-164) and builds the same graph (:424-763); it differs in registering the StarChat markers.
bool Engine::load_gpt2(const std::string& path, std::string& err, int device, bool starcoder) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        err = "no HIP device visible: this library runs on MI355X only and has no CPU fallback";
        return false;
    }
    if (device < 0 || device >= ndev) { err = "HIP device ordinal out of range"; return false; }
    device_ = device;
    HIP_OK(hipSetDevice(device_));
    LegacyGgmlFile f;
    if (!f.open(path)) { err = f.error(); return false; }
    hp_.arch = "gpt2";
    hp_.n_vocab = f.hparams[0];
    n_ctx_ = f.hparams[1];          // the reference sizes memory_k / memory_v and wpe from the file, not from the config
    hp_.n_ctx_train = n_ctx_;
    hp_.n_embd = f.hparams[2];
    hp_.n_head = hp_.n_head_kv = f.hparams[3];
    hp_.n_layer = f.hparams[4];
    hp_.n_ff = 4 * hp_.n_embd;
    hp_.n_rot = hp_.head_dim();
    hp_.rms_eps = 1e-5f;            // ggml_norm(ctx, a) wrapper: models/common.h:211-213
    if (n_ctx_ > kMaxCtx) { err = "context length above " + std::to_string(kMaxCtx) + " not supported yet"; return false; }
    if (hp_.n_embd % 128) { err = "gpt2: n_embd must be a multiple of 128 for the 32-block mat-vec kernels"; return false; }
    vocab_.load_legacy(f.vocab);
    if (starcoder) vocab_.mark_starcoder_specials();
    l0_ = 0;
    l1_ = hp_.n_layer;
    HIP_OK(hipStreamCreate(&stream_));
    const int E = hp_.n_embd, F = hp_.n_ff, V = hp_.n_vocab;
    auto mat = [&](const std::string& name, DevMat& m, int M, int K) {
        const GgufTensor* t = f.tensor(name);
        if (!t) { err = "missing tensor " + name; return false; }
        if (t->ne[0] != K || t->ne[1] != M) { err = "bad shape for " + name; return false; }
        if (t->type != GT_Q4_0 && t->type != GT_Q8_0) { err = name + ": only Q4_0 / Q8_0 legacy weights are supported"; return false; }
        if (!upload_matrix(t, m, false, err)) return false;
        if (!upload_l9b({{t, &m}}, false, err)) return false;
        weight_bytes_ += t->nbytes;
        return true;
    };
    auto vec = [&](const std::string& name, float** out, int n) {
        const GgufTensor* t = f.tensor(name);
        if (!t || t->type != GT_F32 || t->ne[0] != n) { err = "bad or missing f32 tensor " + name; return false; }
        return upload_f32(t, out, n, err);
    };
    const GgufTensor* wte = f.tensor("model/wte");
    if (!wte || wte->ne[0] != E || wte->ne[1] != V) { err = "bad model/wte"; return false; }
    {   // row lookup copy in file layout
        tok_embd_.type = wte->type; tok_embd_.K = E; tok_embd_.M = V;
        uint8_t* d = nullptr;
        if (!dev_alloc(dev_allocs_, &d, wte->nbytes, err)) return false;
        HIP_OK(hipMemcpy(d, wte->data, wte->nbytes, hipMemcpyHostToDevice));
        tok_embd_.raw = d;
    }
    {   // lm_head = its own tensor if the file has one, else the tied wte (gpt2.cc:357-368)
        const bool own = f.tensor("model/lm_head") != nullptr;
        if (!mat(own ? "model/lm_head" : "model/wte", output_, V, E)) return false;
    }
    {
        const GgufTensor* t = f.tensor("model/wpe");
        if (!t || t->type != GT_F32 || t->ne[0] != E || t->ne[1] != n_ctx_) { err = "bad model/wpe"; return false; }
        if (!dev_alloc(dev_allocs_, &wpe_, (size_t)E * n_ctx_, err)) return false;
        HIP_OK(hipMemcpy(wpe_, t->data, (size_t)E * n_ctx_ * 4, hipMemcpyHostToDevice));
    }
    if (!vec("model/ln_f/g", &output_norm_, E) || !vec("model/ln_f/b", &output_norm_b_, E)) return false;
    layers_.resize(hp_.n_layer);
    for (int i = 0; i < hp_.n_layer; ++i) {
        const std::string p = "model/h" + std::to_string(i) + "/";
        Layer& L = layers_[i];
        if (!vec(p + "ln_1/g", &L.attn_norm, E) || !vec(p + "ln_1/b", &L.attn_norm_b, E) ||
            !vec(p + "ln_2/g", &L.ffn_norm, E) || !vec(p + "ln_2/b", &L.ffn_norm_b, E) ||
            !mat(p + "attn/c_attn/w", L.wqkv, 3 * E, E) || !vec(p + "attn/c_attn/b", &L.b_qkv, 3 * E) ||
            !mat(p + "attn/c_proj/w", L.wo, E, E) || !vec(p + "attn/c_proj/b", &L.b_wo, E) ||
            !mat(p + "mlp/c_fc/w", L.w_up, F, E) || !vec(p + "mlp/c_fc/b", &L.b_up, F) ||
            !mat(p + "mlp/c_proj/w", L.w_down, E, F) || !vec(p + "mlp/c_proj/b", &L.b_down, E))
            return false;
    }
    if (!dev_alloc(dev_allocs_, &qkv_tmp_, (size_t)3 * E, err) ||
        !dev_alloc(dev_allocs_, &kmem_, (size_t)hp_.n_layer * n_ctx_ * E, err) ||
        !dev_alloc(dev_allocs_, &vmem_, (size_t)hp_.n_layer * n_ctx_ * E, err))
        return false;
    HIP_OK(hipMemset(kmem_, 0, (size_t)hp_.n_layer * n_ctx_ * E * 4));
    HIP_OK(hipMemset(vmem_, 0, (size_t)hp_.n_layer * n_ctx_ * E * 4));
    if (!alloc_state(err)) return false;
    HIP_OK(hipDeviceSynchronize());
    return true;
}


// MPT from the legacy GGML container (reference mpt_model_load, models/llms/mpt.cc:50-363): quantized wte (row lookup AND tied
// output head), per layer two bias-free LayerNorms, fused Wqkv, out_proj, up_proj, down_proj; fp16 K / V memory.
bool Engine::load_mpt(const std::string& path, int context_length, std::string& err, int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        err = "no HIP device visible: this library runs on MI355X only and has no CPU fallback";
        return false;
    }
    if (device < 0 || device >= ndev) { err = "HIP device ordinal out of range"; return false; }
    device_ = device;
    HIP_OK(hipSetDevice(device_));
    LegacyGgmlFile f;
    if (!f.open(path, true)) { err = f.error(); return false; }
    hp_.arch = "mpt";
    hp_.n_vocab = f.hparams[0];
    hp_.n_ctx_train = f.hparams[1];
    n_ctx_ = std::min(f.hparams[1], context_length > 0 ? context_length : 2048);   // mpt.cc:15, :80, :605-607
    hp_.n_embd = f.hparams[2];
    hp_.n_head = hp_.n_head_kv = f.hparams[3];
    hp_.n_layer = f.hparams[4];
    if (hp_.n_embd <= 0 || hp_.n_head <= 0 || hp_.n_layer <= 0 || n_ctx_ <= 0 || hp_.n_embd % hp_.n_head) { err = "mpt: bad hyper-parameters"; return false; }
    hp_.n_ff = 4 * hp_.n_embd;
    hp_.n_rot = hp_.head_dim();
    hp_.rms_eps = 1e-5f;            // ggml_norm(ctx, a) wrapper: models/common.h:211-213
    clip_qkv_ = f.clip_qkv;
    if (n_ctx_ > kMaxCtxFused) { err = "context length above " + std::to_string(kMaxCtxFused) + " not supported yet"; return false; }
    if (hp_.n_embd % 128) { err = "mpt: d_model must be a multiple of 128 for the 32-block mat-vec kernels"; return false; }
    if (hp_.head_dim() != 64 && hp_.head_dim() != 112 && hp_.head_dim() != 128) { err = "mpt: head sizes other than 64 / 112 / 128 are not supported"; return false; }
    vocab_.load_legacy(f.vocab);
    l0_ = 0;
    l1_ = hp_.n_layer;
    HIP_OK(hipStreamCreate(&stream_));
    const int E = hp_.n_embd, F = hp_.n_ff, V = hp_.n_vocab;
    auto mat = [&](const std::string& name, DevMat& m, int M, int K) {
        const GgufTensor* t = f.tensor(name);
        if (!t) { err = "missing tensor " + name; return false; }
        if (t->ne[0] != K || t->ne[1] != M) { err = "bad shape for " + name; return false; }
        if (t->type != GT_Q4_0 && t->type != GT_Q8_0) { err = name + ": only Q4_0 / Q8_0 legacy weights are supported"; return false; }
        if (!upload_matrix(t, m, false, err)) return false;
        if (!upload_l9b({{t, &m}}, false, err)) return false;
        weight_bytes_ += t->nbytes;
        return true;
    };
    auto vec = [&](const std::string& name, float** out, int n) {
        const GgufTensor* t = f.tensor(name);
        if (!t || t->type != GT_F32 || t->ne[0] != n) { err = "bad or missing f32 tensor " + name; return false; }
        return upload_f32(t, out, n, err);
    };
    const GgufTensor* wte = f.tensor("transformer.wte.weight");
    if (!wte || wte->ne[0] != E || wte->ne[1] != V) { err = "bad transformer.wte.weight"; return false; }
    {   // row lookup copy in file layout
        tok_embd_.type = wte->type; tok_embd_.K = E; tok_embd_.M = V;
        uint8_t* d = nullptr;
        if (!dev_alloc(dev_allocs_, &d, wte->nbytes, err)) return false;
        HIP_OK(hipMemcpy(d, wte->data, wte->nbytes, hipMemcpyHostToDevice));
        tok_embd_.raw = d;
    }
    if (!mat("transformer.wte.weight", output_, V, E)) return false;   // the output head is the embedding matrix (mpt.cc:561)
    if (!vec("transformer.norm_f.weight", &output_norm_, E)) return false;
    if (!dev_alloc(dev_allocs_, &zero_bias_, (size_t)E, err)) return false;
    HIP_OK(hipMemset(zero_bias_, 0, (size_t)E * 4));
    output_norm_b_ = zero_bias_;
    layers_.resize(hp_.n_layer);
    for (int i = 0; i < hp_.n_layer; ++i) {
        const std::string p = "transformer.blocks." + std::to_string(i) + ".";
        Layer& L = layers_[i];
        if (!vec(p + "norm_1.weight", &L.attn_norm, E) || !vec(p + "norm_2.weight", &L.ffn_norm, E) ||
            !mat(p + "attn.Wqkv.weight", L.wqkv, 3 * E, E) || !mat(p + "attn.out_proj.weight", L.wo, E, E) ||
            !mat(p + "ffn.up_proj.weight", L.w_up, F, E) || !mat(p + "ffn.down_proj.weight", L.w_down, E, F))
            return false;
        L.attn_norm_b = L.ffn_norm_b = zero_bias_;
    }
    {   // ALiBi slopes, computed as ggml_compute_forward_alibi_f32 computes them (ggml.c:12228-12247), with this host's powf
        std::vector<float> m((size_t)hp_.n_head);
        const int n2 = 1 << (int)floor(log2(hp_.n_head));
        const float m0 = powf(2.0f, -(f.alibi_bias_max) / n2), m1 = powf(2.0f, -(f.alibi_bias_max / 2.0f) / n2);
        for (int k = 0; k < hp_.n_head; ++k) m[(size_t)k] = k < n2 ? powf(m0, k + 1) : powf(m1, 2 * (k - n2) + 1);
        if (!dev_alloc(dev_allocs_, &alibi_, m.size(), err)) return false;
        HIP_OK(hipMemcpy(alibi_, m.data(), m.size() * 4, hipMemcpyHostToDevice));
    }
    if (!dev_alloc(dev_allocs_, &qkv_tmp_, (size_t)3 * E, err)) return false;
    if (!alloc_state(err)) return false;
    HIP_OK(hipDeviceSynchronize());
    return true;
}

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

static bool launch_matvec_kq(MatvecArgs& a, hipStream_t s, std::string& err) {
    ++g_kq_launches;
    for (int j = 0; j < a.njobs; ++j) {
        if (!a.job[j].w.r9) { err = "mat-vec: K-quant matrix without a LAYOUT_L9 arena"; return false; }
    }
    const int ta = a.job[0].w.type;
    int tb = 0, item0 = 0, na = 0;
    double bytes_a = 0.0, bytes_b = 0.0;
    const int spu = l9_spu(ta, a.K);
    for (int j = 0; j < a.njobs; ++j) {
        const int tj = a.job[j].w.type;
        const int units = a.gateup ? a.job[j].w.M : (a.job[j].w.M + 1) / 2;
        a.job[j].pair0 = item0;
        item0 += units;
        const double bytes = (double)units * spu * l9_record_bytes(tj);
        if (tj == ta && tb == 0) { na += units; bytes_a += bytes; }
        else if ((tb == 0 && tj == GT_Q6_K) || tj == tb) { tb = tj; bytes_b += bytes; }
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
    if (tb != 0) {
        const int nwb = std::max(1, std::min(15, (int)lround(16.0 * bytes_b / (bytes_a + bytes_b))));
        a.nwA = 16 - nwb;
    }
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
    const dim3 grid((unsigned)gx), block(1024);
    {
        if (a.emb_out && (tb != 0 || a.K > 16384)) { err = "emb_out on a mixed-type or wide launch"; return false; }
        if (tb != 0 && a.K > 16384) { err = "mixed-type launch with K > 16384"; return false; }
#define V9L(MK, TAV, TBV, LNV, EMBV) do { \
        auto kfn = matvec_v9_kernel<MK, TAV, TBV, LNV, EMBV>; \
        constexpr size_t smem = sizeof(SmemV9<MK>); \
        CT_OPTIN_ONCE(kfn, smem); \
        CT_LAUNCH_DYN(kfn, grid, block, smem, s, a); } while (0)
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
static bool launch_matvec(MatvecArgs& a, hipStream_t s, std::string& err) {
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

bool Engine::run_matvec(MatvecArgs& a, std::string& err) { return launch_matvec(a, stream_, err); }

// One fused attention launch for the current token over this layer's fp16 KV cache (kernels_exact.h).
void Engine::launch_attention(uint16_t* kc, uint16_t* vc, int nt) {
    const int hd = hp_.head_dim();
    AttnArgsX ax = AttnArgsX();
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
    if (trace_site_ && !strcmp(trace_site_, "attn")) ax.trace = trace_buf_;
    const dim3 ag((unsigned)hp_.n_head, (unsigned)(hd / 64), (unsigned)std::max(1, nt));   // nt > 0: the tokens of a prompt chunk
    const size_t smem = (size_t)((n_ctx_ + 63) & ~63) * 4;   // the probability row
#define ATTN(NTV, HDV, ALLV, GRID) do { \
        auto kfn = attn_fused_exact_kernel<NTV, HDV, ALLV>; \
        CT_OPTIN_ONCE(kfn, (size_t)kMaxCtxFused * 4); \
        CT_LAUNCH_DYN(kfn, GRID, dim3(NTV), smem, stream_, ax); } while (0)
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
#define ATTNL(HDV, NTV) do { \
                auto kfn = attn_chunk_long_kernel<HDV, NTV>; \
                CT_OPTIN_ONCE(kfn, cap); \
                CT_LAUNCH_DYN(kfn, gl, dim3(NTV * 64), sm, stream_, ax, nt, row); } while (0)
            if (hd == 128) { if (ntok == 16) ATTNL(128, 16); else ATTNL(128, 8); }
            else { if (ntok == 16) ATTNL(64, 16); else ATTNL(64, 8); }
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
        CT_LAUNCH_DYN(kfn, g9, b9, smem, stream_, ax, ng); } while (0)
        // ring depth of the K / V requests; seven score waves with two K-row slots each measured best at contexts <= 1024 (3 / 4 / 5 score
        // waves, four slots: 0-4 % slower per token on the 7B, profiles/r03_attn9_score_waves_ab.txt)
        static const int deep_min = env_int("CT_AMD_ATTN_DEEP_CTX", 1024);   // measurement switch: contexts above this take the deep-ring form
        const bool deep = n_ctx_ > deep_min;
        if (deep) {   // 512 threads: 8 - pv_waves score waves (kernels_attn9.h)
            const dim3 b9d(512);
#define ATTN9D(HDV, NWVV) do { \
            auto kfn = attn_decode9_kernel<HDV, 4, 16, NWVV, 512>; \
            CT_OPTIN_ONCE(kfn, (size_t)kMaxCtxFused * 4); \
            CT_LAUNCH_DYN(kfn, g9, b9d, smem, stream_, ax, ng); } while (0)
            if (hd == 128) { if (pv_waves == 1) ATTN9D(128, 7); else if (pv_waves == 2) ATTN9D(128, 6); else ATTN9D(128, 4); }
            else { if (pv_waves == 1) ATTN9D(64, 7); else if (pv_waves == 2) ATTN9D(64, 6); else ATTN9D(64, 4); }
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
    // Tokens per workgroup: 32 amortise the weight unpack over two matrix products — unless the site then has fewer workgroups than
    // the chip has CUs (Wo, ffn_down of a 7B: 32 x 4): 16-token groups double the workgroups and every CU gets one.
    int tg = nt > 16 ? 32 : 16;
    if (tg == 32) {
        int items = 0;
        for (int j = 0; j < m.njobs; ++j) items += m.gateup ? (m.job[j].w.M + 7) / 8 : (m.job[j].w.M + 15) / 16;
        if (((items + kPgWaves - 1) / kPgWaves) * ((nt + 31) / 32) < chip_cus() * (8 / kPgWaves)) tg = 16;
    }
    if (pg_force_tg_ == 16 || pg_force_tg_ == 32) tg = pg_force_tg_;
    const int groups = (nt + tg - 1) / tg, nb = m.K / 256;
    if ((size_t)groups * nb * pg_stage_bytes(tg) > acts_h_half_) { err = "stage images exceed their buffer"; return false; }
    bool has45 = false, has6 = false;
    for (int j = 0; j < m.njobs; ++j) (m.job[j].w.type == GT_Q6_K ? has6 : has45) = true;
    uint8_t* img45 = has45 ? acts_h_ : nullptr;
    uint8_t* img6 = has6 ? acts_h_ + acts_h_half_ : nullptr;
    const dim3 qg((unsigned)nt), qb(1024);
    if (m.pro == PRO_LAYERNORM) {   // falcon: n_embd-long inputs only
        if (m.K <= 4096) CT_LAUNCH((pg_quantize_kernel<4096, true>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, img45, img6, tg, m.norm_b);
        else CT_LAUNCH((pg_quantize_kernel<12288, true>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, img45, img6, tg, m.norm_b);
    } else if (m.K <= 4096) CT_LAUNCH((pg_quantize_kernel<4096, false>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, img45, img6, tg, (const float*)nullptr);
    else if (m.K <= 12288) CT_LAUNCH((pg_quantize_kernel<12288, false>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, img45, img6, tg, (const float*)nullptr);
    else CT_LAUNCH((pg_quantize_kernel<32768, false>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, img45, img6, tg, (const float*)nullptr);
    constexpr int NW = kPgWaves;
    for (const int ty : {GT_Q4_K, GT_Q5_K, GT_Q6_K}) {
        PgArgs a;
        a.m = m;
        a.acts = ty == GT_Q6_K ? img6 : img45;
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
            CT_LAUNCH_DYN(kfn, grid, block, smem, stream_, a); } while (0)
#define PG_T(TYV) do { if (tg == 16) { if (m.gateup) PG(TYV, 16, true); else PG(TYV, 16, false); } \
                       else { if (m.gateup) PG(TYV, 32, true); else PG(TYV, 32, false); } } while (0)
        static const char* pg_trace = getenv("CT_AMD_PG_TRACE");   // measurement only: "gate_up" / "qkv" / "wo" / "down": in-kernel stamps of that site
        if (pg_trace && ty == GT_Q4_K && tg == 32 && pg_trace_site_ && !strcmp(pg_trace, pg_trace_site_) && nt > 32) {
            a.m.dbg |= 32; a.m.dbg_sink = (float*)trace_buf_;
            if (m.gateup) { auto kfn = matmul_pg_kernel<GT_Q4_K, 32, NW, true, true>; CT_OPTIN_ONCE(kfn, 3 * (size_t)PgStage<32>::BYTES + 1024); CT_LAUNCH_DYN(kfn, grid, block, smem + 1024, stream_, a); }
            else { auto kfn = matmul_pg_kernel<GT_Q4_K, 32, NW, false, true>; CT_OPTIN_ONCE(kfn, 3 * (size_t)PgStage<32>::BYTES + 1024); CT_LAUNCH_DYN(kfn, grid, block, smem + 1024, stream_, a); }
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
    prof_begin(site, "matvec_pf", bytes);
    const dim3 qg((unsigned)nt), qb(1024);
    if (m.job[0].w.layout == LAYOUT_G4) {   // Q8_0 / Q4_0 weights: Q8_0 activation images, kernels_pf.h
        const int aw32 = pf_act_words_q32(m.K);
        // 8 tokens per workgroup on the matrix-core form while their images fit the CU's LDS (K <= 16384: 144 KB); wider rows take
        // the dot4 form with 4.  (Measured on config 3, profiles/r02_q80_chunk_sites.txt: 16 / 32 tokens per workgroup change
        // nothing — the kernel is bound by instruction issue, not by the weight traffic — and the dot4 form at 8 tokens is 6 % slower.)
        const int tb = (size_t)kPfTokens * aw32 * 4 <= (size_t)150 * 1024 ? kPfTokens : kPfTokens / 2;
        const bool mfma = tb == kPfTokens;          // lane sums on v_mfma_i32_4x4x4_16b_i8 (one image per token)
        const int paired = mfma ? 0 : 1;
        if (m.K <= 12288) CT_LAUNCH((pf_quantize_q80_kernel<12288>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, acts_, aw32, m.norm_b, paired);
        else CT_LAUNCH((pf_quantize_q80_kernel<32768>), qg, qb, stream_, x, ldx, m.norm_w, m.K, m.pro, m.eps, acts_, aw32, m.norm_b, paired);
        PfArgs a;
        a.m = m;
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
        const size_t smem = (size_t)tb * aw32 * 4;
#define PF32(TBV, GUV) do { \
            auto kfn = matvec_pf_kernel<TBV, GUV>; \
            CT_OPTIN_ONCE(kfn, (size_t)150 * 1024); \
            CT_LAUNCH_DYN(kfn, grid, block, smem, stream_, a); } while (0)
        if (mfma) {
            if (m.gateup) { auto kfn = matvec_pfm_kernel<true>; CT_OPTIN_ONCE(kfn, (size_t)150 * 1024); CT_LAUNCH_DYN(kfn, grid, block, smem, stream_, a); }
            else { auto kfn = matvec_pfm_kernel<false>; CT_OPTIN_ONCE(kfn, (size_t)150 * 1024); CT_LAUNCH_DYN(kfn, grid, block, smem, stream_, a); }
        } else { if (m.gateup) PF32(kPfTokens / 2, true); else PF32(kPfTokens / 2, false); }
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

// llm_build_llama (llama.cpp:2162-2491) for nt tokens of one batch_eval chunk at once: the same launches as token_step,
// each over rows [c0, c0 + nt) of the chunk (kernels_pf.h).  The cursor in d_state_ is at token c0 on entry.
bool Engine::chunk_step(int c0, int nt, bool want_logits, std::string& err) {
    if (hp_.falcon()) return chunk_step_falcon(c0, nt, want_logits, err);
    if (hp_.gpt2()) return chunk_step_gpt2(nt, want_logits, err);
    if (hp_.mpt()) return chunk_step_mpt(nt, want_logits, err);
    const int E = hp_.n_embd, G = hp_.n_embd_gqa(), F = hp_.n_ff, hd = hp_.head_dim();
    const int* d_pos = d_state_ + 1;
    if (l0_ == 0) {
        CT_LAUNCH(embed_row_kernel, dim3((unsigned)std::max(1, E / 256), (unsigned)nt), dim3(256), stream_, tok_embd_.raw, tok_embd_.type, E,
                  (const int*)d_tokens_, (const int*)d_state_, xb_);
    } else {
        HIP_OK(hipMemcpyAsync(xb_, xio_ + (size_t)c0 * E, (size_t)nt * E * 4, hipMemcpyDeviceToDevice, stream_));
    }
    MatvecArgs base = MatvecArgs();
    base.rope_cs = rope_cs_;
    base.pos = d_pos;
    base.n_ctx = n_ctx_;
    base.head_dim = hd;
    base.n_embd_gqa = G;
    base.v_stride = v_stride_;
    base.silu_tab = silu_tab_;
    base.eps = hp_.rms_eps;
    for (int il = l0_; il < l1_; ++il) {
        const Layer& L = layers_[il];
        uint16_t* kc = kcache_ + (size_t)(il - l0_) * n_ctx_ * G;
        uint16_t* vc = vcache_ + (size_t)(il - l0_) * v_stride_ * G;
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_RMSNORM; a.norm_w = L.attn_norm;
            a.q_f16 = q_f16_b_; a.kcache = kc; a.vcache = vc;
            set_jobs(a, {{&L.wq, EPI_ROPE_Q}, {&L.wk, EPI_ROPE_K}, {&L.wv, EPI_V}});
            if (!pf_matvec(a, xb_, E, nt, 0, 0, "qkv", (double)(L.wq.bytes + L.wk.bytes + L.wv.bytes), err)) return false;
        }
        if (site_on("attn_fused")) {
            prof_begin("attn_fused", "attn_fused_exact_kernel", 0.0);
            launch_attention(kc, vc, nt);
            prof_end();
        }
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_PLAIN; a.out = xb_; a.res = xb_;
            set_jobs(a, {{&L.wo, EPI_ADD}});
            if (!pf_matvec(a, attn_out_b_, E, nt, E, E, "wo", (double)L.wo.bytes, err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_RMSNORM; a.norm_w = L.ffn_norm; a.out = hb_;
            a.job[0].w = L.w_gate; a.job[0].pair0 = 0; a.job[0].epi = EPI_SILU_MUL;
            a.job[1].w = L.w_up; a.job[1].pair0 = 0; a.job[1].epi = EPI_SILU_MUL;
            a.njobs = 2; a.gateup = 1; a.n_pairs = F;
            if (L.w_gu.r2) { a.job[0].w = L.w_gu; a.njobs = 1; }   // K-quants: the fused matrix of the decode path (kernels_pg.h)
            if (!pf_matvec(a, xb_, E, nt, F, 0, "gate_up", (double)(L.w_gate.bytes + L.w_up.bytes), err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = F; a.pro = PRO_PLAIN; a.out = xb_; a.res = xb_;
            set_jobs(a, {{&L.w_down, EPI_ADD}});
            if (!pf_matvec(a, hb_, F, nt, E, E, "down", (double)L.w_down.bytes, err)) return false;
        }
    }
    if (l1_ < hp_.n_layer) {
        HIP_OK(hipMemcpyAsync(xio_ + (size_t)c0 * E, xb_, (size_t)nt * E * 4, hipMemcpyDeviceToDevice, stream_));
    } else if (want_logits) {   // the chunk's last token only (llama.cpp:2955-2959 keeps the last column)
        const float* xl = xb_ + (size_t)(nt - 1) * E;
        MatvecArgs a = base;
        a.K = E; a.pro = PRO_RMSNORM; a.x = xl; a.norm_w = output_norm_; a.out = d_logits_;
        set_jobs(a, {{&output_, EPI_STORE}});
        if (kq_can(a)) a.emb_out = d_emb_;   // generation 7 stores the final-norm output from its prologue
        else CT_LAUNCH((rmsnorm_f32_kernel<256>), dim3(1), dim3(256), stream_, xl, (const float*)output_norm_, d_emb_, E, hp_.rms_eps);
        if (!run_matvec(a, err)) return false;
    }
    CT_LAUNCH(advance_state_n_kernel, dim3(1), dim3(64), stream_, d_state_, nt);
    return true;
}


// llm_build_falcon (llama.cpp:2493-2798) for the nt tokens of a chunk: token_step_falcon's launches over rows of the chunk.
bool Engine::chunk_step_falcon(int c0, int nt, bool want_logits, std::string& err) {
    const int E = hp_.n_embd, G = hp_.n_embd_gqa(), F = hp_.n_ff, hd = hp_.head_dim();
    const int* d_pos = d_state_ + 1;
    if (l0_ == 0) {
        CT_LAUNCH(embed_row_kernel, dim3((unsigned)std::max(1, E / 256), (unsigned)nt), dim3(256), stream_, tok_embd_.raw, tok_embd_.type, E,
                  (const int*)d_tokens_, (const int*)d_state_, xb_);
    } else {
        HIP_OK(hipMemcpyAsync(xb_, xio_ + (size_t)c0 * E, (size_t)nt * E * 4, hipMemcpyDeviceToDevice, stream_));
    }
    MatvecArgs base = MatvecArgs();
    base.rope_cs = rope_cs_;
    base.pos = d_pos;
    base.n_ctx = n_ctx_;
    base.head_dim = hd;
    base.n_embd_gqa = G;
    base.v_stride = v_stride_;
    base.silu_tab = silu_tab_;
    base.gelu_tab = gelu_tab_;
    base.eps = hp_.rms_eps;
    for (int il = l0_; il < l1_; ++il) {
        const Layer& L = layers_[il];
        uint16_t* kc = kcache_ + (size_t)(il - l0_) * n_ctx_ * G;
        uint16_t* vc = vcache_ + (size_t)(il - l0_) * v_stride_ * G;
        {   // LayerNorm -> Q8_K -> fused QKV rows (f32, un-rotated), one row of E + 2G per token
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM;
            a.norm_w = L.attn_norm2 ? L.attn_norm2 : L.attn_norm;
            a.norm_b = L.attn_norm2 ? L.attn_norm2_b : L.attn_norm_b;
            a.out = qkv_tmp_b_;
            set_jobs(a, {{&L.wqkv, EPI_STORE}});
            if (!pf_matvec(a, xb_, E, nt, E + 2 * G, 0, "qkv", (double)L.wqkv.bytes, err)) return false;
        }
        if (site_on("rope_store"))
            CT_LAUNCH(falcon_rope_store_kernel, dim3((unsigned)(hp_.n_head + 2 * hp_.n_head_kv), (unsigned)nt), dim3((unsigned)(hd / 2)), stream_,
                      (const float*)qkv_tmp_b_, q_f16_b_, kc, vc, (const float*)rope_cs_, d_pos, hp_.n_head, hp_.n_head_kv, hd, n_ctx_,
                      v_stride_);
        if (site_on("attn_fused")) launch_attention(kc, vc, nt);
        {   // Wo, kept apart: the residual is added after the MLP
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_PLAIN; a.out = attn_proj_b_;
            set_jobs(a, {{&L.wo, EPI_STORE}});
            if (!pf_matvec(a, attn_out_b_, E, nt, E, 0, "wo", (double)L.wo.bytes, err)) return false;
        }
        {   // LayerNorm(attn_norm) -> Q8_K -> W_up -> GELU (parallel block: the MLP reads the attention norm)
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.norm_w = L.attn_norm; a.norm_b = L.attn_norm_b; a.out = hb_;
            set_jobs(a, {{&L.w_up, EPI_GELU}});
            if (!pf_matvec(a, xb_, E, nt, F, 0, "ffn_up", (double)L.w_up.bytes, err)) return false;
        }
        {   // W_down -> (ffn + attn_out) + x
            MatvecArgs a = base;
            a.K = F; a.pro = PRO_PLAIN; a.out = xb_; a.res = attn_proj_b_; a.res2 = xb_;
            set_jobs(a, {{&L.w_down, EPI_ADD2}});
            if (!pf_matvec(a, hb_, F, nt, E, E, "down", (double)L.w_down.bytes, err)) return false;
        }
    }
    if (l1_ < hp_.n_layer) {
        HIP_OK(hipMemcpyAsync(xio_ + (size_t)c0 * E, xb_, (size_t)nt * E * 4, hipMemcpyDeviceToDevice, stream_));
    } else if (want_logits) {
        const float* xl = xb_ + (size_t)(nt - 1) * E;
        CT_LAUNCH((layernorm_f32_kernel<256>), dim3(1), dim3(256), stream_, xl, (const float*)output_norm_, (const float*)output_norm_b_,
                  d_emb_, E, hp_.rms_eps);
        MatvecArgs a = base;
        a.K = E; a.pro = PRO_LAYERNORM; a.x = xl; a.norm_w = output_norm_; a.norm_b = output_norm_b_; a.out = d_logits_;
        set_jobs(a, {{&output_, EPI_STORE}});
        if (!run_matvec(a, err)) return false;
    }
    CT_LAUNCH(advance_state_n_kernel, dim3(1), dim3(64), stream_, d_state_, nt);
    return true;
}

// gpt2_eval (models/llms/gpt2.cc:391-699) for the nt tokens of a chunk: token_step_gpt2's launches over rows of the chunk; the
// K / V rows of all its tokens are appended to the F32 cache before the attention launch.
bool Engine::chunk_step_gpt2(int nt, bool want_logits, std::string& err) {
    const int E = hp_.n_embd, F = hp_.n_ff, hd = hp_.head_dim();
    const int* d_pos = d_state_ + 1;
    CT_LAUNCH(embed_row_kernel, dim3((unsigned)std::max(1, E / 256), (unsigned)nt), dim3(256), stream_, tok_embd_.raw, tok_embd_.type, E,
              (const int*)d_tokens_, (const int*)d_state_, xb_, (const float*)wpe_);
    MatvecArgs base = MatvecArgs();
    base.pos = d_pos;
    base.n_ctx = n_ctx_;
    base.head_dim = hd;
    base.n_embd_gqa = E;
    base.v_stride = v_stride_;
    base.silu_tab = silu_tab_;
    base.gelu_tab = gelu_tab_;
    base.eps = hp_.rms_eps;
    const float kq_scale = (float)(1.0 / sqrt((double)((float)E / (float)hp_.n_head)));   // gpt2.cc:540-543 (see launch_attention)
    for (int il = 0; il < hp_.n_layer; ++il) {
        const Layer& L = layers_[il];
        float* km = kmem_ + (size_t)il * n_ctx_ * E;
        float* vm = vmem_ + (size_t)il * n_ctx_ * E;
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.norm_w = L.attn_norm; a.norm_b = L.attn_norm_b;
            a.out = qkv_tmp_b_; a.bias = L.b_qkv;
            set_jobs(a, {{&L.wqkv, EPI_BIAS_STORE}});
            if (!pf_matvec(a, xb_, E, nt, 3 * E, 0, "qkv", (double)L.wqkv.bytes, err)) return false;
        }
        CT_LAUNCH(gpt2_kv_append_kernel, dim3((unsigned)nt), dim3(256), stream_, (const float*)qkv_tmp_b_, km, vm, d_pos, E);
        CT_LAUNCH(attn_f32_exact_kernel, dim3((unsigned)hp_.n_head, (unsigned)nt), dim3(256), stream_, (const float*)qkv_tmp_b_, km, vm,
                  attn_out_b_, (const uint16_t*)exp_tab_, d_pos, (const int*)(d_state_ + 2), E, hd, kq_scale, 0);
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_PLAIN; a.out = xb_; a.res = xb_; a.bias = L.b_wo;
            set_jobs(a, {{&L.wo, EPI_BIAS_ADD}});
            if (!pf_matvec(a, attn_out_b_, E, nt, E, E, "wo", (double)L.wo.bytes, err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.norm_w = L.ffn_norm; a.norm_b = L.ffn_norm_b; a.out = hb_; a.bias = L.b_up;
            set_jobs(a, {{&L.w_up, EPI_BIAS_GELU}});
            if (!pf_matvec(a, xb_, E, nt, F, 0, "ffn_up", (double)L.w_up.bytes, err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = F; a.pro = PRO_PLAIN; a.out = xb_; a.res = xb_; a.bias = L.b_down;
            set_jobs(a, {{&L.w_down, EPI_BIAS_ADD}});
            if (!pf_matvec(a, hb_, F, nt, E, E, "down", (double)L.w_down.bytes, err)) return false;
        }
    }
    if (want_logits) {
        MatvecArgs a = base;
        a.K = E; a.pro = PRO_LAYERNORM; a.x = xb_ + (size_t)(nt - 1) * E; a.norm_w = output_norm_; a.norm_b = output_norm_b_; a.out = d_logits_;
        set_jobs(a, {{&output_, EPI_STORE}});
        if (!run_matvec(a, err)) return false;
    }
    CT_LAUNCH(advance_state_n_kernel, dim3(1), dim3(64), stream_, d_state_, nt);
    return true;
}


// mpt_eval for the nt tokens of a chunk: token_step_mpt's launches over the rows of the chunk.
bool Engine::chunk_step_mpt(int nt, bool want_logits, std::string& err) {
    const int E = hp_.n_embd, F = hp_.n_ff, hd = hp_.head_dim();
    const int* d_pos = d_state_ + 1;
    CT_LAUNCH(embed_row_kernel, dim3((unsigned)std::max(1, E / 256), (unsigned)nt), dim3(256), stream_, tok_embd_.raw, tok_embd_.type, E,
              (const int*)d_tokens_, (const int*)d_state_, xb_);
    MatvecArgs base = MatvecArgs();
    base.pos = d_pos;
    base.n_ctx = n_ctx_;
    base.head_dim = hd;
    base.n_embd_gqa = E;
    base.v_stride = v_stride_;
    base.silu_tab = silu_tab_;
    base.gelu_tab = gelu_tab_;
    base.eps = hp_.rms_eps;
    for (int il = 0; il < hp_.n_layer; ++il) {
        const Layer& L = layers_[il];
        uint16_t* kc = kcache_ + (size_t)il * n_ctx_ * E;
        uint16_t* vc = vcache_ + (size_t)il * v_stride_ * E;
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.norm_w = L.attn_norm; a.norm_b = L.attn_norm_b; a.out = qkv_tmp_b_;
            set_jobs(a, {{&L.wqkv, EPI_STORE}});
            if (!pf_matvec(a, xb_, E, nt, 3 * E, 0, "qkv", (double)L.wqkv.bytes, err)) return false;
        }
        CT_LAUNCH(mpt_store_kernel, dim3((unsigned)(3 * hp_.n_head), (unsigned)nt), dim3((unsigned)hd), stream_, (const float*)qkv_tmp_b_, q_f16_b_,
                  kc, vc, d_pos, hp_.n_head, hd, n_ctx_, v_stride_, clip_qkv_);
        launch_attention(kc, vc, nt);
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_PLAIN; a.out = xb_; a.res = xb_;
            set_jobs(a, {{&L.wo, EPI_ADD}});
            if (!pf_matvec(a, attn_out_b_, E, nt, E, E, "wo", (double)L.wo.bytes, err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.norm_w = L.ffn_norm; a.norm_b = L.ffn_norm_b; a.out = hb_;
            set_jobs(a, {{&L.w_up, EPI_GELU}});
            if (!pf_matvec(a, xb_, E, nt, F, 0, "ffn_up", (double)L.w_up.bytes, err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = F; a.pro = PRO_PLAIN; a.out = xb_; a.res = xb_;
            set_jobs(a, {{&L.w_down, EPI_ADD}});
            if (!pf_matvec(a, hb_, F, nt, E, E, "down", (double)L.w_down.bytes, err)) return false;
        }
    }
    if (want_logits) {
        MatvecArgs a = base;
        a.K = E; a.pro = PRO_LAYERNORM; a.x = xb_ + (size_t)(nt - 1) * E; a.norm_w = output_norm_; a.norm_b = output_norm_b_; a.out = d_logits_;
        set_jobs(a, {{&output_, EPI_STORE}});
        if (!run_matvec(a, err)) return false;
    }
    CT_LAUNCH(advance_state_n_kernel, dim3(1), dim3(64), stream_, d_state_, nt);
    return true;
}

// A whole-model handle launches nothing in chunk_step that depends on c0 (the cursor lives in d_state_), so the ~300 launches
// of a chunk shape seen before are replayed from a graph: the first use of a shape runs eagerly (it also performs the
// one-time dynamic-LDS opt-ins), the second captures.
bool Engine::run_chunk(int c0, int nt, bool want_logits, std::string& err) {
    chunk_below_128_ = req_past_ + c0 + nt <= 128;   // positions of this chunk: [n_past + c0, n_past + c0 + nt)
#ifndef CT_EMU
    if (use_graph_ && !prof_ && !only_site_) {
        // the attention kernel depends on the flag; a pipeline stage (not the whole model) copies its rows from / to the hand-off
        // buffer at offset c0, so its graphs are per (c0, shape) — a request's micro-batches start at the same offsets every time
        const bool whole = l0_ == 0 && l1_ == hp_.n_layer;
        const long long key = ((long long)(whole ? 0 : c0 + 1) << 24) | (long long)(4 * nt + 2 * (chunk_below_128_ ? 1 : 0) + (want_logits ? 1 : 0));
        auto it = chunk_graphs_.find(key);
        if (it == chunk_graphs_.end() && chunk_seen_[key]++ >= 1) {
            hipGraph_t g = nullptr;
            HIP_OK(hipStreamBeginCapture(stream_, hipStreamCaptureModeGlobal));
            const bool ok = chunk_step(c0, nt, want_logits, err);
            const hipError_t e = hipStreamEndCapture(stream_, &g);
            if (!ok) return false;
            if (e != hipSuccess) { err = std::string("hipStreamEndCapture (chunk) failed: ") + hipGetErrorString(e); return false; }
            hipGraphExec_t ex = nullptr;
            HIP_OK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
            HIP_OK(hipGraphDestroy(g));
            it = chunk_graphs_.emplace(key, ex).first;
        }
        if (it != chunk_graphs_.end()) {
            HIP_OK(hipGraphLaunch(it->second, stream_));
            return true;
        }
    }
#endif
    return chunk_step(c0, nt, want_logits, err);
}

bool Engine::token_step(bool want_logits, std::string& err) {
    if (hp_.falcon()) return token_step_falcon(want_logits, err);
    if (hp_.gpt2()) return token_step_gpt2(want_logits, err);
    if (hp_.mpt()) return token_step_mpt(want_logits, err);
    const int E = hp_.n_embd, G = hp_.n_embd_gqa(), F = hp_.n_ff, hd = hp_.head_dim();
    const int* d_pos = d_state_ + 1;
    if (l0_ == 0) {
        if (site_on("embed")) {
        prof_begin("embed", "embed_row_kernel", (double)ggml_row_bytes(tok_embd_.type, E));
        CT_LAUNCH(embed_row_kernel, dim3((unsigned)std::max(1, E / 256)), dim3(256), stream_, tok_embd_.raw, tok_embd_.type, E,
                  (const int*)d_tokens_, (const int*)d_state_, x_);
        prof_end();
        }
    } else {  // inner stage: this token's residual-stream row was handed over by the previous stage
        CT_LAUNCH(stage_row_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), stream_, (const float*)xio_, x_, E,
                  (const int*)d_state_, 0);
    }
    MatvecArgs base = MatvecArgs();
    base.rope_cs = rope_cs_;
    base.pos = d_pos;
    base.n_ctx = n_ctx_;
    base.head_dim = hd;
    base.n_embd_gqa = G;
    base.v_stride = v_stride_;
    base.silu_tab = silu_tab_;
    base.eps = hp_.rms_eps;
    base.dbg = env_int("CT_AMD_DBG", 0);
    base.dbg_sink = scores_;
    for (int il = l0_; il < l1_; ++il) {
        const Layer& L = layers_[il];
        uint16_t* kc = kcache_ + (size_t)(il - l0_) * n_ctx_ * G;
        uint16_t* vc = vcache_ + (size_t)(il - l0_) * v_stride_ * G;
        {   // RMSNorm -> Q8_K -> {Wq,Wk,Wv} -> RoPE -> fp16 Q / KV-cache append
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_RMSNORM; a.x = x_; a.norm_w = L.attn_norm;
            a.q_f16 = q_f16_; a.kcache = kc; a.vcache = vc;
            set_jobs(a, {{&L.wq, EPI_ROPE_Q}, {&L.wk, EPI_ROPE_K}, {&L.wv, EPI_V}});  // types may differ per matrix
            apply_trace(a, "qkv");
            if (site_on("qkv")) {
                prof_begin("qkv", "matvec", (double)(L.wq.bytes + L.wk.bytes + L.wv.bytes));
                if (!run_matvec(a, err)) return false;
                prof_end();
            }
            debug_dump("1qkv", il);
        }
        if (site_on("attn_fused")) {
            prof_begin("attn_fused", "attn_fused_exact_kernel", 0.0);
            launch_attention(kc, vc);
            prof_end();
        }
        debug_dump("2attn", il);
        {   // Q8_K(attn) -> Wo -> + residual
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_PLAIN; a.x = attn_out_; a.out = x_; a.res = x_;
            set_jobs(a, {{&L.wo, EPI_ADD}});
            apply_trace(a, "wo");
            if (site_on("wo")) {
                prof_begin("wo", "matvec", (double)L.wo.bytes);
                if (!run_matvec(a, err)) return false;
                prof_end();
            }
            debug_dump("3wo", il);
        }
        {   // RMSNorm -> Q8_K -> {W_gate, W_up} -> SiLU(gate)*up
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_RMSNORM; a.x = x_; a.norm_w = L.ffn_norm; a.out = h_;
            if (L.w_gu.r9) {   // one job, the fused matrix (LAYOUT_L9 arena)
                a.job[0].w = L.w_gu; a.job[0].pair0 = 0; a.job[0].epi = EPI_SILU_MUL;
                a.njobs = 1; a.gateup = 1; a.n_pairs = F;
            } else {
                a.job[0].w = L.w_gate; a.job[0].pair0 = 0; a.job[0].epi = EPI_SILU_MUL;
                a.job[1].w = L.w_up; a.job[1].pair0 = 0; a.job[1].epi = EPI_SILU_MUL;
                a.njobs = 2; a.gateup = 1; a.n_pairs = F;
            }
            apply_trace(a, "gate_up");
            if (site_on("gate_up")) {
                prof_begin("gate_up", "matvec", (double)(L.w_gate.bytes + L.w_up.bytes));
                if (!run_matvec(a, err)) return false;
                prof_end();
            }
            debug_dump("4gateup", il);
        }
        {   // Q8_K(h) -> W_down -> + residual
            MatvecArgs a = base;
            a.K = F; a.pro = PRO_PLAIN; a.x = h_; a.out = x_; a.res = x_;
            set_jobs(a, {{&L.w_down, EPI_ADD}});
            apply_trace(a, "down");
            if (site_on("down")) {
                prof_begin("down", "matvec_k12288", (double)L.w_down.bytes);
                if (!run_matvec(a, err)) return false;
                prof_end();
            }
            debug_dump("5down", il);
        }
    }
    if (l1_ < hp_.n_layer) {  // hand this token's residual-stream row to the next stage
        CT_LAUNCH(stage_row_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), stream_, (const float*)x_, xio_, E,
                  (const int*)d_state_, 1);
    } else if (want_logits) {
        MatvecArgs a = base;
        a.K = E; a.pro = PRO_RMSNORM; a.x = x_; a.norm_w = output_norm_; a.out = d_logits_;
        set_jobs(a, {{&output_, EPI_STORE}});
        if (kq_can(a)) a.emb_out = d_emb_;   // generation 7 stores the final-norm output from its prologue
        else if (!only_site_)
            CT_LAUNCH((rmsnorm_f32_kernel<256>), dim3(1), dim3(256), stream_, (const float*)x_, (const float*)output_norm_, d_emb_, E,
                      hp_.rms_eps);
        apply_trace(a, "lm_head");
        if (site_on("lm_head")) {
            prof_begin("lm_head", "matvec", (double)output_.bytes);
            if (!run_matvec(a, err)) return false;
            prof_end();
        }
    }
    if (!only_site_) CT_LAUNCH(advance_state_kernel, dim3(1), dim3(64), stream_, d_state_);
    if (dump_dir_) ++dump_seq_;
    return true;
}

// llm_build_falcon (llama.cpp:2493-2798), one token: per layer
//   LayerNorm(attn_norm_2 or attn_norm) -> Q8_K -> fused Wqkv -> [neox RoPE, fp16 Q, KV append] -> attention -> Wo
//   LayerNorm(attn_norm) -> Q8_K -> W_up -> GELU table -> Q8_K -> W_down -> (ffn + attn_out) + x
bool Engine::token_step_falcon(bool want_logits, std::string& err) {
    const int E = hp_.n_embd, G = hp_.n_embd_gqa(), F = hp_.n_ff, hd = hp_.head_dim();
    const int* d_pos = d_state_ + 1;
    if (l0_ == 0) {
        if (site_on("embed")) {
            CT_LAUNCH(embed_row_kernel, dim3((unsigned)std::max(1, E / 256)), dim3(256), stream_, tok_embd_.raw, tok_embd_.type, E,
                      (const int*)d_tokens_, (const int*)d_state_, x_);
        }
    } else {  // inner pipeline stage: this token's residual-stream row was handed over by the previous stage
        CT_LAUNCH(stage_row_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), stream_, (const float*)xio_, x_, E,
                  (const int*)d_state_, 0);
    }
    MatvecArgs base = MatvecArgs();
    base.rope_cs = rope_cs_;
    base.pos = d_pos;
    base.n_ctx = n_ctx_;
    base.head_dim = hd;
    base.n_embd_gqa = G;
    base.v_stride = v_stride_;
    base.silu_tab = silu_tab_;
    base.gelu_tab = gelu_tab_;
    base.eps = hp_.rms_eps;
    base.dbg = env_int("CT_AMD_DBG", 0);
    base.dbg_sink = scores_;
    for (int il = l0_; il < l1_; ++il) {
        const Layer& L = layers_[il];
        uint16_t* kc = kcache_ + (size_t)(il - l0_) * n_ctx_ * G;
        uint16_t* vc = vcache_ + (size_t)(il - l0_) * v_stride_ * G;
        {   // LayerNorm -> Q8_K -> fused QKV rows (f32, un-rotated)
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.x = x_;
            a.norm_w = L.attn_norm2 ? L.attn_norm2 : L.attn_norm;
            a.norm_b = L.attn_norm2 ? L.attn_norm2_b : L.attn_norm_b;
            a.out = qkv_tmp_;
            set_jobs(a, {{&L.wqkv, EPI_STORE}});
            apply_trace(a, "qkv");
            if (site_on("qkv")) {
                prof_begin("qkv", "matvec", (double)L.wqkv.bytes);
                if (!run_matvec(a, err)) return false;
                prof_end();
            }
        }
        if (site_on("rope_store")) {
            prof_begin("rope_store", "falcon_rope_store_kernel", 0.0);
            CT_LAUNCH(falcon_rope_store_kernel, dim3((unsigned)(hp_.n_head + 2 * hp_.n_head_kv)), dim3((unsigned)(hd / 2)), stream_,
                      (const float*)qkv_tmp_, q_f16_, kc, vc, (const float*)rope_cs_, d_pos, hp_.n_head, hp_.n_head_kv, hd, n_ctx_,
                      v_stride_);
            prof_end();
        }
        if (site_on("attn_fused")) {
            prof_begin("attn_fused", "attn_fused_exact_kernel", 0.0);
            launch_attention(kc, vc);
            prof_end();
        }
        {   // Q8_K(attn) -> Wo  (kept apart: the residual is added after the MLP)
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_PLAIN; a.x = attn_out_; a.out = attn_proj_;
            set_jobs(a, {{&L.wo, EPI_STORE}});
            apply_trace(a, "wo");
            if (site_on("wo")) {
                prof_begin("wo", "matvec", (double)L.wo.bytes);
                if (!run_matvec(a, err)) return false;
                prof_end();
            }
        }
        {   // LayerNorm(attn_norm) -> Q8_K -> W_up -> GELU   (the MLP reads the attention norm: parallel block)
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.x = x_; a.norm_w = L.attn_norm; a.norm_b = L.attn_norm_b; a.out = h_;
            set_jobs(a, {{&L.w_up, EPI_GELU}});
            apply_trace(a, "ffn_up");
            if (site_on("ffn_up")) {
                prof_begin("ffn_up", "matvec", (double)L.w_up.bytes);
                if (!run_matvec(a, err)) return false;
                prof_end();
            }
        }
        {   // Q8_K(h) -> W_down -> (ffn + attn_out) + x
            MatvecArgs a = base;
            a.K = F; a.pro = PRO_PLAIN; a.x = h_; a.out = x_; a.res = attn_proj_; a.res2 = x_;
            set_jobs(a, {{&L.w_down, EPI_ADD2}});
            apply_trace(a, "down");
            if (site_on("down")) {
                prof_begin("down", "matvec", (double)L.w_down.bytes);
                if (!run_matvec(a, err)) return false;
                prof_end();
            }
        }
    }
    if (l1_ < hp_.n_layer) {  // hand this token's residual-stream row to the next stage
        CT_LAUNCH(stage_row_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), stream_, (const float*)x_, xio_, E,
                  (const int*)d_state_, 1);
    } else if (want_logits) {
        if (!only_site_)
            CT_LAUNCH((layernorm_f32_kernel<256>), dim3(1), dim3(256), stream_, (const float*)x_, (const float*)output_norm_,
                      (const float*)output_norm_b_, d_emb_, E, hp_.rms_eps);
        MatvecArgs a = base;
        a.K = E; a.pro = PRO_LAYERNORM; a.x = x_; a.norm_w = output_norm_; a.norm_b = output_norm_b_; a.out = d_logits_;
        set_jobs(a, {{&output_, EPI_STORE}});
        apply_trace(a, "lm_head");
        if (site_on("lm_head")) {
            prof_begin("lm_head", "matvec", (double)output_.bytes);
            if (!run_matvec(a, err)) return false;
            prof_end();
        }
    }
    if (!only_site_) CT_LAUNCH(advance_state_kernel, dim3(1), dim3(64), stream_, d_state_);
    return true;
}

// gpt2_eval (models/llms/gpt2.cc:391-699), one token: wte + wpe, then per layer
//   LN(ln_1) -> Q8_0 -> c_attn + b -> [append K,V rows to the F32 cache] -> F32 attention -> c_proj + b -> + x
//   LN(ln_2) -> Q8_0 -> c_fc + b -> GELU table -> Q8_0 -> c_proj + b -> + x;   final LN -> lm_head (tied wte)
bool Engine::token_step_gpt2(bool want_logits, std::string& err) {
    const int E = hp_.n_embd, F = hp_.n_ff, hd = hp_.head_dim();
    const int* d_pos = d_state_ + 1;
    CT_LAUNCH(embed_row_kernel, dim3((unsigned)std::max(1, E / 256)), dim3(256), stream_, tok_embd_.raw, tok_embd_.type, E,
              (const int*)d_tokens_, (const int*)d_state_, x_, (const float*)wpe_);
    MatvecArgs base = MatvecArgs();
    base.pos = d_pos;
    base.n_ctx = n_ctx_;
    base.head_dim = hd;
    base.n_embd_gqa = E;
    base.v_stride = v_stride_;
    base.silu_tab = silu_tab_;
    base.gelu_tab = gelu_tab_;
    base.eps = hp_.rms_eps;
    base.dbg_sink = scores_;
    const float kq_scale = (float)(1.0 / sqrt((double)((float)E / (float)hp_.n_head)));   // gpt2.cc:540-543 (see launch_attention)
    for (int il = 0; il < hp_.n_layer; ++il) {
        const Layer& L = layers_[il];
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.x = x_; a.norm_w = L.attn_norm; a.norm_b = L.attn_norm_b;
            a.out = qkv_tmp_; a.bias = L.b_qkv;
            set_jobs(a, {{&L.wqkv, EPI_BIAS_STORE}});
            if (!run_matvec(a, err)) return false;
        }
        CT_LAUNCH(attn_f32_exact_kernel, dim3((unsigned)hp_.n_head), dim3(256), stream_, (const float*)qkv_tmp_,
                  kmem_ + (size_t)il * n_ctx_ * E, vmem_ + (size_t)il * n_ctx_ * E, attn_out_, (const uint16_t*)exp_tab_, d_pos,
                  (const int*)(d_state_ + 2), E, hd, kq_scale);
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_PLAIN; a.x = attn_out_; a.out = x_; a.res = x_; a.bias = L.b_wo;
            set_jobs(a, {{&L.wo, EPI_BIAS_ADD}});
            if (!run_matvec(a, err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.x = x_; a.norm_w = L.ffn_norm; a.norm_b = L.ffn_norm_b; a.out = h_; a.bias = L.b_up;
            set_jobs(a, {{&L.w_up, EPI_BIAS_GELU}});
            if (!run_matvec(a, err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = F; a.pro = PRO_PLAIN; a.x = h_; a.out = x_; a.res = x_; a.bias = L.b_down;
            set_jobs(a, {{&L.w_down, EPI_BIAS_ADD}});
            if (!run_matvec(a, err)) return false;
        }
    }
    if (want_logits) {
        MatvecArgs a = base;
        a.K = E; a.pro = PRO_LAYERNORM; a.x = x_; a.norm_w = output_norm_; a.norm_b = output_norm_b_; a.out = d_logits_;
        set_jobs(a, {{&output_, EPI_STORE}});
        if (!run_matvec(a, err)) return false;
    }
    CT_LAUNCH(advance_state_kernel, dim3(1), dim3(64), stream_, d_state_);
    return true;
}


// mpt_eval (models/llms/mpt.cc:365-590), one token: wte row, then per layer
//   norm * ln_1 -> Q8_0 -> Wqkv -> clamp -> fp16 Q / K / V -> fp16 attention with the ALiBi term -> out_proj -> + x
//   norm * ln_2 -> Q8_0 -> up_proj -> GELU table -> Q8_0 -> down_proj -> + x;   final norm * norm_f -> wte as the head
bool Engine::token_step_mpt(bool want_logits, std::string& err) {
    const int E = hp_.n_embd, F = hp_.n_ff, hd = hp_.head_dim();
    const int* d_pos = d_state_ + 1;
    CT_LAUNCH(embed_row_kernel, dim3((unsigned)std::max(1, E / 256)), dim3(256), stream_, tok_embd_.raw, tok_embd_.type, E,
              (const int*)d_tokens_, (const int*)d_state_, x_);
    MatvecArgs base = MatvecArgs();
    base.pos = d_pos;
    base.n_ctx = n_ctx_;
    base.head_dim = hd;
    base.n_embd_gqa = E;
    base.v_stride = v_stride_;
    base.silu_tab = silu_tab_;
    base.gelu_tab = gelu_tab_;
    base.eps = hp_.rms_eps;
    base.dbg_sink = scores_;
    for (int il = 0; il < hp_.n_layer; ++il) {
        const Layer& L = layers_[il];
        uint16_t* kc = kcache_ + (size_t)il * n_ctx_ * E;
        uint16_t* vc = vcache_ + (size_t)il * v_stride_ * E;
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.x = x_; a.norm_w = L.attn_norm; a.norm_b = L.attn_norm_b; a.out = qkv_tmp_;
            set_jobs(a, {{&L.wqkv, EPI_STORE}});
            if (!run_matvec(a, err)) return false;
        }
        CT_LAUNCH(mpt_store_kernel, dim3((unsigned)(3 * hp_.n_head)), dim3((unsigned)hd), stream_, (const float*)qkv_tmp_, q_f16_, kc, vc, d_pos,
                  hp_.n_head, hd, n_ctx_, v_stride_, clip_qkv_);
        launch_attention(kc, vc);
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_PLAIN; a.x = attn_out_; a.out = x_; a.res = x_;
            set_jobs(a, {{&L.wo, EPI_ADD}});
            if (!run_matvec(a, err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.x = x_; a.norm_w = L.ffn_norm; a.norm_b = L.ffn_norm_b; a.out = h_;
            set_jobs(a, {{&L.w_up, EPI_GELU}});
            if (!run_matvec(a, err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = F; a.pro = PRO_PLAIN; a.x = h_; a.out = x_; a.res = x_;
            set_jobs(a, {{&L.w_down, EPI_ADD}});
            if (!run_matvec(a, err)) return false;
        }
    }
    if (want_logits) {
        MatvecArgs a = base;
        a.K = E; a.pro = PRO_LAYERNORM; a.x = x_; a.norm_w = output_norm_; a.norm_b = output_norm_b_; a.out = d_logits_;
        set_jobs(a, {{&output_, EPI_STORE}});
        if (!run_matvec(a, err)) return false;
    }
    CT_LAUNCH(advance_state_kernel, dim3(1), dim3(64), stream_, d_state_);
    return true;
}

// One token step is ~6 launches per layer; replaying it from a hipGraph removes the host launch cost (eager goes
// host-bound below ~3 us per kernel — guide "graph-replay-floor").  Two graphs: with and without the lm_head tail.
bool Engine::ensure_graphs(std::string& err) {
#ifndef CT_EMU
    if (graph_step_) return true;
    for (int head = 0; head < 2; ++head) {
        hipGraph_t g = nullptr;
        HIP_OK(hipStreamBeginCapture(stream_, hipStreamCaptureModeGlobal));
        const bool ok = token_step(head == 1, err);
        hipError_t e = hipStreamEndCapture(stream_, &g);
        if (!ok) return false;
        if (e != hipSuccess) { err = std::string("hipStreamEndCapture failed: ") + hipGetErrorString(e); return false; }
        hipGraphExec_t ex = nullptr;
        HIP_OK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        HIP_OK(hipGraphDestroy(g));
        (head ? graph_step_head_ : graph_step_) = ex;
    }
#endif
    (void)err;
    return true;
}

bool Engine::eval(const int* tokens, int n, int n_past, std::string& err, int batch) {
    if (l0_ != 0 || l1_ != hp_.n_layer) { err = "this handle is a pipeline stage: use eval_stage"; return false; }
    return eval_stage(tokens, n, n_past, nullptr, nullptr, err, batch);
}

bool Engine::req_begin(const int* tokens, int n, int n_past, int batch, std::string& err) {
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
    HIP_OK(hipMemcpyAsync(d_state_, &h_scalars_[0], (size_t)(4 + n) * 4, hipMemcpyHostToDevice, stream_));   // cursor + token ids
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
    CT_LAUNCH(argmax_first_kernel, dim3(1), dim3(1024), stream_, (const float*)d_logits_, hp_.n_vocab, d_argmax_);
    HIP_OK(hipMemcpyAsync(&h_scalars_[n_ctx_ + 12], d_argmax_, 4, hipMemcpyDeviceToHost, stream_));
    outputs_on_host_ = false;
    return true;
}

void Engine::fetch_outputs() {
    if (outputs_on_host_ || !have_logits_) return;
    (void)hipSetDevice(device_);
    (void)hipMemcpy(h_logits_, d_logits_, ((size_t)hp_.n_vocab + hp_.n_embd) * 4, hipMemcpyDeviceToHost);
    outputs_on_host_ = true;
}

bool Engine::req_wait(int n, int n_past, std::string& err) {
    HIP_OK(hipSetDevice(device_));
    HIP_OK(hipStreamSynchronize(stream_));
    HIP_OK(hipGetLastError());
    have_logits_ = l1_ == hp_.n_layer;
    last_token_ = h_scalars_[4 + n - 1];
    last_pos_ = n_past + n - 1;
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
    if (last_pos_ < 0) { err = "nothing evaluated yet"; return false; }
    h_scalars_[0] = 0; h_scalars_[1] = last_pos_; h_scalars_[2] = last_pos_ + 1; h_scalars_[4] = last_token_;
    HIP_OK(hipMemcpyAsync(d_tokens_, &h_scalars_[4], 4, hipMemcpyHostToDevice, stream_));
    HIP_OK(hipMemcpyAsync(d_state_, &h_scalars_[0], 12, hipMemcpyHostToDevice, stream_));
    trace_site_ = site;
    const bool ok = token_step(true, err);
    trace_site_ = nullptr;
    if (!ok) return false;
    HIP_OK(hipStreamSynchronize(stream_));
    HIP_OK(hipMemcpy(out, trace_buf_, (size_t)std::min(n, 256) * 8, hipMemcpyDeviceToHost));
    return true;
#else
    (void)site; (void)out; (void)n; err = "needs the HIP build"; return false;
#endif
}

bool Engine::profile_decode(int iters, std::vector<LaunchStat>& out, std::string& err) {
    out.clear();
#ifndef CT_EMU
    HIP_OK(hipSetDevice(device_));
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
