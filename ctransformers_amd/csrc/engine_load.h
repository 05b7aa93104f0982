// Model loading of the engine: file staging and GPU repack, weight arenas, tables, KV cache and scratch — part of engine.cc (one
// translation unit); included there, inside namespace ctamd.  Not a stand-alone header.

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
        // the eight mins as 48 CONTIGUOUS bits of the pair W3:W2 (lane l takes min l with one 64-bit shift by 6 l), scales 0..4 in W1,
        // 5 and 6 whole in W3, scale 7 split 2 + 4 over the spare bits (every lane pays for the split once, none selects a word)
        const uint32_t W1 = sc[0] | (sc[1] << 6) | (sc[2] << 12) | (sc[3] << 18) | (sc[4] << 24) | ((sc[7] & 3u) << 30);
        const uint32_t W2 = mn[0] | (mn[1] << 6) | (mn[2] << 12) | (mn[3] << 18) | (mn[4] << 24) | ((mn[5] & 3u) << 30);
        const uint32_t W3 = (mn[5] >> 2) | (mn[6] << 4) | (mn[7] << 10) | (sc[5] << 16) | (sc[6] << 22) | ((sc[7] >> 2) << 28);
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
    const int ng = (nb + 3) / 4, bb = q8 ? 34 : 18, rec = q8 ? 1088 : 576, dbase = q8 ? 1024 : 512, nl = q8 ? 8 : 4;
    const long long n = (long long)((M + 7) / 8) * ng * 32;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int bi = (int)(i & 3), r = (int)((i >> 2) & 7);
        const long long tg = i >> 5;
        const int g = (int)(tg % ng), tl = (int)(tg / ng), row = tl * 8 + r;
        if (row >= M || g * 4 + bi >= nb) continue;   // zero rows / zero blocks behind a row's end (the buffer is cleared first)
        const uint8_t* blk = src + ((size_t)row * nb + (size_t)g * 4 + bi) * bb;
        uint8_t* rp = dst + (size_t)tg * rec;
        memcpy(rp + dbase + r * 8 + bi * 2, blk, 2);
        for (int l = 0; l < nl; ++l) memcpy(rp + (r * nl + l) * 16 + bi * 4, blk + 2 + 4 * l, 4);
    }
}

// falcon: the rows of attn_qkv (file layout, staged on the device) reordered inside every Q and K head so that the NEOX rotation pair (i, i + head_dim / 2)
// becomes rows (2i, 2i + 1) — the row pair the mat-vec kernels finish in neighbouring lanes (kernels_v9.h epilogue, MatvecArgs::rope_neox).  Every layout is
// built from the reordered copy; V rows keep their places.  One workgroup per destination row, 2-byte pieces (row bytes are even for every block type).
__global__ void __launch_bounds__(256) falcon_permute_rows_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int row_bytes, int n_rope_rows, int hd) {
    const int r = (int)blockIdx.x;
    int sr = r;
    if (r < n_rope_rows) { const int p = r % hd; sr = r - p + (p >> 1) + ((p & 1) ? (hd >> 1) : 0); }
    const uint16_t* s = (const uint16_t*)(src + (size_t)sr * row_bytes);
    uint16_t* d = (uint16_t*)(dst + (size_t)r * row_bytes);
    for (int i = (int)threadIdx.x; i < row_bytes / 2; i += 256) d[i] = s[i];
}

// ---- LAYOUT_M8 (quant.h; kernels_mm8.h) ------------------------------------------------------------------------------------------------------
// One file-layout K-quant block -> row r (0..31) of a (tile, K-step) record.
CT_HD static inline void place_m8(int type, uint8_t* rec, int r, const uint8_t* blk) {
    if (type != GT_Q6_K) {
        const uint8_t* q = blk + 4;
        uint32_t sc[8], mn[8];
        for (int jj = 0; jj < 8; ++jj) {   // reference packing: k_quants.c:306-314
            if (jj < 4) { sc[jj] = q[jj] & 63; mn[jj] = q[jj + 4] & 63; }
            else { sc[jj] = (q[jj + 4] & 0xF) | ((q[jj - 4] >> 6) << 4); mn[jj] = (q[jj + 4] >> 4) | ((q[jj] >> 6) << 4); }
        }
        const uint32_t W1 = sc[0] | (sc[1] << 6) | (sc[2] << 12) | (sc[3] << 18) | (sc[4] << 24) | ((mn[7] & 3u) << 30);
        const uint32_t W2 = sc[5] | (sc[6] << 6) | (sc[7] << 12) | (mn[0] << 18) | (mn[1] << 24) | (((mn[7] >> 2) & 3u) << 30);
        const uint32_t W3 = mn[2] | (mn[3] << 6) | (mn[4] << 12) | (mn[5] << 18) | (mn[6] << 24) | ((mn[7] >> 4) << 30);
        uint8_t* hdr = rec + r * 16;
        memcpy(hdr, blk, 4);
        memcpy(hdr + 4, &W1, 4); memcpy(hdr + 8, &W2, 4); memcpy(hdr + 12, &W3, 4);
        const uint8_t* qs = blk + (type == GT_Q4_K ? 16 : 48);
        uint8_t* dq = rec + (type == GT_Q4_K ? 512 : 1536);
        if (type == GT_Q5_K)
            for (int c = 0; c < 2; ++c) memcpy(rec + 512 + (c * 32 + r) * 16, blk + 16 + 16 * c, 16);
        for (int g = 0; g < 4; ++g)
            for (int c = 0; c < 2; ++c) memcpy(dq + g * 1024 + (c * 32 + r) * 16, qs + 32 * g + 16 * c, 16);
    } else {   // ql[128] | qh[64] | scales[16] | d
        for (int h = 0; h < 2; ++h)
            for (int o = 0; o < 2; ++o)
                for (int c = 0; c < 2; ++c) memcpy(rec + (2 * h + o) * 1024 + (c * 32 + r) * 16, blk + 64 * h + 32 * o + 16 * c, 16);
        for (int h = 0; h < 2; ++h)
            for (int c = 0; c < 2; ++c) memcpy(rec + 4096 + h * 1024 + (c * 32 + r) * 16, blk + 128 + 32 * h + 16 * c, 16);
        memcpy(rec + 6144 + r * 16, blk + 192, 16);
        memcpy(rec + 6656 + r * 2, blk + 208, 2);
    }
}
// One file-layout Q8_0 / Q4_0 block -> block jj (0..7) of row r of a (tile, K-step) record.
CT_HD static inline void place_m8_b32(int type, uint8_t* rec, int r, int jj, const uint8_t* blk) {
    memcpy(rec + r * 16 + jj * 2, blk, 2);
    if (type == GT_Q8_0) {
        for (int c = 0; c < 2; ++c) memcpy(rec + 512 + jj * 1024 + (c * 32 + r) * 16, blk + 2 + 16 * c, 16);
    } else {
        memcpy(rec + 512 + jj * 512 + r * 16, blk + 2, 16);
    }
}
// row of the matrix (and which tensor: gate / up of a fused matrix) behind row r of tile `tile`
CT_HD static inline int m8_row_of(int tile, int r, bool fused, bool& second) {
    second = fused && r >= 16;
    return fused ? 16 * tile + (r & 15) : 32 * tile + r;
}
__global__ void __launch_bounds__(256) repack_m8_kernel(int type, const uint8_t* __restrict__ sa, const uint8_t* __restrict__ sb, uint8_t* __restrict__ dst,
                                                        int M, int nbf, int n_tiles) {
    const bool b32 = is_block32(type);
    const int ns = b32 ? (nbf + 7) / 8 : nbf, per = b32 ? 8 : 1, bb = ggml_block_bytes(type), rec = m8_record_bytes(type);
    const long long n = (long long)n_tiles * ns * 32 * per;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int jj = (int)(i % per), r = (int)((i / per) & 31);
        const long long ts = i / per / 32;
        const int s = (int)(ts % ns), tile = (int)(ts / ns);
        bool second;
        const int row = m8_row_of(tile, r, sb != nullptr, second), b = b32 ? 8 * s + jj : s;
        if (row >= M || b >= nbf) continue;   // zero rows / zero blocks (the arena is cleared first)
        const uint8_t* blk = (second ? sb : sa) + ((size_t)row * nbf + b) * bb;
        uint8_t* rp = dst + (size_t)ts * rec;
        if (b32) place_m8_b32(type, rp, r, jj, blk);
        else place_m8(type, rp, r, blk);
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
    if (perm_scratch_) { (void)hipFree(perm_scratch_); perm_scratch_ = nullptr; perm_scratch_bytes_ = 0; }
    if (dev_file_) { (void)hipFree(dev_file_); dev_file_ = nullptr; }
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

// LAYOUT_M8 arenas (kernels_mm8.h) of the matrices in `parts`, one allocation; `fuse`: pairs (gate, up) become ONE matrix of tiles
// (16 gate rows | 16 up rows), described by the pair's DevMat.  Called only where the handle runs the order-free prompt kernels.
bool Engine::upload_m8(const std::vector<std::pair<const GgufTensor*, DevMat*>>& parts, bool fuse, std::string& err) {
    struct Plan { const GgufTensor* ta; const GgufTensor* tb; DevMat* m; size_t off; int type, K, M, nbf, ns, n_tiles; };
    std::vector<Plan> plan;
    size_t total = 0;
    for (size_t i = 0; i < parts.size(); i += fuse ? 2 : 1) {
        const GgufTensor* ta = parts[i].first;
        const GgufTensor* tb = fuse ? parts[i + 1].first : nullptr;
        if (!ta || (fuse && !tb)) { err = "missing tensor for the M8 layout"; return false; }
        if (!(is_kquant(ta->type) || is_block32(ta->type))) return true;   // other types keep the exact forms
        if (tb && (tb->type != ta->type || tb->ne[0] != ta->ne[0] || tb->ne[1] != ta->ne[1])) { err = "gate/up tensors differ in type or shape"; return false; }
        Plan p;
        p.ta = ta; p.tb = tb; p.m = parts[i].second; p.type = ta->type; p.K = (int)ta->ne[0]; p.M = (int)ta->ne[1];
        p.nbf = p.K / ggml_block_elems(p.type);
        p.ns = m8_steps(p.type, p.K);
        if (p.K % ggml_block_elems(p.type) || p.K > 32768) return true;
        p.n_tiles = tb ? (p.M + 15) / 16 : (p.M + 31) / 32;
        p.off = total;
        total += (size_t)p.n_tiles * p.ns * m8_record_bytes(p.type);
        plan.push_back(p);
    }
    if (plan.empty()) return true;
    uint8_t* d = nullptr;
    if (!dev_alloc(dev_allocs_, &d, total + 16384, err)) return false;   // + one record: the kernels request a K-step past a tile's last one (clamped, but keep the slack)
    if (dev_file_) {
        HIP_OK(hipMemsetAsync(d, 0, total + 16384, stream_));
        for (const Plan& p : plan) {
            const long long n = (long long)p.n_tiles * p.ns * 32 * (is_block32(p.type) ? 8 : 1);
            const unsigned gx = (unsigned)std::min<long long>((n + 255) / 256, 65535LL * 16);
            CT_LAUNCH(repack_m8_kernel, dim3(gx), dim3(256), stream_, p.type, staged(p.ta), p.tb ? staged(p.tb) : (const uint8_t*)nullptr, d + p.off, p.M, p.nbf, p.n_tiles);
        }
    } else {
        std::vector<uint8_t> st(total, 0);
        for (const Plan& p : plan) {
            const bool b32 = is_block32(p.type);
            const int bb = ggml_block_bytes(p.type), rec = m8_record_bytes(p.type);
            uint8_t* dst = st.data() + p.off;
            parallel_rows(p.n_tiles, [&](int t0, int t1) {
                for (int tile = t0; tile < t1; ++tile)
                    for (int r = 0; r < 32; ++r) {
                        bool second;
                        const int row = m8_row_of(tile, r, p.tb != nullptr, second);
                        if (row >= p.M) continue;
                        const uint8_t* src = (second ? p.tb->data : p.ta->data) + (size_t)row * p.nbf * bb;
                        for (int b = 0; b < p.nbf; ++b) {
                            if (b32) place_m8_b32(p.type, dst + ((size_t)tile * p.ns + (b >> 3)) * rec, r, b & 7, src + (size_t)b * bb);
                            else place_m8(p.type, dst + ((size_t)tile * p.ns + b) * rec, r, src + (size_t)b * bb);
                        }
                    }
            });
        }
        HIP_OK(hipMemset(d + total, 0, 16384));
        HIP_OK(hipMemcpy(d, st.data(), total, hipMemcpyHostToDevice));
    }
    for (const Plan& p : plan) p.m->m8 = d + p.off;
    return true;
}

bool Engine::upload_matrix(const GgufTensor* t, DevMat& m, bool keep_raw, std::string& err) {
    m.type = t->type;
    m.K = (int)t->ne[0];
    m.M = (int)t->ne[1];
    const int be = ggml_block_elems(t->type), bb = ggml_block_bytes(t->type);
    if (t->type == GT_F16 || t->type == GT_F32 || is_raw32(t->type)) {   // rows stay in file layout (kernels_f16.h, kernels_raw32.h); token steps only — prompts of such a handle run token by token
        const std::string tn = t->type == GT_F16 ? "F16" : t->type == GT_F32 ? "F32" : (t->type == GT_Q4_1 ? "Q4_1" : (t->type == GT_Q5_0 ? "Q5_0" : "Q5_1"));
        if (m.K % 32 || m.K > (t->type == GT_F32 ? 16384 : 32768)) { err = "tensor " + t->name + ": " + tn + " rows of " + std::to_string(m.K) + " elements are not supported"; return false; }
        m.nb = is_raw32(t->type) ? m.K / 32 : m.K; m.bytes = t->nbytes; m.layout = is_raw32(t->type) ? LAYOUT_RAW32 : LAYOUT_F16;
        uint8_t* d = nullptr;
        if (!dev_alloc(dev_allocs_, &d, t->nbytes + 64, err)) return false;
        if (dev_file_) HIP_OK(hipMemcpyAsync(d, staged(t), t->nbytes, hipMemcpyDeviceToDevice, stream_));
        else HIP_OK(hipMemcpy(d, t->data, t->nbytes, hipMemcpyHostToDevice));
        m.raw = d;
        has_raw_ = true;
        return true;
    }
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
        // Rows that are not whole groups of four blocks (real Falcon-7B: n_embd 4544 = 142 blocks): the last group of a row is padded
        // with zero blocks (d = 0) here, the activation images likewise (kernels_pf.h), the decode arena's last record too (upload_l9b)
        m.layout = LAYOUT_G4;
        const bool q8 = t->type == GT_Q8_0;
        const int n_tiles = (M + 7) / 8, ng = (nb + 3) / 4, rec = q8 ? 1088 : 576, dbase = q8 ? 1024 : 512;
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
                            if (g * 4 + i >= nb) continue;   // zero block behind the row's end
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
    // the reference takes any context_length (llama.cc:90-92).  Up to kMaxCtxFused the attention kernels keep a token's probability row
    // in LDS; above it the row lives in global memory, prompts run token by token and nothing is tuned (kernels_exact.h GPROB)
    if (n_ctx_ > (1 << 20)) { err = "context_length above 1048576 is not supported"; return false; }

    HIP_OK(hipStreamCreate(&stream_));
    {   // CT_AMD_PREFILL=fast: prompt chunks in the order-free form (kernels_mm8.h); exact: the bit-identical chunk kernels (kernels_pg.h / kernels_pf.h)
        const char* pm = getenv("CT_AMD_PREFILL");
        fast_pf_ = pm && *pm ? (!strcmp(pm, "fast") || !strcmp(pm, "1")) : kPrefillFastDefault;
    }
    bool r2_auto = true;   // mat() also makes the matrix's own R2C4 copy (false: the caller places several matrices in one arena)
    bool m8_auto = true;   // ... and, on a handle with the order-free prompt kernels, its LAYOUT_M8 copy (false: the output head, which only token steps read)
    const GgufTensor* t;
    auto mat = [&](const std::string& name, DevMat& m, int M, int K, bool raw = false) {
        t = f.tensor(name);
        if (!t) { err = "missing tensor " + name; return false; }
        if (t->ne[0] != K || t->ne[1] != M) { err = "bad shape for " + name; return false; }
        if (!upload_matrix(t, m, raw, err)) return false;
        if (is_kquant(t->type) && r2_auto && !upload_r2c4({{t, &m}}, false, err)) return false;
        if (is_block32(t->type) && r2_auto && !upload_l9b({{t, &m}}, false, err)) return false;
        if (fast_pf_ && m8_auto && !upload_m8({{t, &m}}, false, err)) return false;
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
            {   // RoPE, fp16 Q and the KV append in the QKV launch's epilogue (token steps): needs the NEOX pairs in neighbouring rows — reorder the staged
                // tensor before any layout is built from it (file bytes on the device only: CT_AMD_GPU_REPACK=0 keeps the separate launch)
                const GgufTensor* tq = f.tensor(p + "attn_qkv.weight");
                const int hd = hp_.head_dim();
                const bool types_ok = tq && (is_kquant(tq->type) || tq->type == GT_Q8_0 || tq->type == GT_Q4_0);
                if (i == l0_) falcon_fold_ = dev_file_ && types_ok && hd % 2 == 0 && (E % 2 == 0) && (G % 2 == 0) && env_int("CT_AMD_FALCON_FOLD", 1) != 0;
                if (falcon_fold_ && !types_ok) { err = "attn_qkv weight types differ between layers in a way the folded QKV epilogue does not take"; return false; }
                if (falcon_fold_) {
                    if (tq->ne[0] != E || tq->ne[1] != E + 2 * G) { err = "bad shape for " + p + "attn_qkv.weight"; return false; }
                    const size_t row_bytes = tq->nbytes / (size_t)(E + 2 * G);
                    if (perm_scratch_bytes_ < tq->nbytes) {
                        if (perm_scratch_) { HIP_OK(hipStreamSynchronize(stream_)); (void)hipFree(perm_scratch_); perm_scratch_ = nullptr; }
                        HIP_OK(hipMalloc((void**)&perm_scratch_, tq->nbytes));
                        perm_scratch_bytes_ = tq->nbytes;
                    }
                    uint8_t* st = const_cast<uint8_t*>(staged(tq));
                    HIP_OK(hipMemcpyAsync(perm_scratch_, st, tq->nbytes, hipMemcpyDeviceToDevice, stream_));
                    CT_LAUNCH(falcon_permute_rows_kernel, dim3((unsigned)(E + 2 * G)), dim3(256), stream_, (const uint8_t*)perm_scratch_, st, (int)row_bytes, E + G, hd);
                }
            }
            if (!mat(p + "attn_qkv.weight", L.wqkv, E + 2 * G, E) || !mat(p + "attn_output.weight", L.wo, E, E) ||
                !mat(p + "ffn_up.weight", L.w_up, F, E) || !mat(p + "ffn_down.weight", L.w_down, E, F))
                return false;
            if (falcon_fold_) {
                if (!L.wqkv.r9) { err = "attn_qkv without a LAYOUT_L9 arena"; return false; }
                const size_t unit_bytes = (size_t)l9_spu(L.wqkv.type, E) * l9_record_bytes(L.wqkv.type);
                L.wq_v = L.wqkv; L.wq_v.M = E; L.wq_v.bytes = L.wqkv.bytes / (size_t)(E + 2 * G) * (size_t)E;
                L.wk_v = L.wqkv; L.wk_v.M = G; L.wk_v.r9 = L.wqkv.r9 + (size_t)(E / 2) * unit_bytes; L.wk_v.bytes = L.wqkv.bytes / (size_t)(E + 2 * G) * (size_t)G;
                L.wv_v = L.wqkv; L.wv_v.M = G; L.wv_v.r9 = L.wqkv.r9 + (size_t)((E + G) / 2) * unit_bytes; L.wv_v.bytes = L.wk_v.bytes;
            }
        }
        if (l1_ == hp_.n_layer) {
            if (!upload_f32(f.tensor("output_norm.weight"), &output_norm_, E, err)) return false;
            if (!upload_f32(f.tensor("output_norm.bias"), &output_norm_b_, E, err)) return false;
            m8_auto = false;
            if (!mat("output.weight", output_, V, E)) return false;
            m8_auto = true;
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
        m8_auto = false;   // (the chunk launch reads the fused gate/up matrix: its LAYOUT_M8 copy is made below)
        if (!mat(p + "ffn_gate.weight", L.w_gate, F, E) || !mat(p + "ffn_up.weight", L.w_up, F, E)) return false;
        m8_auto = true;
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
        if (fast_pf_ && (is_kquant(L.w_gate.type) || is_block32(L.w_gate.type)) && F % 16 == 0 &&
            !upload_m8({{f.tensor(p + "ffn_gate.weight"), &L.w_gu}, {f.tensor(p + "ffn_up.weight"), &L.w_gu}}, true, err))
            return false;
    }
    if (l1_ == hp_.n_layer) {
        if (!upload_f32(f.tensor("output_norm.weight"), &output_norm_, E, err)) return false;
        m8_auto = false;
        if (!mat("output.weight", output_, V, E)) return false;
        m8_auto = true;
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
    // V rows (one per channel): 16-byte aligned, and NOT a power of two apart — with a context of 2^k positions the rows of a head would sit exactly
    // 2^(k+1) bytes apart and a token step's short reads of all of them (4096 rows x 280 bytes at position 140 of a 7B) fall on a fraction of the
    // memory channels: 64 halves of padding measured +0.9 % decode on the 7B (763.9 -> 771.4 tok/s; 32 .. 320 all within 0.3 % of it;
    // profiles/r06_v_row_padding.txt).  CT_AMD_V_PAD overrides (halves, multiples of 32).
    v_stride_ = (n_ctx_ + 31) / 32 * 32 + std::max(0, env_int("CT_AMD_V_PAD", 64)) / 32 * 32;
    const size_t k_elems = (size_t)(l1_ - l0_) * n_ctx_ * G, v_elems = (size_t)(l1_ - l0_) * v_stride_ * G;
    if (!dev_alloc(dev_allocs_, &kcache_, k_elems, err) || !dev_alloc(dev_allocs_, &vcache_, v_elems + 64, err)) return false;
    HIP_OK(hipMemset(kcache_, 0, k_elems * 2));
    HIP_OK(hipMemset(vcache_, 0, v_elems * 2));
    if (!dev_alloc(dev_allocs_, &x_, (size_t)E, err) || !dev_alloc(dev_allocs_, &attn_out_, (size_t)E, err) ||
        !dev_alloc(dev_allocs_, &h_, (size_t)F, err) || !dev_alloc(dev_allocs_, &q_f16_, (size_t)E, err) ||
        !dev_alloc(dev_allocs_, &scores_, (size_t)hp_.n_head * n_ctx_, err) ||
        !dev_alloc(dev_allocs_, &d_logits2_[0], (size_t)V + E, err) ||   // [logits | final-norm embedding]: one D2H copy per eval
        !dev_alloc(dev_allocs_, &d_logits2_[1], (size_t)V + E, err) ||   // the pair member a speculative continuation step writes (engine.h)
        !dev_alloc(dev_allocs_, &pick_ws_, 2 * 16 * 2048, err) ||
        !dev_alloc(dev_allocs_, &trace_buf_, 512, err) || !dev_alloc(dev_allocs_, &d_argmax2_[0], 8, err) || !dev_alloc(dev_allocs_, &d_state_, (size_t)n_ctx_ + 8, err))   // [cursor(4) | tokens]: one H2D copy
        return false;
    d_argmax2_[1] = d_argmax2_[0] + 1;
    HIP_OK(hipMemset(pick_ws_, 0, (size_t)2 * 16 * 2048 * 4));
    d_tokens_ = d_state_ + 4;
    HIP_OK(hipMemset(d_state_, 0, ((size_t)n_ctx_ + 8) * 4));   // [4 + n_ctx]: the token epoch (kernels_qa9.h)
    dbg_qa_timeout_ = env_int("CT_AMD_DBG_QA_TIMEOUT", 0);
    fuse_qa_ = env_int("CT_AMD_FUSE_QA", 1) != 0 && !hp_.legacy() && (!hp_.falcon() || falcon_fold_);
    if (fuse_qa_) {
        const size_t words = (size_t)(hp_.n_head + 2 * hp_.n_head_kv) * (hp_.head_dim() / 2) * 2;
        if (!dev_alloc(dev_allocs_, &xq_, words + 16, err)) return false;
        HIP_OK(hipMemset(xq_, 0, (words + 16) * 4));
    }
    attn_share_ = env_int("CT_AMD_ATTN_SHARE", 1) != 0 && !hp_.legacy();
    {   // order-free V*P of the long-context decode attention: opt-in, like the order-free prompt kernels (the default stays bit-identical)
        const char* da = getenv("CT_AMD_DECODE_ATTN");
        attn_free_ = da && (!strcmp(da, "fast") || !strcmp(da, "1")) && !hp_.legacy();
    }
    if (attn_share_ && n_ctx_ > 1024 && n_ctx_ <= kMaxCtxFused) {   // (the deep form of the decode attention: contexts above 1024)
        const size_t words = (size_t)hp_.n_head * n_ctx_ * 2;
        if (!dev_alloc(dev_allocs_, &xs_, words + 16, err)) return false;
        HIP_OK(hipMemset(xs_, 0, (words + 16) * 4));
    }
    if (has_raw_ && !dev_alloc(dev_allocs_, &f16_tmp_, (size_t)std::max(std::max(V, 2 * F), E + 2 * G) + 64, err)) return false;
    // prompt chunks (kernels_pg.h: K-quants; kernels_pf.h: Q8_0 / Q4_0): n_embd <= 12288, n_ff <= 32768
    pf_ok_ = E <= 12288 && F <= 32768 && n_ctx_ <= kMaxCtxFused && env_int("CT_AMD_PF", 1) != 0;
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
        pf_cap_ = kPfChunk;
        if (fast_pf_ && !hp_.legacy()) {   // every mat-mul site of every layer on the order-free kernels: chunks of up to kPfChunkFast tokens
            bool all = true;
            auto ok8 = [&](const DevMat& w, bool gu) { return w.m8 && (gu ? w.M % 16 == 0 : w.M % 32 == 0) && w.K <= 32768; };
            for (int i = l0_; i < l1_ && all; ++i) {
                const Layer& L = layers_[i];
                if (hp_.falcon()) all = ok8(L.wqkv, false) && ok8(L.wo, false) && ok8(L.w_up, false) && ok8(L.w_down, false);
                else all = ok8(L.wq, false) && ok8(L.wk, false) && ok8(L.wv, false) && ok8(L.wo, false) && ok8(L.w_gu, true) && ok8(L.w_down, false) &&
                           is_block32(L.wq.type) == is_block32(L.wk.type) && is_block32(L.wq.type) == is_block32(L.wv.type);
            }
            if (all) pf_cap_ = std::max(kPfChunk, std::min(kPfChunkFast, env_int("CT_AMD_PF_CAP", kPfChunkFast)) & ~31);
        }
        pf_chunk_ = std::max(pf_min_, std::min(pf_cap_, env_int("CT_AMD_PF_CHUNK", pf_cap_)));
        if (!dev_alloc(dev_allocs_, &xb_, (size_t)pf_cap_ * E, err) || !dev_alloc(dev_allocs_, &attn_out_b_, (size_t)pf_cap_ * E, err) ||
            !dev_alloc(dev_allocs_, &hb_, (size_t)pf_cap_ * F, err) || !dev_alloc(dev_allocs_, &q_f16_b_, (size_t)pf_cap_ * E, err))
            return false;
        pg_force_tg_ = env_int("CT_AMD_PG_TG", 0);
        if (!kq_model) {   // Q8_0 activation images of the Q8_0 / Q4_0 chunk kernel
            // + 32 images: a workgroup copies all the images of its token group (up to 32), the last group's reach past the chunk's end (their results are dropped)
            if (!dev_alloc(dev_allocs_, &acts_, (size_t)(kPfChunk + 32) * pf_act_words_q32((std::max(E, F) + 127) & ~127), err)) return false;
            HIP_OK(hipMemset(acts_, 0, (size_t)(kPfChunk + 32) * pf_act_words_q32((std::max(E, F) + 127) & ~127) * sizeof(int)));
        }
        if (kq_model || mixed_model) {   // 128 tokens of stage images per block and layout (kernels_pg.h PgStage)
            acts_h_half_ = (size_t)(std::max(E, F) / 256) * std::max((kPfChunk / 16) * PgStage<16>::BYTES, (kPfChunk / 32) * PgStage<32>::BYTES) + 4096;
            if (!dev_alloc(dev_allocs_, &acts_h_, 2 * acts_h_half_, err)) return false;
            HIP_OK(hipMemset(acts_h_, 0, 2 * acts_h_half_));   // token slots past the chunk's end are read (and their results dropped)
        }
        if (fast_pf_) {   // activation units of the order-free kernels: [K-steps of the widest input][token tiles of a chunk]
            acts8_bytes_ = (size_t)((std::max(E, F) + 255) / 256) * ((pf_cap_ + 31) / 32) * kMm8Unit + 4096;
            if (!dev_alloc(dev_allocs_, &acts8_, acts8_bytes_, err)) return false;
            HIP_OK(hipMemset(acts8_, 0, acts8_bytes_));
            if (const char* sh = getenv("CT_AMD_MM8_SHAPE")) { if (sscanf(sh, "%d,%d", &mm8_force_ntt_, &mm8_force_ks_) != 2) mm8_force_ntt_ = mm8_force_ks_ = 0; }
        }
        if (hp_.falcon() && (!dev_alloc(dev_allocs_, &qkv_tmp_b_, (size_t)pf_cap_ * (E + 2 * G), err) ||
                             !dev_alloc(dev_allocs_, &attn_proj_b_, (size_t)pf_cap_ * E, err)))
            return false;
        if (hp_.legacy() && !dev_alloc(dev_allocs_, &qkv_tmp_b_, (size_t)kPfChunk * 3 * E, err)) return false;
    }
    HIP_OK(hipHostMalloc(&h_logits_, ((size_t)V + E) * 4));
    h_emb_ = h_logits_ + V;
    HIP_OK(hipHostMalloc(&h_scalars_, ((size_t)n_ctx_ + 16) * 4));
    memset(h_scalars_, 0, ((size_t)n_ctx_ + 16) * 4);
    for (int b = 0; b < 2; ++b) {   // the head launch writes the greedy pick straight into this pinned word
#ifdef CT_EMU
        pick_host2_[b] = &h_scalars_[n_ctx_ + 12 + b];
#else
        void* dp = nullptr;
        HIP_OK(hipHostGetDevicePointer(&dp, &h_scalars_[n_ctx_ + 12 + b], 0));
        pick_host2_[b] = (int*)dp;
#endif
    }
    select_out(0);
#ifdef CT_EMU
    qa_err_ = &h_scalars_[n_ctx_ + 14];
#else
    {
        void* dp = nullptr;
        HIP_OK(hipHostGetDevicePointer(&dp, &h_scalars_[n_ctx_ + 14], 0));
        qa_err_ = (int*)dp;
    }
#endif
    use_graph_ = env_int("CT_AMD_GRAPH", 1) != 0;
    fold_on_ = env_int("CT_AMD_HEAD_FOLD", 1);
    spec_on_ = env_int("CT_AMD_SPEC", 1) != 0;
    if ((stamps_level_ = env_int("CT_AMD_STAMPS", 0)) != 0) {
        if (!dev_alloc(dev_allocs_, &stamps_, 40008, err)) return false;
        HIP_OK(hipMemset(stamps_, 0, 40008 * 8));
    }
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
    if (hp_.n_embd % 32) { err = "gpt2: n_embd must be a multiple of 32 (whole 32-element blocks)"; return false; }   // GPT-2 XL: 1600
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
        if (!is_block32(t->type) && !is_raw32(t->type) && t->type != GT_F16 && t->type != GT_F32) { err = name + ": only F32 / F16 / Q4_0 / Q4_1 / Q5_0 / Q5_1 / Q8_0 legacy weights are supported"; return false; }
        if (!upload_matrix(t, m, false, err)) return false;
        if (is_block32(t->type) && !upload_l9b({{t, &m}}, false, err)) return false;
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
    if (hp_.n_embd % 32) { err = "mpt: d_model must be a multiple of 32 (whole 32-element blocks)"; return false; }
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
        if (!is_block32(t->type) && !is_raw32(t->type) && t->type != GT_F16 && t->type != GT_F32) { err = name + ": only F32 / F16 / Q4_0 / Q4_1 / Q5_0 / Q5_1 / Q8_0 legacy weights are supported"; return false; }
        if (!upload_matrix(t, m, false, err)) return false;
        if (is_block32(t->type) && !upload_l9b({{t, &m}}, false, err)) return false;
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
