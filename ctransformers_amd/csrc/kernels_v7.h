// Mat-vec, generation 7 (K-quants, every K): waves that never wait for each other.
//
// Generations 5/6 split the K-blocks of an 8-row tile over the 16 waves of a workgroup and replayed the reference's f32
// fma chain (the only order-dependent part of a row's dot product, kernels_exact.h) from LDS chain storage behind
// workgroup barriers / LDS counters.  In-kernel traces showed what that costs per launch besides the streaming time: a
// prologue that cannot end before the first round of weights has landed, a pipeline bubble behind it, and a tail of
// block math + chain replay + epilogue after the last load — about 10 us per launch, more than the 1.5-8 us the bytes take.
//
// Here the unit of work is ONE PAIR OF ROWS, owned by one wave from its first byte to its epilogue:
//   * layout LAYOUT_R2C4 (quant.h): a record holds 2 rows x 4 consecutive K-blocks (8 block slots, the same 1152 / 1408 /
//     1680 bytes as the 8 file blocks it holds, fields grouped: kernels_kq.h img_load / img_to_regs); slot p = 4*row + c.
//     A row pair is `spu = ceil(nb / 4)` consecutive records = one contiguous stream (4.6 KB at K = 4096).  gate/up
//     launches use a fused matrix whose pair is (gate row r, up row r).
//   * a step = one record: lane (p, g) does the integer work of block 4*s + c of row `p >> 2` exactly as before (nibble
//     unpack, dot4 against the Q8_K image in LDS, 6-bit scales, the AVX-lane transpose-reduce) and leaves
//     (float)sumi[l(g)], y.d*d, -y.d*dmin, (float)prod in a 448-byte wave-private LDS buffer; after an LDS-only wait every
//     lane reads back the four blocks' operands of ITS chain and performs the reference's fmas in block order.  The chain
//     state lives in registers from step to step; nothing is shared between waves after the prologue barrier.
//   * a 4-deep register ring of records per wave keeps 16 x 4 x 1.1 KB = 73 KB per CU in flight; a wave walks its units
//     (unit = first + i * stride, interleaved over workgroups so every CU gets the same bytes +- one unit) as one flat
//     sequence of steps, the ring running across unit boundaries.  The loop that consumes the ring contains NO other vector
//     memory instruction and no conditional one: every step issues exactly one record's loads (past the end: the last
//     record again), results go to an LDS row and the fused epilogues (with their stores and table lookups) run after the
//     loop, their operands (residual, RoPE cos/sin) requested before it.  Only then does hipcc count the ring with
//     `s_waitcnt vmcnt(6)`; a store or a branch around a load anywhere in the loop makes it wait `vmcnt(0)` at every step
//     (loads and stores retire out of order on one counter), i.e. one memory latency per step and wave: 2.5 TB/s.
//   * the activation vector is requested BEFORE the first weight records (vmcnt is in order: the prologue then waits for
//     16 KB, not for the first 73 KB of weights), quantized once per workgroup (prologue of kernels_exact.h, unchanged
//     arithmetic), and the Q8_K image is stored with a per-block skew (q8w_of) that keeps the four blocks a wave reads at
//     once on different LDS banks.
//   * a launch with two weight types (attn_q/k = Q4_K, attn_v = Q6_K) gives each type its own waves of every workgroup,
//     split in proportion to the bytes.
#pragma once
#include "kernels_kq.h"

// Q8_K image of the activation vector (kernels_exact.h ActLdsX without the 16-sums nobody reads, q8 skewed).
template <int MAXK> struct ActLds7 {
    int q8[MAXK / 4 + MAXK / 128 + 16];
    float yd[MAXK / 256 + 4];   // + the padding slots of a last record (read, never used)
    int sb[MAXK / 32 + 32];
    double red[16];
};
// word offset of block b's 64 quant words: groups of four blocks are 264 words apart and blocks 2, 3 of a group are skewed
// by 8 words, so the b128 reads of a step (4 blocks, kernels_v7.h header) hit every bank once per 16-lane service group.
DEV int q8w_of(int b) { return 264 * (b >> 2) + 64 * (b & 3) + 8 * ((b >> 1) & 1); }

template <int MAXK, bool EW = true> struct ProRegs7 {
    static constexpr int ROUNDS = (MAXK / 256 + 63) / 64;
    // the norm weights are requested with the activations (16 more registers) — not in the two-type launches, whose two inlined
    // block loops leave no room for them (they spilled 16 registers there)
    static constexpr bool EARLY_W = EW && MAXK <= 16384;
    float4 v[ROUNDS][4];
    float4 w[EARLY_W ? ROUNDS : 1][4];
};

// Prologue part 1: request this thread's 16 consecutive activations (16 lanes per 256-block) and their norm weights —
// nothing waits here.  The first instructions of the kernel: whatever is requested later queues behind the weight stream.
template <int MAXK, bool EW> DEV void pro7_load(ProRegs7<MAXK, EW>& P, const float* __restrict__ x, const float* __restrict__ nw, int K, int pro) {
    const int tid = (int)threadIdx.x, sub = tid & 15, grp = tid >> 4;
    const int nblk = K >> 8;
    // Only the waves that own blocks load (for K = 4096: 4 of 16 — the others would put 100 KB of redundant requests in front
    // of the weight stream), under ONE wave-uniform branch; inside it every load is unconditional (block index clamped,
    // the activations standing in for absent norm weights): a per-load select between "load" and "constant" makes hipcc
    // branch around each load and wait for it on the spot.
    if (uniform_int(((int)threadIdx.x >> 6) * 4) >= nblk) return;
    const float* __restrict__ wsrc = pro != PRO_PLAIN ? nw : x;
#pragma unroll
    for (int rd = 0; rd < ProRegs7<MAXK, EW>::ROUNDS; ++rd) {
        int b = grp + rd * 64;
        b = b < nblk ? b : nblk - 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) P.v[rd][k] = *(const float4*)(x + b * 256 + sub * 16 + k * 4);
        if constexpr (ProRegs7<MAXK, EW>::EARLY_W) {
#pragma unroll
            for (int k = 0; k < 4; ++k) P.w[rd][k] = *(const float4*)(wsrc + b * 256 + sub * 16 + k * 4);
        }
    }
}

// Prologue part 2: (RMSNorm | LayerNorm | nothing) -> Q8_K into LDS.  Arithmetic of prologue_q8k_exact16 / _ln
// (kernels_exact.h; reference k_quants.c:1191-1226 with the build's fused fma, RMSNorm ggml.c:10700-10716, LayerNorm
// ggml.c:10605-10654).  Ends with a workgroup barrier.  `emb_out` (block 0 only): the normalised vector as f32 — the
// final-norm "embeddings" output of the ABI, produced by the lm_head launch (EMB instantiation) instead of a launch of its own.
template <int MAXK, bool LN, bool EMB, bool EW>
DEV void pro7_finish(ActLds7<MAXK>& L, ProRegs7<MAXK, EW>& P, const float* __restrict__ nw, const float* __restrict__ nbias, int K,
                     int pro, float eps, float* __restrict__ emb_out) {
    constexpr int ROUNDS = ProRegs7<MAXK, EW>::ROUNDS, NW = 16;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, sub = tid & 15, grp = tid >> 4;
    const int nblk = K >> 8;
    const bool wave_live = uniform_int(wv * 4) < nblk;   // dead waves skip the arithmetic (wave-uniform branches)
    float scale = 1.0f;
    if (pro == PRO_RMSNORM) {
        double s = 0.0;
        if (wave_live) {
#pragma unroll
            for (int rd = 0; rd < ROUNDS; ++rd) {
                if (grp + rd * 64 < nblk) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        s += (double)(P.v[rd][k].x * P.v[rd][k].x);
                        s += (double)(P.v[rd][k].y * P.v[rd][k].y);
                        s += (double)(P.v[rd][k].z * P.v[rd][k].z);
                        s += (double)(P.v[rd][k].w * P.v[rd][k].w);
                    }
                }
            }
            s = wave_sum_fast(s);
        }
        if (lane == 0) L.red[wv] = wave_live ? s : 0.0;
        __syncthreads();
        if (wave_live) {
            double tot = 0.0;
            for (int w = 0; w < NW; ++w) tot += L.red[w];
            const float mean = (float)(tot / (double)K);
            scale = 1.0f / sqrtf(mean + eps);
        }
    }
    if constexpr (LN) {
        if (pro == PRO_LAYERNORM) {
            double s1 = 0.0;
            if (wave_live) {
#pragma unroll
                for (int rd = 0; rd < ROUNDS; ++rd) {
                    if (grp + rd * 64 < nblk) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            s1 += (double)P.v[rd][k].x; s1 += (double)P.v[rd][k].y; s1 += (double)P.v[rd][k].z; s1 += (double)P.v[rd][k].w;
                        }
                    }
                }
                s1 = wave_sum_fast(s1);
            }
            if (lane == 0) L.red[wv] = wave_live ? s1 : 0.0;
            __syncthreads();
            double tot = 0.0;
            for (int w = 0; w < NW; ++w) tot += L.red[w];
            const float mean = (float)(tot / (double)K);
            __syncthreads();   // L.red is reused for the second moment
            double s2 = 0.0;
            if (wave_live) {
#pragma unroll
                for (int rd = 0; rd < ROUNDS; ++rd) {
                    if (grp + rd * 64 < nblk) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float4& q = P.v[rd][k];
                            q.x -= mean; q.y -= mean; q.z -= mean; q.w -= mean;
                            s2 += (double)(q.x * q.x); s2 += (double)(q.y * q.y); s2 += (double)(q.z * q.z); s2 += (double)(q.w * q.w);
                        }
                    }
                }
                s2 = wave_sum_fast(s2);
            }
            if (lane == 0) L.red[wv] = wave_live ? s2 : 0.0;
            __syncthreads();
            double tot2 = 0.0;
            for (int w = 0; w < NW; ++w) tot2 += L.red[w];
            const float variance = (float)(tot2 / (double)K);
            scale = 1.0f / sqrtf(variance + eps);
        }
    }
    if (wave_live) {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int b = grp + rd * 64;
            const bool live = b < nblk;            // uniform within a 16-lane row, may differ between rows of a wave
            float t[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float4 q = live ? P.v[rd][k] : float4{0.f, 0.f, 0.f, 0.f};
                if (live && pro != PRO_PLAIN) {
                    float4 w4;
                    if constexpr (ProRegs7<MAXK, EW>::EARLY_W) w4 = P.w[rd][k];
                    else w4 = *(const float4*)(nw + b * 256 + sub * 16 + k * 4);
                    q.x = (q.x * scale) * w4.x;
                    q.y = (q.y * scale) * w4.y;
                    q.z = (q.z * scale) * w4.z;
                    q.w = (q.w * scale) * w4.w;
                    if constexpr (LN) {
                        if (pro == PRO_LAYERNORM) {
                            const float4 b4 = *(const float4*)(nbias + b * 256 + sub * 16 + k * 4);
                            q.x += b4.x; q.y += b4.y; q.z += b4.z; q.w += b4.w;
                        }
                    }
                    if constexpr (EMB) {   // a store here is a pending write at the entry of the streaming loop: lm_head instantiation only
                        if (emb_out && blockIdx.x == 0) *(float4*)(emb_out + b * 256 + sub * 16 + k * 4) = q;
                    }
                }
                t[4 * k] = q.x; t[4 * k + 1] = q.y; t[4 * k + 2] = q.z; t[4 * k + 3] = q.w;
            }
            float am = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) am = fmaxf(am, fabsf(t[e]));
            float amax = am;
            amax = fmaxf(amax, lane_xor1(amax));
            amax = fmaxf(amax, lane_xor2(amax));
            amax = fmaxf(amax, lane_xor4(amax));
            amax = fmaxf(amax, lane_xor8(amax));
            // first element (lowest index) attaining amax keeps its sign
            const unsigned long long hit = __ballot(am == amax);
            const unsigned row_bits = (unsigned)((hit >> (lane & 48)) & 0xFFFFu);
            const int first = (lane & 48) + (__ffsll((unsigned long long)row_bits) - 1);
            float mine = 0.0f;
#pragma unroll
            for (int e = 15; e >= 0; --e) mine = (fabsf(t[e]) == amax) ? t[e] : mine;
            const float maxv = __shfl(mine, first);
            int packed[4] = {0, 0, 0, 0}, s16 = 0;
            float d = 0.0f;
            if (amax != 0.0f) {
                const float iscale = -128.f / maxv;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int q = ((int)f32_to_bits(fmaf(iscale, t[e], 12582912.f)) & 0x007fffff) - 0x00400000;
                    q = q > 127 ? 127 : q;
                    packed[e >> 2] |= (q & 0xff) << (8 * (e & 3));
                    s16 += q;
                }
                d = 1.0f / iscale;
            }
            const int s32 = s16 + lane_xor1(s16);
            if (live) {
                int* qd = &L.q8[q8w_of(b) + sub * 4];
#pragma unroll
                for (int k = 0; k < 4; ++k) qd[k] = packed[k];
                if ((sub & 1) == 0) L.sb[b * 8 + (sub >> 1)] = s32;
                if (sub == 0) L.yd[b] = d;
            }
        }
    }
    __syncthreads();
}

// Wave-private exchange buffer of one step: chain operands of the step's 2 rows x 4 blocks.
struct WaveBuf7 {
    float S[2][8][4];    // [row][g][c]   (float)sumi[l(g)]
    float PM[2][4][4];   // [row][t][c]   (float)prod (min term; Q4_K: t = 0..3; Q5_K: the block total, replicated)
    float D[2][4];       // [row][c]      y.d * fp16(d)
    float DM[2][4];      // [row][c]     -y.d * fp16(dmin)
};
constexpr int kV7MaxUnits = 32;   // units one wave may own in a launch (its results wait in LDS for the epilogue pass)
template <int MAXK> struct SmemV7 {
    ActLds7<MAXK> L;
    WaveBuf7 WB[16][2];
    float RES[16][2 * kV7MaxUnits];   // [wave][2 * unit + row]
};

// All units of one wave: items first, first + stride, ... < end of the launch's concatenated unit list.
// base / g0: first record and first item of the wave's type group.
template <int TYPE, int MAXK, class Pro>
DEV void v7_run(const MatvecArgs& a, SmemV7<MAXK>& SM, const uint8_t* base, int g0, int first, int stride, int end, int lane, int wv,
                const LaneGeom& G, Pro pro) {
    constexpr bool mins = TYPE != GT_Q6_K;
    constexpr uint32_t REC = rec_bytes<TYPE>();
    const int nb = a.K >> 8, spu = (nb + 3) >> 2;
    const int cb = G.r & 3, row = G.r >> 2;
    const int q8c = 64 * cb + 8 * (cb >> 1);   // q8w_of(4 * s + cb) - 264 * s
    const size_t unit_bytes = (size_t)spu * REC;
    const bool trace = (a.dbg & 32) && blockIdx.x == 0 && lane == 0;
    unsigned long long* tr = (unsigned long long*)a.dbg_sink + 16 * wv;
    const int nu = first < end ? (end - first + stride - 1) / stride : 0;   // units of this wave (host: <= kV7MaxUnits)
    // ---- prefetch cursor: the step whose record is requested next (stays on the last record once everything is requested) ----
    int pf_it = first, pf_s = 0, pf_left = nu * spu;
    const uint8_t* pf_ptr = base + (size_t)((nu > 0 ? first : g0) - g0) * unit_bytes;
    BlkImg<TYPE> ring[4];
    auto issue = [&](BlkImg<TYPE>& slot) __attribute__((always_inline)) {
        slot = img_load<TYPE>(pf_ptr, G);   // unconditional, same instruction count every step (see the header)
        if (pf_left > 1) {
            --pf_left;
            pf_ptr += REC;
            if (++pf_s == spu) {
                pf_s = 0;
                pf_it += stride;
                pf_ptr = base + (size_t)(pf_it - g0) * unit_bytes;
            }
        }
    };
    // Records requested before the prologue: about 32 KB per CU over the 16 waves — what the CU's memory pipeline keeps in flight.
    // More does not start the stream earlier; it only blocks the waves in the issue of loads the pipeline cannot take yet, and
    // with them the prologue's barriers (measured: with four records per wave the last wave reached the first barrier 2.5 us
    // after the kernel started).
    constexpr int PRE = sizeof(BlkImg<TYPE>) > 40 ? 1 : 2;
#pragma unroll
    for (int k = 0; k < PRE; ++k) issue(ring[k]);
    // trace stamps are kept in registers until the loop is over (a store before it would be a pending write at its entry)
    const unsigned long long t1 = trace ? clock64_dev() : 0ull;
    pro();
    const unsigned long long t2 = trace ? clock64_dev() : 0ull;
#pragma unroll
    for (int k = PRE; k < 4; ++k) issue(ring[k]);
    // ---- epilogue operands of lane l = (unit l >> 1, row l & 1), requested before the loop (its body must not contain a load
    //      besides the ring's); unconditional loads: operands a lane does not need are read from the activation vector ----
    const int p1 = a.njobs > 1 ? a.job[1].pair0 : 0x7fffffff, p2 = a.njobs > 2 ? a.job[2].pair0 : 0x7fffffff;
    const int e_it = first + (lane >> 1) * stride;
    const bool e_valid = (lane >> 1) < nu;
    const int e_j = e_it >= p2 ? 2 : (e_it >= p1 ? 1 : 0);
    const int e_p0 = e_j == 2 ? p2 : (e_j == 1 ? p1 : 0);
    const int e_M = e_j == 2 ? a.job[2].w.M : (e_j == 1 ? a.job[1].w.M : a.job[0].w.M);
    const int e_epi = e_j == 2 ? a.job[2].epi : (e_j == 1 ? a.job[1].epi : a.job[0].epi);
    const int e_u = e_it - e_p0;
    const int e_r = a.gateup ? e_u : 2 * e_u + (lane & 1);   // output row of this lane
    const bool e_own = e_valid && e_r < e_M;
    const bool need_res = e_own && (e_epi == EPI_ADD || e_epi == EPI_ADD2), need_res2 = e_own && e_epi == EPI_ADD2;
    const bool need_rope = e_own && (e_epi == EPI_ROPE_Q || e_epi == EPI_ROPE_K);
    // the cursor position: only launches that write the KV cache / rotate (QKV) read it; after the barriers, so that this
    // scalar-cache round trip holds up no other wave
    bool need_pos = false;
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) need_pos = need_pos || (jj < a.njobs && (a.job[jj].epi == EPI_ROPE_Q || a.job[jj].epi == EPI_ROPE_K || a.job[jj].epi == EPI_V));
    const int pos = (need_pos && a.pos) ? sload_i32(a.pos) : 0;
    const float e_res = (need_res ? a.res : a.x)[need_res ? e_r : 0];
    const float e_res2 = (need_res2 ? a.res2 : a.x)[need_res2 ? e_r : 0];
    const float2 e_cs = *(const float2*)((need_rope ? a.rope_cs : a.x) +
                                         (need_rope ? ((size_t)pos * (a.head_dim >> 1) + ((e_r % a.head_dim) >> 1)) * 2 : 0));
    // ---- consume: one flat sequence of nu * spu steps ----
    const int total = nu * spu;
    int s = 0, ui = 0, par = 0;
    float acc = 0.0f, accm = 0.0f;
    // Always whole iterations of four steps (no exit in the middle: the ring then keeps its registers from iteration to
    // iteration — with a `break` hipcc shuffles the slots at the loop head, which reads all four and so waits for all four).
    // The up to three surplus steps at the very end re-process the last record; their results are dropped (ui >= nu).
    for (int st = 0; st < total; st += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // block 4 * s + cb of the row; q8w_of(b) = 264 * s + q8c.  The padding slots of a row's last record (b >= nb) hold zero
            // blocks and read activation words past the image (inside the LDS arrays): their operands are never used (nv below).
            float sv, dv, mv, pv;
            img_to_regs<TYPE>(ring[k], 4 * s + cb, 264 * s + q8c, SM.L, G, sv, dv, mv, pv);
            reg_fence(sv, dv, mv, pv);   // every use of the record is over before its registers are given to the next load
            issue(ring[k]);
            WaveBuf7& W = SM.WB[wv][par];
            par ^= 1;
            W.S[row][G.g][cb] = sv;
            if constexpr (mins) {
                if (G.h == 0) W.PM[row][G.c][cb] = pv;
            }
            if (G.g == 0) {
                W.D[row][cb] = dv;
                if constexpr (mins) W.DM[row][cb] = mv;
            }
            wave_lds_sync();
            const float4 s4 = *(const float4*)W.S[row][G.g];
            const float4 d4 = *(const float4*)W.D[row];
            float4 p4 = float4{0.f, 0.f, 0.f, 0.f}, m4 = float4{0.f, 0.f, 0.f, 0.f};
            if constexpr (mins) {
                p4 = *(const float4*)W.PM[row][G.c];
                m4 = *(const float4*)W.DM[row];
            }
            const int nv = nb - 4 * s;   // blocks of this step that exist (wave-uniform)
            acc = fmaf(d4.x, s4.x, acc);
            if constexpr (mins) accm = fmaf(m4.x, p4.x, accm);   // lanes with h == 1 carry garbage here; never read
            if (nv > 3) {   // the common case as one scalar branch (three selects per chain otherwise)
                acc = fmaf(d4.y, s4.y, acc); acc = fmaf(d4.z, s4.z, acc); acc = fmaf(d4.w, s4.w, acc);
                if constexpr (mins) { accm = fmaf(m4.y, p4.y, accm); accm = fmaf(m4.z, p4.z, accm); accm = fmaf(m4.w, p4.w, accm); }
            } else {
                if (nv > 1) { acc = fmaf(d4.y, s4.y, acc); if constexpr (mins) accm = fmaf(m4.y, p4.y, accm); }
                if (nv > 2) { acc = fmaf(d4.z, s4.z, acc); if constexpr (mins) accm = fmaf(m4.z, p4.z, accm); }
            }
            if (s + 1 < spu) { ++s; continue; }
            // ---- unit end: the reference's reduction tree; the row results wait in LDS for the epilogue pass ----
            float res = hsum8_exact_dpp(acc);
            if constexpr (mins) {
                if constexpr (TYPE == GT_Q4_K) {
                    const float wsum = accm + lane_xor4(accm);
                    accm = wsum + lane_xor2(wsum);
                }
                accm = __shfl(accm, lane & ~7);
                res = res + accm;
            }
            if ((lane & 31) == 0 && ui < nu) SM.RES[wv][2 * ui + row] = res;
            ++ui;
            acc = 0.0f; accm = 0.0f;
            s = 0;
        }
    }
    if (trace) { tr[1] = t1; tr[2] = t2; tr[3] = clock64_dev(); }
    if (nu == 0) return;
    // ---- epilogue pass: lane l finishes row (l & 1) of unit l >> 1 ----
    wave_lds_sync();
    const float res = SM.RES[wv][lane];
    const float other = lane_xor1(res);   // the unit's other row (RoPE partner / up projection)
    if (a.gateup) {   // fused matrix: row 0 of the pair = gate row u, row 1 = up row u
        if (e_own && (lane & 1) == 0) a.out[e_r] = f16_bits_to_f32(a.silu_tab[f32_to_f16_bits(res)]) * other;
        return;
    }
    if (!e_own) return;
    if (e_epi == EPI_ADD) {
        a.out[e_r] = res + e_res;
    } else if (e_epi == EPI_STORE) {
        a.out[e_r] = res;
    } else if (e_epi == EPI_V) {
        a.vcache[(size_t)e_r * a.v_stride + pos] = f32_to_f16_bits(res);
    } else if (e_epi == EPI_GELU) {
        a.out[e_r] = f16_bits_to_f32(a.gelu_tab[f32_to_f16_bits(res)]);
    } else if (e_epi == EPI_ADD2) {
        a.out[e_r] = (res + e_res) + e_res2;
    } else {   // EPI_ROPE_Q / EPI_ROPE_K: the pair (2u, 2u + 1) is one rotation (reference ggml.c:12536-12537, fma forms of the build)
        const float o = (e_r & 1) ? fmaf(res, e_cs.x, other * e_cs.y) : fmaf(res, e_cs.x, -(other * e_cs.y));
        if (e_epi == EPI_ROPE_Q) a.q_f16[e_r] = f32_to_f16_bits(o);
        else a.kcache[kcache_off(pos, e_r, a.head_dim, a.n_ctx)] = f32_to_f16_bits(o);
    }
}

// TA / TB: weight types of the two job groups (TB == 0: one group).  Dynamic LDS: sizeof(SmemV7<MAXK>).
template <int MAXK, int TA, int TB, bool LN, bool EMB = false>
__global__ void __launch_bounds__(1024) matvec_v7_kernel(const MatvecArgs a) {
    CT_DYN_SMEM(smem_raw);
    SmemV7<MAXK>& SM = *reinterpret_cast<SmemV7<MAXK>*>(smem_raw);
    ProRegs7<MAXK, TB == 0> P;
    pro7_load<MAXK, TB == 0>(P, a.x, a.norm_w, a.K, a.pro);
    const int lane = lane_id();
    const int wv = uniform_int(wave_id());
    const LaneGeom G = lane_geom(lane);
    const bool trace = (a.dbg & 32) && blockIdx.x == 0 && lane == 0;
    unsigned long long* tr = (unsigned long long*)a.dbg_sink + 16 * wv;
    const unsigned long long t0 = trace ? clock64_dev() : 0ull;
    auto pro = [&]() __attribute__((always_inline)) {
        pro7_finish<MAXK, LN, EMB, TB == 0>(SM.L, P, a.norm_w, a.norm_b, a.K, a.pro, a.eps, a.emb_out);
    };
    const int grid = (int)gridDim.x, bx = (int)blockIdx.x;
    if constexpr (TB != 0) {
        const int nwA = a.nwA;
        if (wv < nwA) v7_run<TA, MAXK>(a, SM, a.baseA, 0, bx + grid * wv, grid * nwA, a.n_groupA, lane, wv, G, pro);
        else v7_run<TB, MAXK>(a, SM, a.baseB, a.n_groupA, a.n_groupA + bx + grid * (wv - nwA), grid * (16 - nwA), a.n_pairs, lane, wv, G, pro);
    } else {
        v7_run<TA, MAXK>(a, SM, a.baseA, 0, bx + grid * wv, grid * 16, a.n_pairs, lane, wv, G, pro);
    }
    if (trace) { tr[0] = t0; tr[6] = clock64_dev(); }
}
