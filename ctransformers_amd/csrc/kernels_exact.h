// Bit-exact mat-vec: reproduces the f32 accumulation ORDER of the reference's AVX2 kernels, not just their
// quantization points, so logits come out bit-identical to the reference CPU build and greedy decoding can never
// drift.  (Without this, a 1e-7 summation-order difference occasionally flips one int8 rounding in the next Q8_K
// activation quantization — ~10 flips per 7B token — and the logits wander at the 1e-4 level.)
//
// What the reference build computes per weight row (k_quants.c:2651-2720 Q4_K, :3183-3262 Q5_K, :3800-3872 Q6_K; all
// use 8 f32 lanes `acc` updated ONCE PER 256-BLOCK, in block order):
//     sumi[l] = sum over the block's 32-element sub-vectors s of  scale_s(l) * dot4(w_s[4l..4l+3], q8_s[4l..4l+3])   (int32)
//     acc[l]  = fma(y.d * fp16(x.d), (float)sumi[l], acc[l])                          l = 0..7
//   Q4_K min term: prod[t] = m[2t]*q8s[2t] + m[2t+1]*q8s[2t+1] (q8s[j] = bsums[2j]+bsums[2j+1]);
//                  acc_m[t] = fma(-y.d*fp16(x.dmin), (float)prod[t], acc_m[t])        t = 0..3
//   Q5_K min term: summs = fma(dmin, (float)(prod[0]+..+prod[3]), summs)              (scalar, gcc contracts it)
//   result = hsum_float_8(acc) [+ (acc_m0+acc_m2)+(acc_m1+acc_m3) | + summs],
//   hsum_float_8(x) = ((x0+x4)+(x2+x6)) + ((x1+x5)+(x3+x7))                           (k_quants.c:90-97)
//
// Mapping onto a wavefront: 8 lanes per weight row (8 rows per wave).  Lane g of a row owns the 16-byte unit g of each
// 128-byte nibble block, i.e. bytes 4k..4k+3 (k = 0..3) of AVX half h of two sub-vectors; the four lanes with the same
// h together cover all sub-vectors.  A 3-shuffle transpose-reduce over those four lanes leaves lane g holding the
// block's complete integer sumi[l(g)], l(g) = 4*(g&1) + 2*((g>>1)&1) + (g>>2), and the lane then performs exactly
// the reference's per-block fma on its own private accumulator.  Integer sums are exact in any order; every f32
// operation is the reference's, in the reference's order.
#pragma once
#include "kernels.h"

template <int MAXK> struct ActLdsX {
    int q8[MAXK / 4];        // int8 quants, 4 per word
    float yd[MAXK / 256];    // Q8_K block scale d
    int bsums[MAXK / 16];    // sums of 16 quants
    int sb[MAXK / 32];       // sums of 32 quants (= q8s[j] of the reference)
    double red[16];
};

// Prologue: (RMSNorm ->) Q8_K into LDS, values held in registers between the two phases (one global read of x).
// Reference k_quants.c:1191-1226 with `iscale*x[j] + 12582912.f` fused into one fma as the reference build does.
// ------------------------------------------------------------------------------------------------------------------
// Bit-exact decode attention.  The reference computes both attention mat-muls with ggml_vec_dot_f16 (ggml.c:2392-2425,
// AVX: 4 accumulators x 8 f32 lanes, 32 elements per step, fma; reduce macro ggml.c:1964-1982):
//     s_j[l] accumulates elements e = 32*step + 8*j + l, in step order          (j = 0..3, l = 0..7)
//     S[l]   = (s_0[l] + s_2[l]) + (s_1[l] + s_3[l])
//     t0[m]  = S[m] + S[m+4]  (m = 0..3);   res = (t0[0] + t0[1]) + (t0[2] + t0[3])
//     leftovers (n % 32): sumf = (double)res; sumf += (double)(x[i]*y[i]) sequentially;  result = (float)sumf
// A quad of lanes (j = lane&3) owns the four accumulator vectors: lane j reads the 16-byte chunks j, j+4, j+8, ... of
// the row (8 halves = its 8 l-lanes for one step) and runs 8 independent fma chains; a 6-shuffle transpose-reduce
// reproduces the reduction tree.
// ------------------------------------------------------------------------------------------------------------------
DEV void unpack8_f16(const u32x4 v, float* f) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[2 * k] = f16_bits_to_f32((uint16_t)(v[k] & 0xFFFF));
        f[2 * k + 1] = f16_bits_to_f32((uint16_t)(v[k] >> 16));
    }
}

// in: acc[8] = s_j[0..7] of quad lane j; out (all 4 lanes): res of the reference's reduce.
// Two DPP quad exchanges per accumulator give every lane S[l] = (s_0[l] + s_2[l]) + (s_1[l] + s_3[l]) (the operands of each
// addition are the reference's, in commuted order in half of the lanes); the rest of the tree is in-lane.  23 VALU
// operations and no LDS traffic — the bpermute/select form this replaces cost more than the dot product itself.
DEV float f16dot_reduce_exact(const float* acc, int j) {
    (void)j;
    float S[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const float x = acc[l] + lane_xor2(acc[l]);
        S[l] = x + lane_xor1(x);
    }
    const float t0 = S[0] + S[4], t1 = S[1] + S[5], t2 = S[2] + S[6], t3 = S[3] + S[7];
    return (t0 + t1) + (t2 + t3);
}

// ggml_vec_dot_f16's scalar tail (ggml.c:2420-2423) over nl (1..31) leftover positions of a V*P dot: sumf += (double)(v[i] * p[i]) in order.
// `p` = the 32 probabilities behind the fma part (16-byte aligned; all 32 lie inside the LDS row, which is rounded up to 64 positions).  They are read
// UNCONDITIONALLY — eight 16-byte LDS reads, one wait; with the read inside `if (i < nl)` every leftover position cost an LDS round trip behind a branch,
// ~100 cycles each and ~1 500 cycles of an average decode attention launch — and so are the products; only the adds are conditional (adding +0.0 could
// turn a -0.0 sum into +0.0).  nl is wave-uniform.
DEV double f16_tail32(double sumf, const u32x4* v, const float* p, int nl) {
    u32x4 pb[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) pb[c] = *(const u32x4*)(p + 4 * c);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (8 * c < nl) {   // (a group of eight leftover positions that does not exist costs no double-precision adds)
            float lf[8], pr[8];
            unpack8_f16(v[c], lf);
#pragma unroll
            for (int i = 0; i < 8; ++i) pr[i] = lf[i] * bits_to_f32(pb[2 * c + (i >> 2)][i & 3]);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (8 * c + i < nl) sumf += (double)pr[i];
        }
    }
    return sumf;
}

struct AttnArgsX {
    const uint16_t* q_f16;
    const uint16_t* kcache;  // layer base [n_head_kv][n_ctx][head_dim] (kcache_off)
    const uint16_t* vcache;  // layer base [n_embd_gqa][v_stride]
    float* scores;           // [n_head][n_ctx]
    float* out;
    const int* pos;          // &cursor[1] (position of token 0 of this launch); the kernel also reads cursor[0] = step through it
    const int* n_total;      // &cursor[2] = n_past + n of the eval; cursor[3] = reference batch size inside the eval (0: one batch):
                             // together they give the end of the reference batch a token belongs to (length of its V*P dot)
    const uint16_t* exp_tab;
    int n_head, n_head_kv, head_dim, n_embd_gqa, n_ctx, v_stride;
    float kq_scale;
    unsigned long long* trace;   // measurement only: s_memtime stamps of workgroup (0,0)
    int q_stride, out_stride;    // prompt chunks (kernels_pf.h): token blockIdx.z has position *pos + z, query row z, output row z
    const float* alibi;          // MPT: per-head slope m_k; the scaled score of key position i becomes fma(m_k, i, score) (ggml.c:12193-12254)
    uint32_t* xs;                // attn_decode9_kernel<.., SHARE>: the heads' shared score rows as tagged granules [n_head][n_ctx] x {score, tag}
    const unsigned* epoch;       // token epoch behind the cursor's token ids (kernels.h:advance_state_kernel); with `layer` the granules' tag
    int* err;                    // pinned host word a timed-out gather raises
    int layer;
    int vt_off, vt_row;          // attn_decode9_kernel<.., VLDS>: byte offset of the V tile in dynamic LDS, halves per channel row of it
};

// Fused form of the two kernels above (one launch per layer instead of two): grid (n_head, head_dim/64), 1024 threads.
// Every workgroup recomputes the (cheap) score row of its head into LDS — 256 positions per pass, a quad per position —
// then runs the softmax and its 64 channels of V*P exactly as attn_softmax_pv_exact_kernel does.  The double-precision
// exp sum is order-free here: the addends are fp16 values in (0, 1] (multiples of 2^-24), so any summation order of up
// to 8192 of them is exact in binary64.
// ALLCH (prompt chunks): one workgroup per (head, token) runs ALL head_dim channels of V*P, 64 at a time, instead of one workgroup
// per 64 channels each recomputing the score row — half the workgroups for the latency-bound chunk launch.
// ALIBI (MPT): ggml_alibi between the scale and the mask — the reference build contracts `i * m_k + src` into one fma.
// Head sizes that are not whole 32-element steps (MPT-30B: 112 = 3 steps + 16): the K.Q dot takes ggml_vec_dot_f16's scalar tail too —
// after the tree reduce, sumf += (double)(k[i] * q[i]) for i = HD & ~31 .. HD - 1 in order (every lane of the quad runs it on the
// same values) — and the V*P part walks the head's channels 64 at a time with the last group partly idle (ALLCH instantiations only).
// GPROB (contexts above kMaxCtxFused, whose probability row does not fit LDS): the row lives in global memory, a.scores[head][n_ctx]
// (one workgroup per head: ALLCH), written and read by this workgroup only, between workgroup barriers.  The reference takes any
// context length (llama.cc:90-92); this form is the slow path that keeps such handles loadable.
template <int NT, int HD, bool ALLCH = false, bool ALIBI = false, bool GPROB = false>
__global__ void __launch_bounds__(NT) attn_fused_exact_kernel(const AttnArgsX a) {
    kernarg_touch<sizeof(AttnArgsX)>();   // (gpu.h)
    static_assert(!GPROB || ALLCH, "the global probability row has one writer: all channels of a head in one workgroup");
    constexpr int NWV = NT / 64, NQ = NT / 4;   // NQ quads: positions per pass
    constexpr int NC = HD / 32;                 // 16-byte chunks of a K row per quad lane
    constexpr int TAIL = HD - 32 * NC;          // elements of the scalar tail of the K.Q dot (0 or 16)
    static_assert(TAIL == 0 || (TAIL == 16 && ALLCH), "head_dim: a multiple of 32, or of 16 with all channels in one workgroup");
    constexpr int PB = 4;                       // positions per quad whose K rows are in flight together
    constexpr int VB = 8;                       // V chunks (32 positions each) in flight together
    CT_DYN_SMEM(smem_raw);   // the score / probability row of this token: n_ctx floats (dynamic: 2 KB at the default context, 128 KB at 32768)
    __shared__ double red[NWV];
    __shared__ float redf[NWV];
    const int h = (int)blockIdx.x;
    float* prob = GPROB ? a.scores + ((size_t)blockIdx.z * a.n_head + h) * a.n_ctx : reinterpret_cast<float*>(smem_raw);
    const bool trace = a.trace && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0;
    unsigned long long* tr = a.trace + 16 * (threadIdx.x >> 6);
    if (trace) tr[0] = clock64_dev();
    const int tok = (int)blockIdx.z;
    const int n_kv = *a.pos + tok + 1;
    // The length of the value dot product is that of the reference batch this token belongs to (its fma / leftover split
    // depends on it).  The cursor {step, pos, n_past + n, batch} describes an eval of n tokens the reference would have run in
    // batches of `batch` (0: one batch): this token is number step + tok of the eval, its batch ends at the next multiple.
    int n_tot = *a.n_total;
    {
        const int bs = a.n_total[1];
        if (bs > 0) {
            const int idx = a.pos[-1] + tok, base = *a.pos - a.pos[-1];
            const int end = (idx / bs + 1) * bs, n_eval = n_tot - base;
            n_tot = base + (end < n_eval ? end : n_eval);
        }
    }
    const int np = n_tot & ~31;
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = wave_id(), j = tid & 3;
    const int hk = h / (a.n_head / a.n_head_kv);
    const float slope = ALIBI ? a.alibi[h] : 0.0f;
    if (trace) { tr[1] = clock64_dev(); tr[7] = (unsigned long long)n_kv; }
    const uint16_t* qrow = a.q_f16 + (size_t)tok * a.q_stride + (size_t)h * HD;
    const uint16_t* kbase = a.kcache + (size_t)hk * a.n_ctx * HD + 8 * j;
    const bool pv_thread = tid < 256;           // 64 channels x 4 lanes run the V*P part
    int d = (int)blockIdx.y * 64 + ((tid & 255) >> 2);
    const uint16_t* vrow = a.vcache + ((size_t)hk * HD + d) * a.v_stride;
    // The V rows do not depend on the probabilities: their first VB chunks are requested now, so that their latency
    // overlaps the score and softmax phases instead of following them.
    u32x4 vv[VB];
    float qf[NC][8];   // this lane's slices of the query, converted once
#pragma unroll
    for (int c = 0; c < NC; ++c) unpack8_f16(ld16(qrow + 32 * c + 8 * j), qf[c]);
    float qt[TAIL ? TAIL : 1];   // the tail elements of the query
    if constexpr (TAIL > 0) {
#pragma unroll
        for (int c = 0; c < TAIL / 8; ++c) unpack8_f16(ld16(qrow + 32 * NC + 8 * c), qt + 8 * c);
    }
    float mx = -INFINITY;
    for (int base = 0; base < n_kv; base += NQ * PB) {
        u32x4 kv[PB][NC];
        u32x4 kt[PB][TAIL ? TAIL / 8 : 1];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int p = base + u * NQ + (tid >> 2);
            const uint16_t* krow = kbase + (size_t)p * HD;
#pragma unroll
            for (int c = 0; c < NC; ++c)   // no clamping: hundreds of idle quads re-reading one row serialise in the L1
                kv[u][c] = (p < n_kv) ? ld16(krow + 32 * c) : u32x4{0u, 0u, 0u, 0u};
            if constexpr (TAIL > 0) {
#pragma unroll
                for (int c = 0; c < TAIL / 8; ++c)   // kbase carries the quad lane's 8 * j: the tail is read from the row's start
                    kt[u][c] = (p < n_kv) ? ld16(krow - 8 * j + 32 * NC + 8 * c) : u32x4{0u, 0u, 0u, 0u};
            }
        }
        if (base == 0) {   // after the K requests (the critical path), before anything waits on them
#pragma unroll
            for (int u = 0; u < VB; ++u) vv[u] = (pv_thread && 32 * u < np) ? ld16(vrow + 32 * u + 8 * j) : u32x4{0u, 0u, 0u, 0u};   // (HD >= 64: the first 64 channels exist)
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int p = base + u * NQ + (tid >> 2);
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float kf[8];
                unpack8_f16(kv[u][c], kf);
#pragma unroll
                for (int l = 0; l < 8; ++l) acc[l] = fmaf(kf[l], qf[c][l], acc[l]);
            }
            float dot = f16dot_reduce_exact(acc, j);
            if constexpr (TAIL > 0) {   // ggml.c:2420-2423
                double sumf = (double)dot;
#pragma unroll
                for (int c = 0; c < TAIL / 8; ++c) {
                    float kf[8];
                    unpack8_f16(kt[u][c], kf);
#pragma unroll
                    for (int l = 0; l < 8; ++l) sumf += (double)(kf[l] * qt[8 * c + l]);
                }
                dot = (float)sumf;
            }
            float sc = dot * a.kq_scale;
            if (ALIBI) sc = fmaf((float)p, slope, sc);
            if (p < n_kv) {
                mx = fmaxf(mx, sc);
                if (j == 0) prob[p] = sc;
            }
        }
    }
    if (trace) tr[2] = clock64_dev();
    mx = wave_max(mx);
    if (lane == 0) redf[wv] = mx;
    __syncthreads();
    mx = redf[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) mx = fmaxf(mx, redf[w]);
    if (trace) tr[3] = clock64_dev();
    double sum = 0.0;
    constexpr int SB = 4;   // exp-table lookups of 4 elements per thread in flight together
    for (int i0 = 0; i0 < n_kv; i0 += NT * SB) {
        uint16_t e16[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) { const int i = i0 + u * NT + tid; e16[u] = i < n_kv ? a.exp_tab[f32_to_f16_bits(prob[i] - mx)] : (uint16_t)0; }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int i = i0 + u * NT + tid;
            if (i < n_kv) { const float e = f16_bits_to_f32(e16[u]); prob[i] = e; sum += (double)e; }
        }
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    double tot = red[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) tot += red[w];
    const float inv = (float)(1.0 / tot);
    for (int i = tid; i < n_kv; i += NT) prob[i] = f16_bits_to_f32(f32_to_f16_bits(prob[i] * inv));
    for (int i = n_kv + tid; i < np; i += NT) prob[i] = 0.0f;  // masked columns of this batch
    __syncthreads();
    if (trace) tr[4] = clock64_dev();
    if (!pv_thread) return;
#pragma unroll 1
    for (int half = 0; half < (ALLCH ? (HD + 63) / 64 : 1); ++half) {
    if (half > 0) {   // the next 64 channels of this head
        d += 64;
        if (HD % 64 != 0 && d >= HD) return;   // quad-uniform (ALLCH: blockIdx.y == 0): the last group of a head of 112 has 48 channels
        vrow += (size_t)64 * a.v_stride;
#pragma unroll
        for (int u = 0; u < VB; ++u) vv[u] = (32 * u < np) ? ld16(vrow + 32 * u + 8 * j) : u32x4{0u, 0u, 0u, 0u};
    }
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i0 = 0; i0 < np; i0 += 32 * VB) {
        if (i0 > 0) {
#pragma unroll
            for (int u = 0; u < VB; ++u)
                if (i0 + 32 * u < np) vv[u] = ld16(vrow + i0 + 32 * u + 8 * j);
        }
#pragma unroll
        for (int u = 0; u < VB; ++u) {
            if (i0 + 32 * u < np) {
                float vf[8];
                unpack8_f16(vv[u], vf);
                const float* pr = &prob[i0 + 32 * u + 8 * j];
#pragma unroll
                for (int l = 0; l < 8; ++l) acc[l] = fmaf(vf[l], pr[l], acc[l]);
            }
        }
    }
    const float res = f16dot_reduce_exact(acc, j);
    double sumf = (double)res;
    if (trace) tr[5] = clock64_dev();
    // Leftover positions np .. n_kv - 1 (fewer than 32; ggml_vec_dot_f16's scalar tail: sumf += (double)(x[i] * y[i]) in order).  Their V
    // values are fetched with four 16-byte loads issued together — one memory latency instead of one per position (the dependent
    // 2-byte loads of the plain loop cost ~280 cycles each, 2 us of a 6.5 us launch at the average of 15 leftovers) — and the
    // double-precision chain then runs from registers.  The row is padded to a multiple of 32 positions (v_stride).
    const int nl = n_kv - np;   // wave-uniform
    if (nl > 0) {
        u32x4 lv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) lv[c] = ld16(vrow + np + 8 * c);
        float lf[32];
#pragma unroll
        for (int c = 0; c < 4; ++c) unpack8_f16(lv[c], lf + 8 * c);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (i < nl) sumf += (double)(lf[i] * prob[np + i]);
        }
    }
    if (j == 0) a.out[(size_t)tok * a.out_stride + (size_t)h * HD + d] = (float)sumf;
    }
    if (trace) tr[6] = clock64_dev();
}

// Prompt chunks: ONE WAVE per (head, token), WPB of them per workgroup, nothing shared and no workgroup barrier.  The arithmetic per
// position / channel is attn_fused_exact_kernel's (a quad of lanes per position with the four AVX accumulators of ggml_vec_dot_f16,
// the fp16 exp table, the order-free double sum, a quad per channel for V*P with the 32-wide fma part and the double-precision
// leftovers) — only the thread mapping changes: 16 positions / 16 channels per pass instead of 64.  The fused kernel is a serial
// latency chain of ~6 us per (head, token) that keeps a whole 256-thread workgroup (and its LDS row) for one token: three workgroups per
// CU, 4096 of them for a 128-token chunk of a 7B = 36 us per layer.  A wave per token keeps 12 tokens in flight per CU.
// Dynamic LDS: WPB probability rows of row_floats each (engine.cc picks WPB so that they fit).
template <int HD, int WPB>
__global__ void __launch_bounds__(WPB * 64) attn_chunk_wave_kernel(const AttnArgsX a, int n_tok, int row_floats) {
    kernarg_touch<sizeof(AttnArgsX)>();   // (gpu.h)
    constexpr int NC = HD / 32;                 // 16-byte chunks of a K row per quad lane
    constexpr int PB = 4;                       // positions per quad whose K rows are in flight together
    constexpr int VB = 4;                       // V chunks (32 positions each) in flight together
    CT_DYN_SMEM(smem_raw);
    const int lane = lane_id(), wv = uniform_int(wave_id()), j = lane & 3, quad = lane >> 2;
    const int h = (int)blockIdx.x, tok = (int)blockIdx.y * WPB + wv;
    if (tok >= n_tok) return;                   // whole wave; no workgroup barrier anywhere below
    float* prob = reinterpret_cast<float*>(smem_raw) + (size_t)wv * row_floats;
    const int n_kv = *a.pos + tok + 1;
    int n_tot = *a.n_total;                     // end of the reference batch this token belongs to (see attn_fused_exact_kernel)
    {
        const int bs = a.n_total[1];
        if (bs > 0) {
            const int idx = a.pos[-1] + tok, base = *a.pos - a.pos[-1];
            const int end = (idx / bs + 1) * bs, n_eval = n_tot - base;
            n_tot = base + (end < n_eval ? end : n_eval);
        }
    }
    const int np = n_tot & ~31;
    const int hk = h / (a.n_head / a.n_head_kv);
    const uint16_t* qrow = a.q_f16 + (size_t)tok * a.q_stride + (size_t)h * HD;
    const uint16_t* kbase = a.kcache + (size_t)hk * a.n_ctx * HD + 8 * j;
    float qf[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c) unpack8_f16(ld16(qrow + 32 * c + 8 * j), qf[c]);
    float mx = -INFINITY;
    for (int base = 0; base < n_kv; base += 16 * PB) {
        u32x4 kv[PB][NC];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int p = base + u * 16 + quad;
            const uint16_t* krow = kbase + (size_t)p * HD;
#pragma unroll
            for (int c = 0; c < NC; ++c) kv[u][c] = (p < n_kv) ? ld16(krow + 32 * c) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int p = base + u * 16 + quad;
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float kf[8];
                unpack8_f16(kv[u][c], kf);
#pragma unroll
                for (int l = 0; l < 8; ++l) acc[l] = fmaf(kf[l], qf[c][l], acc[l]);
            }
            const float sc = f16dot_reduce_exact(acc, j) * a.kq_scale;
            if (p < n_kv) {
                mx = fmaxf(mx, sc);
                if (j == 0) prob[p] = sc;
            }
        }
    }
    mx = wave_max(mx);
    wave_lds_sync();
    double sum = 0.0;
    for (int i0 = 0; i0 < n_kv; i0 += 64) {
        const int i = i0 + lane;
        if (i < n_kv) { const float e = f16_bits_to_f32(a.exp_tab[f32_to_f16_bits(prob[i] - mx)]); prob[i] = e; sum += (double)e; }
    }
    sum = wave_sum(sum);
    const float inv = (float)(1.0 / sum);
    for (int i = lane; i < n_kv; i += 64) prob[i] = f16_bits_to_f32(f32_to_f16_bits(prob[i] * inv));
    for (int i = n_kv + lane; i < np; i += 64) prob[i] = 0.0f;   // masked columns of this batch
    wave_lds_sync();
    const int nl = n_kv - np;                   // leftover positions (< 32), wave-uniform
    // 16 channels per pass (a quad each), TWO passes together: every batch of V loads is one memory latency on this wave's serial
    // chain, and the chain is all a wave-per-token kernel has — 8 single passes cost 8 latencies (~1.3 us each), pairs cost 4.
#pragma unroll 1
    for (int pass = 0; pass < HD / 16; pass += 2) {
        const uint16_t* vrow[2];
        float acc[2][8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            vrow[t] = a.vcache + ((size_t)hk * HD + (pass + t) * 16 + quad) * a.v_stride;
#pragma unroll
            for (int l = 0; l < 8; ++l) acc[t][l] = 0.0f;
        }
        for (int i0 = 0; i0 < np; i0 += 32 * VB) {
            u32x4 vv[2][VB];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < VB; ++u) vv[t][u] = (i0 + 32 * u < np) ? ld16(vrow[t] + i0 + 32 * u + 8 * j) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < VB; ++u) {
                    if (i0 + 32 * u < np) {
                        float vf[8];
                        unpack8_f16(vv[t][u], vf);
                        const float* pr = &prob[i0 + 32 * u + 8 * j];
#pragma unroll
                        for (int l = 0; l < 8; ++l) acc[t][l] = fmaf(vf[l], pr[l], acc[t][l]);
                    }
                }
        }
        u32x4 lv[2][4];
        if (nl > 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c) lv[t][c] = ld16(vrow[t] + np + 8 * c);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            double sumf = (double)f16dot_reduce_exact(acc[t], j);
            if (nl > 0) {
                float lf[32];
#pragma unroll
                for (int c = 0; c < 4; ++c) unpack8_f16(lv[t][c], lf + 8 * c);
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    if (i < nl) sumf += (double)(lf[i] * prob[np + i]);
                }
            }
            if (j == 0) a.out[(size_t)tok * a.out_stride + (size_t)h * HD + (pass + t) * 16 + quad] = (float)sumf;
        }
    }
}

// Prompt chunks whose positions all lie below 128 (the first chunk of a prompt — every BASELINE prompt): the K rows and V rows of a
// head are brought into LDS ONCE per 16 tokens and the 16 waves of the workgroup (a token each, as in attn_chunk_wave_kernel, same
// arithmetic) read them from there.  Per (head, token) the wave kernel fetches 64 KB of K/V through L2 — 268 MB for a 128-token chunk of
// a 7B, the launch is L2-bandwidth bound at ~20 us; here it is 17 MB and one workgroup per CU.
// LDS: K tile [128 positions][HD halves + 32 pad] | V tile [HD channels][128 positions + 32 pad] | 16 probability rows of 160 floats.
// The 64-byte pads put the four quads of a ds_read_b128 phase on different banks (row strides 320 / 192 bytes: 16 (p mod 4) + 4 j).
template <int HD>
__global__ void __launch_bounds__(1024) attn_chunk_tile_kernel(const AttnArgsX a, int n_tok) {
    kernarg_touch<sizeof(AttnArgsX)>();   // (gpu.h)
    constexpr int NC = HD / 32, PB = 4, VB = 4;
    constexpr int KS = HD * 2 + 64, VS = 256 + 64, ROW = 160, NCH = HD / 8;   // bytes, bytes, floats, 16-byte chunks per K row
    CT_DYN_SMEM(smem_raw);
    unsigned char* Kt = smem_raw;
    unsigned char* Vt = smem_raw + 128 * KS;
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = uniform_int(wave_id()), j = lane & 3, quad = lane >> 2;
    // tokens are dealt to the workgroups round-robin (wave w of workgroup y: token y + w * gridDim.y): causal attention costs a token
    // in proportion to its position, so consecutive tokens per workgroup would leave the last workgroup with 4x the average work
    const int h = (int)blockIdx.x, tok = (int)blockIdx.y + wv * (int)gridDim.y;
    const int hk = h / (a.n_head / a.n_head_kv);
    const bool trace = a.trace && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0;   // measurement only (CT_AMD_PG_TRACE=attn)
    unsigned long long* tr = a.trace + 16 * wv;
    if (trace) tr[0] = clock64_dev();
    const int pos0 = *a.pos;
    const int n_kv_wg = pos0 + n_tok;           // <= 128 (the host selects this kernel only then)
    {
        const uint16_t* kb = a.kcache + (size_t)hk * a.n_ctx * HD;
        for (int i = tid; i < n_kv_wg * NCH; i += 1024) {
            const int p = i / NCH, c = i - p * NCH;
            *(u32x4*)(Kt + p * KS + c * 16) = ld16(kb + (size_t)p * HD + 8 * c);
        }
        const uint16_t* vb = a.vcache + (size_t)hk * HD * a.v_stride;
        for (int i = tid; i < HD * 16; i += 1024) {   // 128 positions of every channel (positions past the batch's end are never used)
            const int ch = i >> 4, c = i & 15;
            *(u32x4*)(Vt + ch * VS + c * 16) = ld16(vb + (size_t)ch * a.v_stride + 8 * c);
        }
    }
    __syncthreads();
    if (tok >= n_tok) return;                   // whole wave, after the only barrier
    if (trace) tr[1] = clock64_dev();
    float* prob = reinterpret_cast<float*>(smem_raw + 128 * KS + HD * VS) + wv * ROW;
    const int n_kv = pos0 + tok + 1;
    int n_tot = *a.n_total;                     // end of the reference batch this token belongs to (see attn_fused_exact_kernel)
    {
        const int bs = a.n_total[1];
        if (bs > 0) {
            const int idx = a.pos[-1] + tok, base = pos0 - a.pos[-1];
            const int end = (idx / bs + 1) * bs, n_eval = n_tot - base;
            n_tot = base + (end < n_eval ? end : n_eval);
        }
    }
    const int np = n_tot & ~31;
    const uint16_t* qrow = a.q_f16 + (size_t)tok * a.q_stride + (size_t)h * HD;
    float qf[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c) unpack8_f16(ld16(qrow + 32 * c + 8 * j), qf[c]);
    float mx = -INFINITY;
    for (int base = 0; base < n_kv; base += 16 * PB) {
        u32x4 kv[PB][NC];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int p = base + u * 16 + quad;
            const int pc = p < 128 ? p : 127;   // rows past n_kv_wg hold stale bytes: read, never used
#pragma unroll
            for (int c = 0; c < NC; ++c) kv[u][c] = *(const u32x4*)(Kt + pc * KS + (32 * c + 8 * j) * 2);
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int p = base + u * 16 + quad;
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float kf[8];
                unpack8_f16(kv[u][c], kf);
#pragma unroll
                for (int l = 0; l < 8; ++l) acc[l] = fmaf(kf[l], qf[c][l], acc[l]);
            }
            const float sc = f16dot_reduce_exact(acc, j) * a.kq_scale;
            if (p < n_kv) {
                mx = fmaxf(mx, sc);
                if (j == 0) prob[p] = sc;
            }
        }
    }
    mx = wave_max(mx);
    wave_lds_sync();
    if (trace) tr[2] = clock64_dev();
    double sum = 0.0;
    for (int i0 = 0; i0 < n_kv; i0 += 64) {
        const int i = i0 + lane;
        if (i < n_kv) { const float e = f16_bits_to_f32(a.exp_tab[f32_to_f16_bits(prob[i] - mx)]); prob[i] = e; sum += (double)e; }
    }
    sum = wave_sum(sum);
    const float inv = (float)(1.0 / sum);
    for (int i = lane; i < n_kv; i += 64) prob[i] = f16_bits_to_f32(f32_to_f16_bits(prob[i] * inv));
    for (int i = n_kv + lane; i < np; i += 64) prob[i] = 0.0f;   // masked columns of this batch
    wave_lds_sync();
    const int nl = n_kv - np;                   // leftover positions (< 32), wave-uniform
    if (trace) tr[3] = clock64_dev();
#pragma unroll 1
    for (int pass = 0; pass < HD / 16; ++pass) {   // 16 channels per pass, a quad each
        const int d = pass * 16 + quad;
        const unsigned char* vrow = Vt + d * VS;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < VB; ++u) {          // np <= 128: at most four steps of 32 positions
            if (32 * u < np) {
                float vf[8];
                unpack8_f16(*(const u32x4*)(vrow + (32 * u + 8 * j) * 2), vf);
                const float* pr = &prob[32 * u + 8 * j];
#pragma unroll
                for (int l = 0; l < 8; ++l) acc[l] = fmaf(vf[l], pr[l], acc[l]);
            }
        }
        double sumf = (double)f16dot_reduce_exact(acc, j);
        if (nl > 0) {
            float lf[32];
#pragma unroll
            for (int c = 0; c < 4; ++c) unpack8_f16(*(const u32x4*)(vrow + (np + 8 * c) * 2), lf + 8 * c);
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if (i < nl) sumf += (double)(lf[i] * prob[np + i]);
            }
        }
        if (j == 0) a.out[(size_t)tok * a.out_stride + (size_t)h * HD + d] = (float)sumf;
    }
    if (trace) { tr[4] = clock64_dev(); tr[5] = (unsigned long long)n_kv; }
}

// Prologue, 16 lanes per 256-block: lane `sub` owns 16 consecutive elements, so the per-block reductions are 4 DPP
// steps inside a row of 16 lanes and all 16 (32) blocks of a round proceed at once.  Same arithmetic as
// prologue_q8k_exact (reference k_quants.c:1191-1226 with the fused fma; RMSNorm ggml.c:10700-10716).
// Two copies on purpose: the LayerNorm form (falcon) needs more live registers; keeping it out of the RMSNorm / plain
// function lets the register allocator treat the two call sites separately (with one merged body every mat-vec
// instantiation started to spill and the Q6_K K=11008 kernel went from 16 to 27 us).
template <int NT, int MAXK>
DEV void prologue_q8k_exact16(ActLdsX<MAXK>& L, const float* __restrict__ x, const float* __restrict__ nw, int K, int pro,
                              float eps) {
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, sub = tid & 15, grp = tid >> 4;
    constexpr int NW = NT / 64, NG = NT / 16;
    constexpr int ROUNDS = (MAXK / 256 + NG - 1) / NG;
    const int nblk = K >> 8;
    // A wave whose four 16-lane rows are all past the last block has nothing to quantize: it must SKIP the arithmetic
    // (wave-uniform branches), not run it predicated off — the prologue is VALU-issue bound (about 250 wave
    // instructions), and for K = 4096 only 4 of the 16 waves (one per SIMD) are live.
    const bool wave_live = uniform_int(wv * 4) < nblk;
    float4 v[ROUNDS][4], nwv[ROUNDS][4];   // the norm weights travel with the activations: requested behind the sums they cost a second memory latency
    double s = 0.0;
    if (wave_live) {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int b = grp + rd * NG;
            if (b < nblk) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[rd][k] = *(const float4*)(x + b * 256 + sub * 16 + k * 4);
                if (pro == PRO_RMSNORM) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) nwv[rd][k] = *(const float4*)(nw + b * 256 + sub * 16 + k * 4);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (pro == PRO_RMSNORM) {
                        s += (double)(v[rd][k].x * v[rd][k].x);
                        s += (double)(v[rd][k].y * v[rd][k].y);
                        s += (double)(v[rd][k].z * v[rd][k].z);
                        s += (double)(v[rd][k].w * v[rd][k].w);
                    }
                }
            }
        }
    }
    float scale = 1.0f;
    if (pro == PRO_RMSNORM) {
        if (wave_live) {
            s = wave_sum_fast(s);
            if (lane == 0) L.red[wv] = s;
        } else if (lane == 0) {
            L.red[wv] = 0.0;
        }
        __syncthreads();
        if (wave_live) {
            // sixteen per-wave partials: lane `sub` of every 16-lane row takes one, the row reduces (a serial loop of dependent LDS reads
            // cost a microsecond here); the sum is regrouped (exact, hence order-free, within the dynamic range kernels_v9.h:pro9_total states)
            static_assert(NW <= 16, "one partial per lane of a 16-lane row");
            double tot = sub < NW ? L.red[sub] : 0.0;
            tot += lane_xor1(tot); tot += lane_xor2(tot); tot += lane_xor4(tot); tot += lane_xor8(tot);
            const float mean = (float)(tot / (double)K);
            scale = 1.0f / sqrtf(mean + eps);
        }
    }
    if (wave_live) {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int b = grp + rd * NG;
            const bool live = b < nblk;            // uniform within a 16-lane row, may differ between rows of a wave
            float t[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float4 q = live ? v[rd][k] : float4{0.f, 0.f, 0.f, 0.f};
                if (live && pro == PRO_RMSNORM) {
                    const float4 w4 = nwv[rd][k];
                    q.x = (q.x * scale) * w4.x;
                    q.y = (q.y * scale) * w4.y;
                    q.z = (q.z * scale) * w4.z;
                    q.w = (q.w * scale) * w4.w;
                }
                t[4 * k] = q.x; t[4 * k + 1] = q.y; t[4 * k + 2] = q.z; t[4 * k + 3] = q.w;
            }
            // largest / smallest signed value instead of a search for the first element of largest magnitude; the search only where a block
            // holds +amax and -amax (kernels_v9.h:pro9_finish has the argument)
            float hi = fmaxf(fmaxf(t[0], t[1]), t[2]), lo = fminf(fminf(t[0], t[1]), t[2]);
#pragma unroll
            for (int e = 3; e < 15; e += 2) { hi = fmaxf(fmaxf(hi, t[e]), t[e + 1]); lo = fminf(fminf(lo, t[e]), t[e + 1]); }
            hi = fmaxf(hi, t[15]); lo = fminf(lo, t[15]);
            hi = fmaxf(hi, lane_xor1(hi)); lo = fminf(lo, lane_xor1(lo));
            hi = fmaxf(hi, lane_xor2(hi)); lo = fminf(lo, lane_xor2(lo));
            hi = fmaxf(hi, lane_xor4(hi)); lo = fminf(lo, lane_xor4(lo));
            hi = fmaxf(hi, lane_xor8(hi)); lo = fminf(lo, lane_xor8(lo));
            const float amax = fmaxf(hi, -lo);
            float maxv = hi == amax ? hi : lo;
            if (__ballot(hi == -lo && amax != 0.0f) != 0ull) {   // the first element (lowest index) attaining amax keeps its sign
                float am = 0.0f;
#pragma unroll
                for (int e = 0; e < 16; ++e) am = fmaxf(am, fabsf(t[e]));
                const unsigned long long hit = __ballot(am == amax);
                const unsigned row_bits = (unsigned)((hit >> (lane & 48)) & 0xFFFFu);
                const int first = (lane & 48) + (__ffsll((unsigned long long)row_bits) - 1);
                float mine = 0.0f;
#pragma unroll
                for (int e = 15; e >= 0; --e) mine = (fabsf(t[e]) == amax) ? t[e] : mine;
                maxv = __shfl(mine, first);
            }
            // nearest_int(iscale * x) = bits(fma(iscale, x, 1.5 * 2^23)) - bits(1.5 * 2^23): the sum stays in the binade of ulp 1, so the quant is
            // its low byte and MIN(127, .) is a float minimum against 1.5 * 2^23 + 127 (the decode prologue's form)
            const bool nz = amax != 0.0f;
            const float iscale = nz ? -128.f / maxv : 0.0f;
            const float d = nz ? 1.0f / iscale : 0.0f;
            int packed[4], s16 = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t by[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) by[e] = f32_to_bits(fminf(fmaf(iscale, t[4 * k + e], 12582912.f), 12583039.f));
                packed[k] = (int)pack_low_bytes(by[0], by[1], by[2], by[3]);
                s16 = sdot4(packed[k], 0x01010101, s16);
            }
            const int s32 = s16 + lane_xor1(s16);
            if (live) {
#pragma unroll
                for (int k = 0; k < 4; ++k) L.q8[b * 64 + sub * 4 + k] = packed[k];
                L.bsums[b * 16 + sub] = s16;
                if ((sub & 1) == 0) L.sb[b * 8 + (sub >> 1)] = s32;
                if (sub == 0) L.yd[b] = d;
            }
        }
    }
    __syncthreads();
}

template <int NT, int MAXK>
DEV void prologue_q8k_exact16_ln(ActLdsX<MAXK>& L, const float* __restrict__ x, const float* __restrict__ nw, int K, int pro,
                              float eps, const float* __restrict__ nbias = nullptr) {
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, sub = tid & 15, grp = tid >> 4;
    constexpr int NW = NT / 64, NG = NT / 16;
    constexpr int ROUNDS = (MAXK / 256 + NG - 1) / NG;
    const int nblk = K >> 8;
    // A wave whose four 16-lane rows are all past the last block has nothing to quantize: it must SKIP the arithmetic
    // (wave-uniform branches), not run it predicated off — the prologue is VALU-issue bound (about 250 wave
    // instructions), and for K = 4096 only 4 of the 16 waves (one per SIMD) are live.
    const bool wave_live = uniform_int(wv * 4) < nblk;
    float4 v[ROUNDS][4];
    double s = 0.0;
    if (wave_live) {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int b = grp + rd * NG;
            if (b < nblk) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[rd][k] = *(const float4*)(x + b * 256 + sub * 16 + k * 4);
                    if (pro == PRO_RMSNORM) {
                        s += (double)(v[rd][k].x * v[rd][k].x);
                        s += (double)(v[rd][k].y * v[rd][k].y);
                        s += (double)(v[rd][k].z * v[rd][k].z);
                        s += (double)(v[rd][k].w * v[rd][k].w);
                    }
                }
            }
        }
    }
    float scale = 1.0f;
    if (pro == PRO_RMSNORM) {
        if (wave_live) {
            s = wave_sum_fast(s);
            if (lane == 0) L.red[wv] = s;
        } else if (lane == 0) {
            L.red[wv] = 0.0;
        }
        __syncthreads();
        if (wave_live) {
            double tot = 0.0;
            for (int w = 0; w < NW; ++w) tot += L.red[w];
            const float mean = (float)(tot / (double)K);
            scale = 1.0f / sqrtf(mean + eps);
        }
    } else if (pro == PRO_LAYERNORM) {
        // ggml_compute_forward_norm_f32 (ggml.c:10605-10654): double sum -> f32 mean; v = x - mean; double sum of v*v ->
        // f32 variance; scale = 1/sqrtf(variance + eps).  v replaces x in the registers.
        double s1 = 0.0;
        if (wave_live) {
#pragma unroll
            for (int rd = 0; rd < ROUNDS; ++rd) {
                if (grp + rd * NG < nblk) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        s1 += (double)v[rd][k].x; s1 += (double)v[rd][k].y; s1 += (double)v[rd][k].z; s1 += (double)v[rd][k].w;
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
                if (grp + rd * NG < nblk) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[rd][k].x -= mean; v[rd][k].y -= mean; v[rd][k].z -= mean; v[rd][k].w -= mean;
                        s2 += (double)(v[rd][k].x * v[rd][k].x); s2 += (double)(v[rd][k].y * v[rd][k].y);
                        s2 += (double)(v[rd][k].z * v[rd][k].z); s2 += (double)(v[rd][k].w * v[rd][k].w);
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
    if (wave_live) {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int b = grp + rd * NG;
            const bool live = b < nblk;            // uniform within a 16-lane row, may differ between rows of a wave
            float t[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float4 q = live ? v[rd][k] : float4{0.f, 0.f, 0.f, 0.f};
                if (live && pro != PRO_PLAIN) {
                    const float4 w4 = *(const float4*)(nw + b * 256 + sub * 16 + k * 4);
                    q.x = (q.x * scale) * w4.x;
                    q.y = (q.y * scale) * w4.y;
                    q.z = (q.z * scale) * w4.z;
                    q.w = (q.w * scale) * w4.w;
                    if (pro == PRO_LAYERNORM) {
                        const float4 b4 = *(const float4*)(nbias + b * 256 + sub * 16 + k * 4);
                        q.x += b4.x; q.y += b4.y; q.z += b4.z; q.w += b4.w;
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
#pragma unroll
                for (int k = 0; k < 4; ++k) L.q8[b * 64 + sub * 4 + k] = packed[k];
                L.bsums[b * 16 + sub] = s16;
                if ((sub & 1) == 0) L.sb[b * 8 + (sub >> 1)] = s32;
                if (sub == 0) L.yd[b] = d;
            }
        }
    }
    __syncthreads();
}

// DPP form of quad_transpose_reduce (lanes g, g^2, g^4, g^6 of a row).
DEV int quad_transpose_reduce_dpp(int p0, int p1, int p2, int p3, int c) {
    const bool odd = (c & 1) != 0;
    const int send0 = odd ? p0 : p2, send1 = odd ? p1 : p3;
    const int keep0 = odd ? p2 : p0, keep1 = odd ? p3 : p1;
    const int q0 = keep0 + lane_xor2(send0);
    const int q1 = keep1 + lane_xor2(send1);
    const bool up = (c & 2) != 0;
    const int send = up ? q0 : q1;
    const int keep = up ? q1 : q0;
    return keep + lane_xor4(send);
}
DEV float hsum8_exact_dpp(float acc) {
    const float t = acc + lane_xor1(acc);
    const float u = t + lane_xor2(t);
    return u + lane_xor4(u);
}

// Per-lane constants of the tile geometry (computed once per kernel).
struct LaneGeom {
    int r, g, c, h;           // row in tile, unit in block, chunk, AVX half
    uint32_t off_hdr, off_qs, off_qh5;     // Q4_K / Q5_K byte offsets inside a record
    uint32_t off6_d, off6_sc, off6_qh, off6_ql;  // Q6_K byte offsets inside a record
    int a45, a6;              // LDS word offsets of this lane's activation bytes inside a block (Q4/5_K, Q6_K)
    int sh16;                 // 16*(c&1): which half of the packed scale words
    int s_lo6, s_hi6, sc_sh6; // Q6_K shifts
};
DEV LaneGeom lane_geom(int lane) {
    LaneGeom G;
    G.r = lane >> 3; G.g = lane & 7; G.c = G.g >> 1; G.h = G.g & 1;
    G.off_hdr = G.r * 16;
    G.off_qs = G.r * 128 + G.g * 16;
    G.off_qh5 = 128 + G.r * 32 + G.h * 16;
    const int n = G.g >> 2, gg = G.g & 3, kq = gg >> 1;
    G.off6_d = G.r * 2;
    G.off6_sc = 16 + G.r * 16;
    G.off6_qh = 144 + G.r * 64 + n * 32 + G.h * 16;
    G.off6_ql = 656 + G.r * 128 + G.g * 16;
    G.a45 = 16 * G.c + 4 * G.h;
    G.a6 = 32 * n + 4 * gg;
    G.sh16 = 16 * (G.c & 1);
    G.s_lo6 = 2 * kq; G.s_hi6 = 4 + 2 * kq; G.sc_sh6 = 8 * gg;
    return G;
}

// rec_base: wave-uniform pointer to the (tile, block) record.

// Prompt chunks at long contexts (positions beyond 128): attn_chunk_tile_kernel's staging applied tile after tile.  A workgroup takes
// one head and NTOK consecutive tokens (a wave each); the K rows, then the V rows, of the positions they attend to pass through ONE
// LDS buffer in tiles of 64 positions, fetched once for the NTOK tokens instead of once per token (attn_chunk_wave_kernel reads
// ~1 MB of K / V per (head, token) at 2k positions through L2).  Each wave keeps its score / probability row in LDS (row_floats
// each) and carries the 8 x 8 V*P accumulators of its 128 (64) channels through the position tiles.  Arithmetic per position and
// channel, and its order, are the other kernels': scores by quads of lanes, fp16 exp table, order-free double sum, V*P in steps of
// 32 positions in order, leftovers in double after the reduce.  Barriers are workgroup-uniform: every wave — also one without a
// token, or past its own last position — walks all tiles of the workgroup.
template <int HD, int NTOK>
__global__ void __launch_bounds__(NTOK * 64) attn_chunk_long_kernel(const AttnArgsX a, int n_tok, int row_floats) {
    kernarg_touch<sizeof(AttnArgsX)>();   // (gpu.h)
    constexpr int NC = HD / 32, PB = 4, TP = 64, NT = NTOK * 64;
    constexpr int KS = HD * 2 + 64, VS = TP * 2 + 64, NCH = HD / 8, NPASS = HD / 16;   // bytes per staged K row / V row
    constexpr int TILE_BYTES = TP * KS > HD * VS ? TP * KS : HD * VS;
    CT_DYN_SMEM(smem_raw);
    unsigned char* T = smem_raw;
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = uniform_int(wave_id()), j = lane & 3, quad = lane >> 2;
    const int h = (int)blockIdx.x, tok0 = (int)blockIdx.y * NTOK, tok = tok0 + wv;
    const int hk = h / (a.n_head / a.n_head_kv);
    const bool live = tok < n_tok;
    const int pos0 = *a.pos;
    float* prob = reinterpret_cast<float*>(smem_raw + TILE_BYTES) + wv * row_floats;
    // end of the reference batch a token belongs to (see attn_fused_exact_kernel); monotone in the token index
    auto batch_end = [&](int t) {
        int nt_ = *a.n_total;
        const int bs = a.n_total[1];
        if (bs > 0) {
            const int idx = a.pos[-1] + t, base = pos0 - a.pos[-1];
            const int end = (idx / bs + 1) * bs, n_eval = nt_ - base;
            nt_ = base + (end < n_eval ? end : n_eval);
        }
        return nt_;
    };
    const int t_last = (tok0 + NTOK < n_tok ? tok0 + NTOK : n_tok) - 1;   // last token of this workgroup
    const int n_kv_wg = pos0 + t_last + 1, n_tot_wg = batch_end(t_last);
    const int n_kv = live ? pos0 + tok + 1 : 0;
    const int n_tot = live ? batch_end(tok) : 0;
    const int np = n_tot & ~31, nl = n_kv - np;         // full 32-steps; leftover positions (< 32 when > 0), wave-uniform
    // ---- scores ------------------------------------------------------------------------------------------------------------
    float mx = -INFINITY;
    {
        F32x2 qf[NC][4];   // the fma chains of the eight AVX lanes run two lanes per instruction (v_pk_fma_f32: each half is the IEEE fma)
        const uint16_t* qrow = a.q_f16 + (size_t)(live ? tok : 0) * a.q_stride + (size_t)h * HD;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float q8[8];
            unpack8_f16(ld16(qrow + 32 * c + 8 * j), q8);
#pragma unroll
            for (int l = 0; l < 4; ++l) qf[c][l] = pk2(q8[2 * l], q8[2 * l + 1]);
        }
        const uint16_t* kb = a.kcache + (size_t)hk * a.n_ctx * HD;
        // the next tile's 16-byte pieces travel in registers while the current tile is used (a tile load is a ~1.5 us round trip to
        // L2 that nothing else hides: one workgroup per CU)
        constexpr int KPT = (TP * NCH + NT - 1) / NT;   // pieces per thread
        u32x4 nx[KPT];
        auto k_fetch = [&](int t0) {
            const int rows = n_kv_wg - t0 < TP ? n_kv_wg - t0 : TP;
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                const int i = tid + q * NT, p = i / NCH, c = i - p * NCH;
                if (i < rows * NCH) nx[q] = ld16(kb + (size_t)(t0 + p) * HD + 8 * c);
            }
        };
        k_fetch(0);
        for (int t0 = 0; t0 < n_kv_wg; t0 += TP) {
            __syncthreads();                        // the previous tile has been read by every wave
            {
                const int rows = n_kv_wg - t0 < TP ? n_kv_wg - t0 : TP;
#pragma unroll
                for (int q = 0; q < KPT; ++q) {
                    const int i = tid + q * NT, p = i / NCH, c = i - p * NCH;
                    if (i < rows * NCH) *(u32x4*)(T + p * KS + c * 16) = nx[q];
                }
            }
            __syncthreads();
            if (t0 + TP < n_kv_wg) k_fetch(t0 + TP);
            if (t0 < n_kv) {
                u32x4 kv[PB][NC];
#pragma unroll
                for (int u = 0; u < PB; ++u)
#pragma unroll
                    for (int c = 0; c < NC; ++c) kv[u][c] = *(const u32x4*)(T + (u * 16 + quad) * KS + (32 * c + 8 * j) * 2);   // rows past `rows`: stale, unused
#pragma unroll
                for (int u = 0; u < PB; ++u) {
                    const int p = t0 + u * 16 + quad;
                    F32x2 ac[4] = {pk2(0.0f, 0.0f), pk2(0.0f, 0.0f), pk2(0.0f, 0.0f), pk2(0.0f, 0.0f)};
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        float kf[8];
                        unpack8_f16(kv[u][c], kf);
#pragma unroll
                        for (int l = 0; l < 4; ++l) ac[l] = pk_fma_f32(pk2(kf[2 * l], kf[2 * l + 1]), qf[c][l], ac[l]);
                    }
                    float acc[8];
#pragma unroll
                    for (int l = 0; l < 4; ++l) { acc[2 * l] = pk_lo(ac[l]); acc[2 * l + 1] = pk_hi(ac[l]); }
                    const float sc = f16dot_reduce_exact(acc, j) * a.kq_scale;
                    if (p < n_kv) {
                        mx = fmaxf(mx, sc);
                        if (j == 0) prob[p] = sc;
                    }
                }
            }
        }
    }
    // ---- softmax over the wave's own row -----------------------------------------------------------------------------------
    mx = wave_max(mx);
    wave_lds_sync();
    {
        double sum = 0.0;
        for (int i0 = 0; i0 < n_kv; i0 += 64) {
            const int i = i0 + lane;
            if (i < n_kv) { const float e = f16_bits_to_f32(a.exp_tab[f32_to_f16_bits(prob[i] - mx)]); prob[i] = e; sum += (double)e; }
        }
        sum = wave_sum(sum);
        const float inv = (float)(1.0 / sum);
        for (int i = lane; i < n_kv; i += 64) prob[i] = f16_bits_to_f32(f32_to_f16_bits(prob[i] * inv));
        for (int i = n_kv + lane; i < np; i += 64) prob[i] = 0.0f;   // masked columns of this batch
    }
    wave_lds_sync();
    // ---- V * P: 16 channels per pass (a quad each), the accumulators of all passes carried through the position tiles ----------
    F32x2 acc[NPASS][4];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
        for (int l = 0; l < 4; ++l) acc[ps][l] = pk2(0.0f, 0.0f);
    const uint16_t* vb = a.vcache + (size_t)hk * HD * a.v_stride;
    // the tile after whose full steps this wave is finished: the one that holds the leftover positions np .. n_kv - 1, or (no
    // leftovers: np >= n_kv >= 1) the one that holds its last full step
    const int t_left = (nl > 0 ? np : np - 1) & ~(TP - 1);
    constexpr int VPT = (HD * (TP / 8) + NT - 1) / NT;
    u32x4 nv[VPT];
    auto v_fetch = [&](int t0) {                        // 64 positions of every channel (past the row's end: slack bytes, never used)
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            const int i = tid + q * NT, ch = i / (TP / 8), c = i - ch * (TP / 8);
            if (i < HD * (TP / 8)) nv[q] = ld16(vb + (size_t)ch * a.v_stride + t0 + 8 * c);
        }
    };
    v_fetch(0);
    for (int t0 = 0; t0 < n_tot_wg; t0 += TP) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            const int i = tid + q * NT, ch = i / (TP / 8), c = i - ch * (TP / 8);
            if (i < HD * (TP / 8)) *(u32x4*)(T + ch * VS + c * 16) = nv[q];
        }
        __syncthreads();
        if (t0 + TP < n_tot_wg) v_fetch(t0 + TP);
        if (t0 < np) {
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const unsigned char* vrow = T + (ps * 16 + quad) * VS;
#pragma unroll
                for (int u = 0; u < TP / 32; ++u) {
                    if (t0 + 32 * u < np) {
                        float vf[8];
                        unpack8_f16(*(const u32x4*)(vrow + (32 * u + 8 * j) * 2), vf);
                        const float* pr = &prob[t0 + 32 * u + 8 * j];
#pragma unroll
                        for (int l = 0; l < 4; ++l) acc[ps][l] = pk_fma_f32(pk2(vf[2 * l], vf[2 * l + 1]), pk2(pr[2 * l], pr[2 * l + 1]), acc[ps][l]);
                    }
                }
            }
        }
        if (live && t0 == t_left) {                     // every full step of this wave is done: reduce, leftovers, store
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int d = ps * 16 + quad;
                float a8[8];
#pragma unroll
                for (int l = 0; l < 4; ++l) { a8[2 * l] = pk_lo(acc[ps][l]); a8[2 * l + 1] = pk_hi(acc[ps][l]); }
                double sumf = (double)f16dot_reduce_exact(a8, j);
                if (nl > 0) {
                    const unsigned char* vrow = T + d * VS + (np - t0) * 2;
                    float lf[32];
#pragma unroll
                    for (int c = 0; c < 4; ++c) unpack8_f16(*(const u32x4*)(vrow + c * 16), lf + 8 * c);
#pragma unroll
                    for (int i = 0; i < 32; ++i) {
                        if (i < nl) sumf += (double)(lf[i] * prob[np + i]);
                    }
                }
                if (j == 0) a.out[(size_t)tok * a.out_stride + (size_t)h * HD + d] = (float)sumf;
            }
        }
    }
}

// The EIGHT-token form of attn_chunk_long_kernel (contexts above 2048: eight probability rows fill the LDS, one workgroup of eight waves per CU) with THREE
// K / V tiles in flight in registers instead of one, every request unconditional and the loops peeled so that hipcc counts the ring (vmcnt(N)): a tile load
// is a ~1.5 us round trip that two waves per SIMD do not cover.  2048-token prompt (7B, ctx 2304): 12 713 against 12 379 tok/s.  A kernel of its own because
// the same source as a sixteen-token instantiation (128 registers) spills (profiles/r05_decode_ab.txt section 6); arithmetic and order are attn_chunk_long_kernel's.
template <bool B> struct A9ReqX { static constexpr bool value = B; };
template <int HD, int NTOK>
__global__ void __launch_bounds__(NTOK * 64) attn_chunk_long8_kernel(const AttnArgsX a, int n_tok, int row_floats) {
    kernarg_touch<sizeof(AttnArgsX)>();   // (gpu.h)
    constexpr int NC = HD / 32, PB = 4, TP = 64, NT = NTOK * 64;
    constexpr int KS = HD * 2 + 64, VS = TP * 2 + 64, NCH = HD / 8, NPASS = HD / 16;   // bytes per staged K row / V row
    constexpr int TILE_BYTES = TP * KS > HD * VS ? TP * KS : HD * VS;
    CT_DYN_SMEM(smem_raw);
    unsigned char* T = smem_raw;
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = uniform_int(wave_id()), j = lane & 3, quad = lane >> 2;
    const int h = (int)blockIdx.x, tok0 = (int)blockIdx.y * NTOK, tok = tok0 + wv;
    const int hk = h / (a.n_head / a.n_head_kv);
    const bool live = tok < n_tok;
    const int pos0 = *a.pos;
    float* prob = reinterpret_cast<float*>(smem_raw + TILE_BYTES) + wv * row_floats;
    // end of the reference batch a token belongs to (see attn_fused_exact_kernel); monotone in the token index
    auto batch_end = [&](int t) {
        int nt_ = *a.n_total;
        const int bs = a.n_total[1];
        if (bs > 0) {
            const int idx = a.pos[-1] + t, base = pos0 - a.pos[-1];
            const int end = (idx / bs + 1) * bs, n_eval = nt_ - base;
            nt_ = base + (end < n_eval ? end : n_eval);
        }
        return nt_;
    };
    const int t_last = (tok0 + NTOK < n_tok ? tok0 + NTOK : n_tok) - 1;   // last token of this workgroup
    const int n_kv_wg = pos0 + t_last + 1, n_tot_wg = batch_end(t_last);
    const int n_kv = live ? pos0 + tok + 1 : 0;
    const int n_tot = live ? batch_end(tok) : 0;
    const int np = n_tot & ~31, nl = n_kv - np;         // full 32-steps; leftover positions (< 32 when > 0), wave-uniform
    // ---- scores ------------------------------------------------------------------------------------------------------------
    float mx = -INFINITY;
    {
        F32x2 qf[NC][4];   // the fma chains of the eight AVX lanes run two lanes per instruction (v_pk_fma_f32: each half is the IEEE fma)
        const uint16_t* qrow = a.q_f16 + (size_t)(live ? tok : 0) * a.q_stride + (size_t)h * HD;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float q8[8];
            unpack8_f16(ld16(qrow + 32 * c + 8 * j), q8);
#pragma unroll
            for (int l = 0; l < 4; ++l) qf[c][l] = pk2(q8[2 * l], q8[2 * l + 1]);
        }
        const uint16_t* kb = a.kcache + (size_t)hk * a.n_ctx * HD;
        // The next THREE tiles' 16-byte pieces travel in registers while the current tile is used: a tile load is a ~1.5 us round trip to L2, a tile's
        // dot products ~0.6 us, and the one workgroup a CU holds (the probability rows fill its LDS) has nothing else to hide it behind — with one tile
        // ahead (rounds 3-5) the kernel ran at the fetch latency: 234 us for the last chunk of a 2048-token prompt against ~80 us of VALU work.
        // Every request is unconditional (rows past the workgroup's last position re-read that row): hipcc counts the ring with vmcnt(N).
        constexpr int KPT = (TP * NCH + NT - 1) / NT;   // pieces per thread
        constexpr int RD = 3;                           // tiles in flight
        u32x4 nx[RD][KPT];
        const int last_row = n_kv_wg - 1;
        auto k_fetch = [&](u32x4 (&dst)[KPT], int t0) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                int i = tid + q * NT;
                i = i < TP * NCH ? i : TP * NCH - 1;
                const int p = i / NCH, c = i - p * NCH;
                const int pp = t0 + p < last_row ? t0 + p : last_row;
                dst[q] = ld16(kb + (size_t)pp * HD + 8 * c);
            }
        };
        auto k_tile = [&](u32x4 (&src)[KPT], int t0, auto REQ) __attribute__((always_inline)) {
            __syncthreads();                        // the previous tile has been read by every wave
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                const int i = tid + q * NT, p = i / NCH, c = i - p * NCH;
                if (i < TP * NCH) *(u32x4*)(T + p * KS + c * 16) = src[q];
            }
            __syncthreads();
            if constexpr (decltype(REQ)::value) k_fetch(src, t0 + RD * TP);
            if (t0 < n_kv) {
                u32x4 kv[PB][NC];
#pragma unroll
                for (int u = 0; u < PB; ++u)
#pragma unroll
                    for (int c = 0; c < NC; ++c) kv[u][c] = *(const u32x4*)(T + (u * 16 + quad) * KS + (32 * c + 8 * j) * 2);   // rows past the last position: copies of it, unused
#pragma unroll
                for (int u = 0; u < PB; ++u) {
                    const int p = t0 + u * 16 + quad;
                    F32x2 ac[4] = {pk2(0.0f, 0.0f), pk2(0.0f, 0.0f), pk2(0.0f, 0.0f), pk2(0.0f, 0.0f)};
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        float kf[8];
                        unpack8_f16(kv[u][c], kf);
#pragma unroll
                        for (int l = 0; l < 4; ++l) ac[l] = pk_fma_f32(pk2(kf[2 * l], kf[2 * l + 1]), qf[c][l], ac[l]);
                    }
                    float acc[8];
#pragma unroll
                    for (int l = 0; l < 4; ++l) { acc[2 * l] = pk_lo(ac[l]); acc[2 * l + 1] = pk_hi(ac[l]); }
                    const float sc = f16dot_reduce_exact(acc, j) * a.kq_scale;
                    if (p < n_kv) {
                        mx = fmaxf(mx, sc);
                        if (j == 0) prob[p] = sc;
                    }
                }
            }
        };
#pragma unroll
        for (int d = 0; d < RD; ++d) k_fetch(nx[d], d * TP);
        int t0 = 0;
        for (; t0 + RD * TP <= n_kv_wg; t0 += RD * TP) {   // whole rounds: no condition around a tile or a request
#pragma unroll
            for (int d = 0; d < RD; ++d) k_tile(nx[d], t0 + d * TP, A9ReqX<true>{});
        }
#pragma unroll
        for (int d = 0; d < RD; ++d)                        // the last, partial round requests nothing (workgroup-uniform conditions: the barriers stay whole)
            if (t0 + d * TP < n_kv_wg) k_tile(nx[d], t0 + d * TP, A9ReqX<false>{});
    }
    // ---- softmax over the wave's own row -----------------------------------------------------------------------------------
    mx = wave_max(mx);
    wave_lds_sync();
    {
        double sum = 0.0;
        for (int i0 = 0; i0 < n_kv; i0 += 64) {
            const int i = i0 + lane;
            if (i < n_kv) { const float e = f16_bits_to_f32(a.exp_tab[f32_to_f16_bits(prob[i] - mx)]); prob[i] = e; sum += (double)e; }
        }
        sum = wave_sum(sum);
        const float inv = (float)(1.0 / sum);
        for (int i = lane; i < n_kv; i += 64) prob[i] = f16_bits_to_f32(f32_to_f16_bits(prob[i] * inv));
        for (int i = n_kv + lane; i < np; i += 64) prob[i] = 0.0f;   // masked columns of this batch
    }
    wave_lds_sync();
    // ---- V * P: 16 channels per pass (a quad each), the accumulators of all passes carried through the position tiles ----------
    F32x2 acc[NPASS][4];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
        for (int l = 0; l < 4; ++l) acc[ps][l] = pk2(0.0f, 0.0f);
    const uint16_t* vb = a.vcache + (size_t)hk * HD * a.v_stride;
    // the tile after whose full steps this wave is finished: the one that holds the leftover positions np .. n_kv - 1, or (no
    // leftovers: np >= n_kv >= 1) the one that holds its last full step
    const int t_left = (nl > 0 ? np : np - 1) & ~(TP - 1);
    constexpr int VPT = (HD * (TP / 8) + NT - 1) / NT;
    constexpr int RDV = 3;                              // tiles in flight, as for the K rows
    u32x4 nv[RDV][VPT];
    const int last_tile = n_tot_wg > 0 ? (n_tot_wg - 1) & ~(TP - 1) : 0;   // requests past it re-read it
    auto v_fetch = [&](u32x4 (&dst)[VPT], int t0) __attribute__((always_inline)) {   // 64 positions of every channel (past the row's end: slack bytes, never used)
        const int tt = t0 < last_tile ? t0 : last_tile;
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            int i = tid + q * NT;
            i = i < HD * (TP / 8) ? i : HD * (TP / 8) - 1;
            const int ch = i / (TP / 8), c = i - ch * (TP / 8);
            dst[q] = ld16(vb + (size_t)ch * a.v_stride + tt + 8 * c);
        }
    };
#pragma unroll
    for (int d = 0; d < RDV; ++d) v_fetch(nv[d], d * TP);
    // one tile: stage, request the tile three further on (REQ), the full 32-steps of this wave in it, and — in the wave's last tile — reduce, leftovers, store
    auto v_tile = [&](u32x4 (&src)[VPT], int t0, auto REQ) __attribute__((always_inline)) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            const int i = tid + q * NT, ch = i / (TP / 8), c = i - ch * (TP / 8);
            if (i < HD * (TP / 8)) *(u32x4*)(T + ch * VS + c * 16) = src[q];
        }
        __syncthreads();
        if constexpr (decltype(REQ)::value) v_fetch(src, t0 + RDV * TP);
        if (t0 < np) {
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const unsigned char* vrow = T + (ps * 16 + quad) * VS;
#pragma unroll
                for (int u = 0; u < TP / 32; ++u) {
                    if (t0 + 32 * u < np) {
                        float vf[8];
                        unpack8_f16(*(const u32x4*)(vrow + (32 * u + 8 * j) * 2), vf);
                        const float* pr = &prob[t0 + 32 * u + 8 * j];
#pragma unroll
                        for (int l = 0; l < 4; ++l) acc[ps][l] = pk_fma_f32(pk2(vf[2 * l], vf[2 * l + 1]), pk2(pr[2 * l], pr[2 * l + 1]), acc[ps][l]);
                    }
                }
            }
        }
        if (live && t0 == t_left) {                     // every full step of this wave is done: reduce, leftovers, store
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int d = ps * 16 + quad;
                float a8[8];
#pragma unroll
                for (int l = 0; l < 4; ++l) { a8[2 * l] = pk_lo(acc[ps][l]); a8[2 * l + 1] = pk_hi(acc[ps][l]); }
                double sumf = (double)f16dot_reduce_exact(a8, j);
                if (nl > 0) {
                    const unsigned char* vrow = T + d * VS + (np - t0) * 2;
                    float lf[32];
#pragma unroll
                    for (int c = 0; c < 4; ++c) unpack8_f16(*(const u32x4*)(vrow + c * 16), lf + 8 * c);
#pragma unroll
                    for (int i = 0; i < 32; ++i) {
                        if (i < nl) sumf += (double)(lf[i] * prob[np + i]);
                    }
                }
                if (j == 0) a.out[(size_t)tok * a.out_stride + (size_t)h * HD + d] = (float)sumf;
            }
        }
    };
    int t0 = 0;
    for (; t0 + RDV * TP <= n_tot_wg; t0 += RDV * TP) {
#pragma unroll
        for (int d = 0; d < RDV; ++d) v_tile(nv[d], t0 + d * TP, A9ReqX<true>{});
    }
#pragma unroll
    for (int d = 0; d < RDV; ++d)
        if (t0 + d * TP < n_tot_wg) v_tile(nv[d], t0 + d * TP, A9ReqX<false>{});
}
