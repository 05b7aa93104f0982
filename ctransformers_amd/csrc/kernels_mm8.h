// Prompt chunks, order-free form (round 6): the integer sub-block dots on the 32x32x32 int8 matrix cores, f32 accumulation in any order.
//
// SURVEY.md Appendix A.3 / A.4 (the numerics contract the north star states: logits within 1e-3 of the reference CPU llm.eval(), greedy tokens
// identical): what must be reproduced are the reference's QUANTIZATION POINTS — activations re-quantized to Q8_K / Q8_0 exactly as
// quantize_row_q8_K / quantize_row_q8_0 do (k_quants.c:1191-1226, ggml.c:1208-1300), integer dot products per 32- (Q6_K: 16-) element sub-block exact —
// while "any summation order is fine" for the f32 part.  kernels_pg.h / kernels_pf.h go further and mirror the eight AVX accumulator lanes of
// ggml_vec_dot_q4_K_q8_K & friends bit for bit; that chain is what keeps them at K = 4 / K = 32-on-f16 matrix instructions and 14 vector instructions
// per matrix instruction.  Here (the reference's own free-order GPU tiles are ggml-cuda.cu:3289-3752, mul_mat_q):
//   K-quants   p_j   = sum_{32} q_w * q8                  one v_mfma_i32_32x32x32_i8 per sub-block: 32 rows x 32 tokens, exact int32
//              isum += sc_j * p_j                         v_mad_i32_i24 (|p_j| <= 32 * 31 * 128 < 2^23), exact int32 (|isum| < 2^31: A.4)
//              acc  += y.d * (d * (float)isum - dmin * sum_j m_j * bsums_j)        once per 256-block; the min term is one v_mfma_f32_32x32x16_f16 on
//                                                                                   integer-valued halves (exact: gpu.h)
//   Q6_K       sub-blocks of 16 with int8 scales: two matrix instructions per 32 elements, each with one K-chunk of the weight operand zeroed
//   Q8_0       p_j on the same instruction with C = 0x4B400000 (the result read as a float is 1.5 * 2^23 + p_j: no conversion), then
//              acc += (p_j * fp16(x.d)) * fp16(y.d) per 32-block (the reference: acc = fma(x.d * y.d, p_j, acc), ggml.c:3321)
// Against the reference every f32 product / sum is the same real number rounded in a different place: ~1e-7 relative per dot (tests: logits within
// 1e-3 of oracle/_ref — measured ~1e-5 on the 7B shapes — and the greedy continuation compared token by token).
//
// Shapes.  A wave owns ONE row tile (32 weight rows: lane & 31 = row, so a lane's sixteen results of a matrix instruction are sixteen tokens of ITS row
// and d, dmin, scales, mins are lane-local) x NTT token tiles (32 tokens each) x one K-slice.  A workgroup = 8 waves = RW row tiles x KS K-slices
// (RW * KS = 8) over one token group; the K-slices' partial sums meet in LDS at the end, in slice order (deterministic).  Weights: LAYOUT_M8 records
// (quant.h), streamed straight into registers, one 16-byte load per lane and piece, a K-step ahead.  Activations: the quantize kernel writes per
// (K-step, token tile) one UNIT in operand order (a ds_read_b128 of lane-linear bytes: conflict-free); the workgroup copies the units of its next step
// into the other half of LDS by LDS-DMA while it computes (drained and fenced by a barrier at the end of every step — MI355X_MICROARCH.md: an LDS-DMA is
// ordered for a ds_read by the issuing wave's vmcnt and a barrier the reader has passed).
#pragma once
#include "kernels_pg.h"

constexpr int kMm8Waves = 8;
constexpr int kMm8Unit = 10240;   // K-quant activations, one (256-block, token tile): q8[8 j][64 lanes][16] | bsum16 as halves [64 lanes][16] | y.d[32] f32 | 1152 * sum(q8)[32] i32 | pad to 1 KB pieces
constexpr int kMm8UnitB = 9216;   // Q8_0 activations, one (8 blocks, token tile):     q8[8 j][64 lanes][16] | y.d[8 j][32] f32
CT_HD static inline int mm8_unit_bytes(int type) { return is_block32(type) ? kMm8UnitB : kMm8Unit; }

struct Mm8Args {
    MatvecArgs m;          // jobs (w.m8 = LAYOUT_M8 records; pair0 = first tile of the job in this launch) and epilogue operands
    const uint8_t* acts;   // units [K-step][token tile of the chunk]
    int n_tok;             // tokens in this chunk
    int ntt;               // token tiles of the image (ceil(n_tok / 32))
    int n_tiles;           // row tiles over all jobs of the launch
    int nb;                // K-steps per row
    int ld_out, ld_res, ld_q;
};

// ---- activation images --------------------------------------------------------------------------------------------------------------------------
// One workgroup per token: (RMSNorm / LayerNorm ->) Q8_K exactly as the decode prologue does it (kernels_exact.h), written in operand order.
// Workgroup -> token as in pg_quantize_kernel: the eight tokens whose 16-byte pieces fill one 128-byte line go to workgroups on one XCD.
template <int MAXK, bool LN, int NT = 1024>
__global__ void __launch_bounds__(NT) mm8_quantize_q8k_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ nw, int K, int pro, float eps,
                                                                uint8_t* __restrict__ img, int ntt, const float* __restrict__ nb_, int n_tok) {
    __shared__ ActLdsX<MAXK> L;
    int t = (int)blockIdx.x;
    {
        const int G = (int)gridDim.x >> 3;
        t = (t % G) * 8 + t / G;
        if (t >= n_tok) return;
    }
    const int tid = (int)threadIdx.x;
    if constexpr (LN) prologue_q8k_exact16_ln<NT, MAXK>(L, x + (size_t)t * ldx, nw, K, pro, eps, nb_);
    else prologue_q8k_exact16<NT, MAXK>(L, x + (size_t)t * ldx, nw, K, pro, eps);
    const int nb = K >> 8, tt = t >> 5, ti = t & 31;
    for (int i = tid; i < nb * 16; i += NT) {   // (block, sub-block j, K-chunk c): elements 32j + 16c .. + 15
        const int b = i >> 4, j = (i >> 1) & 7, c = i & 1;
        const int* w = &L.q8[b * 64 + j * 8 + c * 4];
        uint32_t* dst = (uint32_t*)(img + ((size_t)b * ntt + tt) * kMm8Unit + j * 1024 + (c * 32 + ti) * 16);
        dst[0] = (uint32_t)w[0]; dst[1] = (uint32_t)w[1]; dst[2] = (uint32_t)w[2]; dst[3] = (uint32_t)w[3];
    }
    for (int i = tid; i < nb * 2; i += NT) {    // sums of 16 as halves (|sum| <= 2048: exact), K-chunk c = sums 8c .. 8c + 7
        const int b = i >> 1, c = i & 1;
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = (uint32_t)f32_to_f16_bits((float)L.bsums[b * 16 + 8 * c + 2 * k]) | ((uint32_t)f32_to_f16_bits((float)L.bsums[b * 16 + 8 * c + 2 * k + 1]) << 16);
        uint32_t* dst = (uint32_t*)(img + ((size_t)b * ntt + tt) * kMm8Unit + 8192 + (c * 32 + ti) * 16);
        dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];
    }
    for (int b = tid; b < nb; b += NT) {
        uint8_t* u = img + ((size_t)b * ntt + tt) * kMm8Unit;
        *(float*)(u + 9216 + ti * 4) = L.yd[b];
        int sum = 0;   // 1152 * (sum of the block's quants): the Q5_K digit planes are stored minus 128 (mm8_step)
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += L.bsums[b * 16 + k];
        *(int*)(u + 9344 + ti * 4) = 1152 * sum;
    }
}

// Q8_0 activations (quantize_row_q8_0's AVX2 form, kernels_q32.h) for the 32-block weight types; a K-step = 8 blocks, blocks behind a row's end are
// zero quants with y.d = 0.
template <int MAXK>
__global__ void __launch_bounds__(1024) mm8_quantize_q80_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ nw, int K, int pro, float eps,
                                                                  uint8_t* __restrict__ img, int ntt, const float* __restrict__ nbias, int n_tok) {
    __shared__ ActLdsQ32<MAXK> L;
    int t = (int)blockIdx.x;
    {
        const int G = (int)gridDim.x >> 3;
        t = (t % G) * 8 + t / G;
        if (t >= n_tok) return;
    }
    const int tid = (int)threadIdx.x;
    prologue_q8_0<MAXK>(L, x + (size_t)t * ldx, nw, K, pro, eps, nbias);
    const int nblk = K >> 5, ns = (nblk + 7) >> 3, tt = t >> 5, ti = t & 31;
    for (int i = tid; i < ns * 16; i += 1024) {   // (K-step, block j, K-chunk c)
        const int s = i >> 4, j = (i >> 1) & 7, c = i & 1, b = 8 * s + j;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (b < nblk) {
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = (uint32_t)L.q8[((b >> 2) * 8 + 4 * c + k) * 4 + (b & 3)];
        }
        uint32_t* dst = (uint32_t*)(img + ((size_t)s * ntt + tt) * kMm8UnitB + j * 1024 + (c * 32 + ti) * 16);
        dst[0] = w[0]; dst[1] = w[1]; dst[2] = w[2]; dst[3] = w[3];
    }
    for (int b = tid; b < ns * 8; b += 1024)
        *(float*)(img + ((size_t)(b >> 3) * ntt + tt) * kMm8UnitB + 8192 + ((b & 7) * 32 + ti) * 4) = b < nblk ? L.yd[b] : 0.0f;
}

// ---- weight records -> registers -------------------------------------------------------------------------------------------------------------------
template <int TYPE> struct Mm8W;
template <> struct Mm8W<GT_Q4_K> { u32x4 h, q[4]; };
template <> struct Mm8W<GT_Q5_K> { u32x4 h, qh, q[4]; };
template <> struct Mm8W<GT_Q6_K> { u32x4 sc, ql[4], qh[2]; uint32_t d; };
template <> struct Mm8W<GT_Q8_0> { u32x4 d, q[8]; };
template <> struct Mm8W<GT_Q4_0> { u32x4 d, q[8]; };

template <int TYPE> DEV Mm8W<TYPE> mm8_load(const uint8_t* __restrict__ rec, int lane) {
    Mm8W<TYPE> R;
    const int r = lane & 31;
    if constexpr (TYPE == GT_Q4_K) {
        R.h = ld16(rec + r * 16);
#pragma unroll
        for (int g = 0; g < 4; ++g) R.q[g] = ld_stream16(rec + 512 + g * 1024 + lane * 16);
    } else if constexpr (TYPE == GT_Q5_K) {
        R.h = ld16(rec + r * 16);
        R.qh = ld_stream16(rec + 512 + lane * 16);
#pragma unroll
        for (int g = 0; g < 4; ++g) R.q[g] = ld_stream16(rec + 1536 + g * 1024 + lane * 16);
    } else if constexpr (TYPE == GT_Q6_K) {
#pragma unroll
        for (int g = 0; g < 4; ++g) R.ql[g] = ld_stream16(rec + g * 1024 + lane * 16);
#pragma unroll
        for (int h = 0; h < 2; ++h) R.qh[h] = ld_stream16(rec + 4096 + h * 1024 + lane * 16);
        R.sc = ld16(rec + 6144 + r * 16);
        R.d = (uint32_t) * (const uint16_t*)(rec + 6656 + r * 2);
    } else if constexpr (TYPE == GT_Q8_0) {
        R.d = ld16(rec + r * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) R.q[j] = ld_stream16(rec + 512 + j * 1024 + lane * 16);
    } else {
        R.d = ld16(rec + r * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) R.q[j] = ld16(rec + 512 + j * 512 + r * 16);
    }
    return R;
}

// The record's registers have landed (the caller drained the memory counter by hand): hipcc tracks only the loads it issued itself, and without this it
// waits `vmcnt(number of its younger loads)` before the first use of the record — which, with the stage copy's LDS-DMA pieces older than those loads,
// makes every step start by waiting for its own copy to land.
template <int TYPE> DEV void mm8_landed(Mm8W<TYPE>& R) {
#ifndef CT_EMU
    if constexpr (TYPE == GT_Q4_K) asm volatile("" : "+v"(R.h), "+v"(R.q[0]), "+v"(R.q[1]), "+v"(R.q[2]), "+v"(R.q[3]));
    else if constexpr (TYPE == GT_Q5_K) asm volatile("" : "+v"(R.h), "+v"(R.qh), "+v"(R.q[0]), "+v"(R.q[1]), "+v"(R.q[2]), "+v"(R.q[3]));
    else if constexpr (TYPE == GT_Q6_K) asm volatile("" : "+v"(R.sc), "+v"(R.ql[0]), "+v"(R.ql[1]), "+v"(R.ql[2]), "+v"(R.ql[3]), "+v"(R.qh[0]), "+v"(R.qh[1]), "+v"(R.d));
    else asm volatile("" : "+v"(R.d), "+v"(R.q[0]), "+v"(R.q[1]), "+v"(R.q[2]), "+v"(R.q[3]), "+v"(R.q[4]), "+v"(R.q[5]), "+v"(R.q[6]), "+v"(R.q[7]));
#else
    (void)R;
#endif
}

#ifdef CT_EMU
static inline u32x4 lds16(const uint8_t* p) { u32x4 r; memcpy(&r, p, 16); return r; }
#else
DEV u32x4 lds16(const uint8_t* p) { return *(const u32x4*)p; }
#endif

// the sixteen accumulators have their values at this point of the program (a scheduling fence on values)
DEV void mm8_pin(f32x16& a) {
#ifndef CT_EMU
    asm volatile("" : "+v"(a));
#else
    (void)a;
#endif
}

// acc[i] += p[i] * s for the sixteen results of a matrix instruction (s lane-local: the scale of this lane's row).  NOT through gpu.h's asm mad24: the
// operands come straight out of a matrix instruction, and hipcc pads the MFMA-write -> VALU-read hazard only for instructions it can see (with the asm
// form the first multiply-adds read the registers before the product had landed: wrong sums on hardware, right ones in the emulator).  The builtin
// multiply + add selects v_mad_i32_i24; the fence keeps hipcc from re-associating the chain of one block into multiplies and three-operand adds.
DEV void mm8_mad16(i32x16& acc, const i32x16& p, int s) {
#ifdef CT_EMU
    for (int i = 0; i < 16; ++i) acc[i] = p[i] * s + acc[i];
#else
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __mul24(p[i], s) + acc[i];
    asm volatile("" : "+v"(acc));
#endif
}

// The sixteen y.d of this lane's tokens (register i <-> token (i & 3) + 8 (i >> 2) + 4 (lane >> 5) of the tile): four broadcast 16-byte LDS reads.
DEV void mm8_yd16(const uint8_t* yd /* f32[32] of the tile */, int lane, float (&da)[16]) {
    const int h = lane >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32x4 v = lds16(yd + (8 * k + 4 * h) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) da[4 * k + e] = bits_to_f32(v[e]);
    }
}

// One K-step (256 elements) of the lane's row against the NTT token tiles whose units start at `stage`.
// `feed(slot)`, slot = 8 * token tile + sub-block: the caller's memory requests of this step (the copy of the next stage), spread over the step — issued in
// one burst at the top they cost every wave of the workgroup ~2 700 cycles in which nothing computes (in-kernel stamps: the CU accepts ~64 bytes of vector
// memory requests per cycle, and all eight waves ask at once).
// `R` holds the record of this K-step on entry and the record at `next_rec` (requested, not yet waited for) on return: the K-quant forms request it once
// their operands are unpacked, the Q8_0 form — whose operands ARE the record's registers — block by block behind the last token tile's use of each.
template <int TYPE, int NTT, class Feed>
DEV void mm8_step(Mm8W<TYPE>& R, const uint8_t* __restrict__ next_rec, const uint8_t* stage, int lane, f32x16 (&acc)[NTT], const Feed& feed) {
    const int c = lane >> 5;
    if constexpr (TYPE == GT_Q4_K || TYPE == GT_Q5_K) {
        // Digit planes: sc_j = 8 h_j + l_j (h, l in 0..7), B1 = q * h_j, B0 = q * l_j are int8 operands (15 * 7 = 105; Q5_K: 31 * 7 = 217, stored as
        // 217 - 128 with the - 128 * sum(a) of the block added back per token), so the products of ALL eight sub-blocks accumulate inside the matrix
        // instruction's int32 result and the scale costs no vector instruction per sub-block: isum = 8 * D1 + D0, exact.  (The first form of this kernel
        // did isum += sc_j * p_j with sixteen v_mad_i32_i24 per matrix instruction: 53 cycles of vector issue beside 36 of the matrix core, vector-bound at
        // 0.15 matrix-core busy.  Two matrix instructions per sub-block instead: matrix-bound.)
        constexpr int UNIT = kMm8Unit;
        constexpr bool Q5 = TYPE == GT_Q5_K;
        const uint32_t W1 = R.h[1], W2 = R.h[2], W3 = R.h[3];
        uint32_t sc[8];
        sc[0] = W1 & 63u; sc[1] = bfe32(W1, 6, 6); sc[2] = bfe32(W1, 12, 6); sc[3] = bfe32(W1, 18, 6); sc[4] = bfe32(W1, 24, 6);
        sc[5] = W2 & 63u; sc[6] = bfe32(W2, 6, 6); sc[7] = bfe32(W2, 12, 6);
        // the four mins of this lane's K-chunk of the min-term product: sums of 16 with index 8c .. 8c + 7 belong to sub-blocks 4c .. 4c + 3
        const int m0 = (int)bfe32(W2, 18, 6), m1 = (int)bfe32(W2, 24, 6), m2 = (int)(W3 & 63u), m3 = (int)bfe32(W3, 6, 6);
        const int m4 = (int)bfe32(W3, 12, 6), m5 = (int)bfe32(W3, 18, 6), m6 = (int)bfe32(W3, 24, 6);
        const int m7 = (int)((W1 >> 30) | ((W2 >> 30) << 2) | ((W3 >> 30) << 4));
        const u32x4 MB = {h2_from_int(c ? m4 : m0), h2_from_int(c ? m5 : m1), h2_from_int(c ? m6 : m2), h2_from_int(c ? m7 : m3)};
        const float dw = f16_bits_to_f32((uint16_t)(R.h[0] & 0xFFFFu)), ndmw = -f16_bits_to_f32((uint16_t)(R.h[0] >> 16));
        u32x4 B1[8], B0[8];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // digit (as two 16-bit lanes: v_pk_mul_lo_u16 multiplies the byte pairs of a dword without a carry between them: 217 < 256)
            const uint32_t ha = (sc[2 * g] >> 3) * 0x10001u, la = (sc[2 * g] & 7u) * 0x10001u;
            const uint32_t hb = (sc[2 * g + 1] >> 3) * 0x10001u, lb = (sc[2 * g + 1] & 7u) * 0x10001u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t w = R.q[g][k];
                uint32_t lo = w & 0x0F0F0F0Fu, hi = (w >> 4) & 0x0F0F0F0Fu;
                if constexpr (Q5) {   // fifth bit of sub-block j: bit j of the qh byte
                    const uint32_t q = R.qh[k];
                    lo |= (2 * g < 4 ? q << (4 - 2 * g) : q >> (2 * g - 4)) & 0x10101010u;
                    hi |= (2 * g + 1 < 4 ? q << (3 - 2 * g) : q >> (2 * g - 3)) & 0x10101010u;
                }
                constexpr uint32_t X = Q5 ? 0x80808080u : 0u;
                B1[2 * g][k] = pk_mul_u16(lo, ha) ^ X; B0[2 * g][k] = pk_mul_u16(lo, la) ^ X;
                B1[2 * g + 1][k] = pk_mul_u16(hi, hb) ^ X; B0[2 * g + 1][k] = pk_mul_u16(hi, lb) ^ X;
            }
        }
        R = mm8_load<TYPE>(next_rec, lane);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const uint8_t* U = stage + tt * UNIT;
            u32x4 A = lds16(U + lane * 16);
            i32x16 D1, D0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const u32x4 An = lds16(U + (j + 1) * 1024 + lane * 16);   // (j = 7: the sums of 16, the min term's operand)
                if (j == 0) { D1 = mfma_i8_32x32x32(A, B1[0]); D0 = mfma_i8_32x32x32(A, B0[0]); }
                else { D1 = mfma_i8_32x32x32_acc(A, B1[j], D1); D0 = mfma_i8_32x32x32_acc(A, B0[j], D0); }
                A = An;
                feed(8 * tt + j);
            }
            const f32x16 M = mfma_f16_32x32x16(A, MB);
            float da[16];
            mm8_yd16(U + 9216, lane, da);
            int sa[16];
            if constexpr (Q5) {   // 1152 * (sum of the block's 256 quants) per token: what the two planes' - 128 took away
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u32x4 v = lds16(U + 9344 + (8 * k + 4 * (lane >> 5)) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) sa[4 * k + e] = (int)v[e];
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int isum = (D1[i] << 3) + D0[i];
                if constexpr (Q5) isum += sa[i];
                const float u = fmaf((float)isum, dw, M[i] * ndmw);
                acc[tt][i] = fmaf(u, da[i], acc[tt][i]);
            }
            mm8_pin(acc[tt]);
        }
    } else if constexpr (TYPE == GT_Q6_K) {
        constexpr int UNIT = kMm8Unit;
        const float dw = f16_bits_to_f32((uint16_t)(R.d & 0xFFFFu));
        int sc[16];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) sc[4 * k + e] = bfe_i32(R.sc[k], 8 * e, 8);
        const uint32_t mk0 = c ? 0u : 0xFFFFFFFFu, mk1 = ~mk0;   // the weight operand with one K-chunk zeroed: elements 0..15 (scale 2s) / 16..31 (scale 2s + 1)
        u32x4 B[8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t lw = R.ql[2 * h + (qd & 1)][k], hw = R.qh[h][k];
                    const uint32_t q = ((qd & 2 ? lw >> 4 : lw) & 0x0F0F0F0Fu) | (((hw >> (2 * qd)) & 0x03030303u) << 4);
                    B[4 * h + qd][k] = (q + 0x60606060u) ^ 0x80808080u;   // q - 32 per byte (q <= 63: no carry between bytes)
                }
        R = mm8_load<TYPE>(next_rec, lane);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const uint8_t* U = stage + tt * UNIT;
            i32x16 ai;
#pragma unroll
            for (int i = 0; i < 16; ++i) ai[i] = 0;
            u32x4 A = lds16(U + lane * 16);
            i32x16 pend0, pend1;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const u32x4 B0 = {B[s][0] & mk0, B[s][1] & mk0, B[s][2] & mk0, B[s][3] & mk0};
                const u32x4 B1 = {B[s][0] & mk1, B[s][1] & mk1, B[s][2] & mk1, B[s][3] & mk1};
                const i32x16 c0 = mfma_i8_32x32x32(A, B0);
                const i32x16 c1 = mfma_i8_32x32x32(A, B1);
                if (s < 7) A = lds16(U + (s + 1) * 1024 + lane * 16);
                if (s > 0) { mm8_mad16(ai, pend0, sc[2 * s - 2]); mm8_mad16(ai, pend1, sc[2 * s - 1]); }
                pend0 = c0; pend1 = c1;
                feed(8 * tt + s);
                sched_fence();
            }
            float da[16];
            mm8_yd16(U + 9216, lane, da);
            mm8_mad16(ai, pend0, sc[14]);
            mm8_mad16(ai, pend1, sc[15]);
            sched_fence();
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[tt][i] = fmaf((float)ai[i] * dw, da[i], acc[tt][i]);
            mm8_pin(acc[tt]);
        }
    } else {   // Q8_0 / Q4_0: eight 32-blocks, each with its own fp16 scale on both sides
        constexpr int UNIT = kMm8UnitB;
        const u32x4 dcur = R.d;   // the row's eight block scales (fp16)
        u32x4 B4[TYPE == GT_Q4_0 ? 8 : 1];
        if constexpr (TYPE == GT_Q4_0) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t n = (c ? R.q[j][k] >> 4 : R.q[j][k]) & 0x0F0F0F0Fu;
                    B4[j][k] = (n + 0x78787878u) ^ 0x80808080u;   // n - 8 per byte
                }
            R = mm8_load<TYPE>(next_rec, lane);
        }
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const uint8_t* U = stage + tt * UNIT;
            // (no hand-made software pipeline here: with the product of block j + 1 issued before the multiply-adds of block j hipcc kept all eight
            // products and their y.d rows live and spilled ~300 registers; the SIMD's other wave covers the matrix instruction's latency instead)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const u32x4 A = lds16(U + j * 1024 + lane * 16);
                i32x16 p;
                if constexpr (TYPE == GT_Q4_0) p = mfma_i8_32x32x32_bias(A, B4[j]);
                else {
                    p = mfma_i8_32x32x32_bias(A, R.q[j]);
                    if (tt == NTT - 1) R.q[j] = ld_stream16(next_rec + 512 + j * 1024 + lane * 16);   // this block's registers are free: the next K-step's block
                }
                const uint32_t dh = dcur[j >> 1];
                const float dwj = f16_bits_to_f32((uint16_t)((j & 1) ? dh >> 16 : dh & 0xFFFFu));
                const float nbd = -12582912.0f * dwj;   // exact (13 significant bits)
                float da[16];
                mm8_yd16(U + 8192 + j * 128, lane, da);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float u = fmaf(bits_to_f32((uint32_t)p[i]), dwj, nbd);   // = p * fp16(x.d), rounded once
                    acc[tt][i] = fmaf(u, da[i], acc[tt][i]);
                }
                mm8_pin(acc[tt]);   // the accumulation of block j happens HERE (left alone hipcc issues the eight products first, sinks the multiply-adds
                                    // of all blocks behind them and spills the y.d rows it has read on the way: 300 registers)
                feed(8 * tt + j);
                sched_fence();
            }
        }
        if constexpr (TYPE == GT_Q8_0) R.d = ld16(next_rec + (lane & 31) * 16);
    }
}

// ---- the launch ---------------------------------------------------------------------------------------------------------------------------------
// grid (ceil(n_tiles / RW), token groups of NTT tiles), 512 threads.  LDS: two stage buffers of KS * NTT units; the waves' partial results reuse them.
template <int TYPE, int NTT, int KS>
__global__ void __launch_bounds__(512, 2) mm8_kernel(const uint8_t* acts0, int nb0, int n_tiles0, const Mm8Args a) {
#ifndef CT_EMU
    __builtin_assume(a.acts == acts0); __builtin_assume(a.nb == nb0); __builtin_assume(a.n_tiles == n_tiles0);
#else
    (void)acts0; (void)nb0; (void)n_tiles0;
#endif
    kernarg_touch<16 + sizeof(Mm8Args)>();
    CT_DYN_SMEM(smem);
    constexpr int RW = kMm8Waves / KS, SU = KS * NTT, UNIT = (TYPE == GT_Q8_0 || TYPE == GT_Q4_0) ? kMm8UnitB : kMm8Unit, SB = SU * UNIT;
    constexpr int PP = UNIT / 1024;                 // DMA pieces per unit (1 KB each: 16 bytes per lane)
    static_assert(UNIT % 1024 == 0, "units are whole 1 KB pieces");
    constexpr int REC = TYPE == GT_Q4_K ? 4608 : (TYPE == GT_Q5_K ? 5632 : (TYPE == GT_Q6_K ? 6720 : (TYPE == GT_Q8_0 ? 8704 : 4608)));
    const MatvecArgs& m = a.m;
    const int lane = lane_id(), wv = uniform_int(wave_id());
    const int rw = wv % RW, ks = wv / RW;
    const int bx = (int)blockIdx.x, tg = (int)blockIdx.y;
    const int nb = a.nb, nsteps = (nb + KS - 1) / KS;
    int tile = bx * RW + rw;
    tile = tile < a.n_tiles ? tile : a.n_tiles - 1;   // a surplus wave walks the last tile again (it takes part in the copies and barriers) and stores nothing
    const uint8_t* w8 = m.job[0].w.m8;
    int tl = tile;
    if (m.njobs > 1 && tile >= m.job[1].pair0) { w8 = m.job[1].w.m8; tl = tile - m.job[1].pair0; }
    if (m.njobs > 2 && tile >= m.job[2].pair0) { w8 = m.job[2].w.m8; tl = tile - m.job[2].pair0; }
    const uint8_t* wrec = w8 + (size_t)tl * nb * REC;
    const uint8_t* src0 = a.acts + (size_t)tg * NTT * UNIT;   // unit (step s, tile q of the group) at + (s * ntt + q) * UNIT
    const size_t step_stride = (size_t)a.ntt * UNIT;
    // token tiles of the last group behind the image's end: their units are those of the group's first tile again (results dropped)
    const int tq_max = a.ntt - tg * NTT;

    f32x16 acc[NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tt][i] = 0.0f;

    // copy the units of step `st` into stage buffer `buf`: piece p = unit (k, q) part `part`
    auto stage_copy = [&](int st, uint8_t* buf) {
#pragma unroll
        for (int n = 0; n < (SU * PP + kMm8Waves - 1) / kMm8Waves; ++n) {
            const int p = n * kMm8Waves + wv;
            if (p < SU * PP) {
                const int u = p / PP, part = p - u * PP, k = u / NTT, q = u - k * NTT;
                int blk = st * KS + k;
                blk = blk < nb ? blk : nb - 1;
                const uint8_t* s = src0 + (size_t)blk * step_stride + (size_t)(q < tq_max ? q : 0) * UNIT;
                glds16_s(s + part * 1024, (uint32_t)lane * 16u, buf + u * UNIT + part * 1024);
            }
        }
    };
    // the same copy, one piece per call: what mm8_step issues between its sub-blocks (slots 0 .. 8 NTT - 1; the pieces of a wave are spread evenly over them)
    constexpr int NPW = (SU * PP + kMm8Waves - 1) / kMm8Waves, NSLOT = 8 * NTT;
    struct Feeder {
        const uint8_t* src0; uint8_t* buf; size_t step_stride; int st, nb, tq_max, wv, lane;
        DEV void operator()(int slot) const {
#pragma unroll
            for (int n = 0; n < NPW; ++n) {
                if (slot != (n * NSLOT) / NPW) continue;   // (compile-time after unrolling: slot and n are constants at every call site)
                // No branch in here: with one, hipcc sinks the accumulation steps of all eight sub-blocks of the caller behind it and spills every product
                // (the Q8_0 form: 300 spilled registers).  A piece index past the stage's last piece copies the last piece again; the step behind the last
                // one copies the last stage again (into the buffer nobody reads any more).
                int pidx = n * kMm8Waves + wv;
                pidx = pidx < SU * PP ? pidx : SU * PP - 1;
                const int u = pidx / PP, part = pidx - u * PP, k = u / NTT, q = u - k * NTT;
                int blk = st * KS + k;
                blk = blk < nb ? blk : nb - 1;
                const uint8_t* s = src0 + (size_t)blk * step_stride + (size_t)(q < tq_max ? q : 0) * UNIT;
                glds16_s(s + part * 1024, (uint32_t)lane * 16u, buf + u * UNIT + part * 1024);
            }
        }
    };
    stage_copy(0, smem);
    int bcur = ks;
    Mm8W<TYPE> R = mm8_load<TYPE>(wrec + (size_t)(bcur < nb ? bcur : nb - 1) * REC, lane);
    vm_wait<0>();
    mm8_landed<TYPE>(R);
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        uint8_t* cur = smem + (st & 1) * SB;
        // the copy of step st + 1 goes into the buffer step st - 1 read (every wave has passed that step's barrier), piece by piece from inside the step
        const Feeder feed = {src0, smem + ((st + 1) & 1) * SB, step_stride, st + 1 < nsteps ? st + 1 : st, nb, tq_max, wv, lane};
        const int bnext = bcur + KS;
        if (bcur < nb) mm8_step<TYPE, NTT>(R, wrec + (size_t)(bnext < nb ? bnext : nb - 1) * REC, cur + ks * NTT * UNIT, lane, acc, feed);
        else {   // a K-slice without a block in the last step: its share of the copy still goes out
#pragma unroll
            for (int sl = 0; sl < NSLOT; ++sl) feed(sl);
        }
        bcur = bnext;
        vm_wait<0>();      // the copy (and the next weights) have landed ...
        mm8_landed<TYPE>(R);
        __syncthreads();   // ... and every wave knows: the next step reads what this one copied
    }
    // the K-slices' partial results meet in LDS: [wave][tile][register][lane]
    float* red = (float*)smem;
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[((wv * NTT + tt) * 16 + i) * 64 + lane] = acc[tt][i];
    __syncthreads();
    const int pos0 = (m.pos ? *m.pos : 0);
    // a thread = (row tile rwo, token tile tt, register quad i4, lane): four consecutive tokens of one row
    for (int idx = (int)threadIdx.x; idx < RW * NTT * 256; idx += 512) {
        const int lo = idx & 63, i4 = (idx >> 6) & 3, tt = (idx >> 8) % NTT, rwo = (idx >> 8) / NTT;
        int tile_o = bx * RW + rwo;
        const bool tile_ok = tile_o < a.n_tiles;
        tile_o = tile_ok ? tile_o : a.n_tiles - 1;
        int M = m.job[0].w.M, epi = m.job[0].epi, tlo = tile_o;
        if (m.njobs > 1 && tile_o >= m.job[1].pair0) { M = m.job[1].w.M; epi = m.job[1].epi; tlo = tile_o - m.job[1].pair0; }
        if (m.njobs > 2 && tile_o >= m.job[2].pair0) { M = m.job[2].w.M; epi = m.job[2].epi; tlo = tile_o - m.job[2].pair0; }
        float res[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float s = red[((rwo * NTT + tt) * 16 + 4 * i4 + e) * 64 + lo];
#pragma unroll
            for (int k = 1; k < KS; ++k) s += red[(((k * RW + rwo) * NTT + tt) * 16 + 4 * i4 + e) * 64 + lo];
            res[e] = s;
        }
        const int r = lo & 31;
        const int t0 = (tg * NTT + tt) * 32 + 8 * i4 + 4 * (lo >> 5);
        if (m.gateup) {   // fused matrix: lanes r < 16 hold gate row 16 tile + r, lanes r >= 16 the up row of the same index
            const int row = tlo * 16 + (r & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float up = lane_xor16(res[e]);
                if (tile_ok && r < 16 && row < M && t0 + e < a.n_tok) {
                    if (epi == EPI_SILU_MUL) m.out[(size_t)(t0 + e) * a.ld_out + row] = f16_bits_to_f32(m.silu_tab[f32_to_f16_bits(res[e])]) * up;
                }
            }
            continue;
        }
        const int row = tlo * 32 + r;
        const bool row_ok = tile_ok && row < M;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int tok = t0 + e, pos = pos0 + tok;
            const bool own = row_ok && tok < a.n_tok;
            if (epi == EPI_ADD) {
                if (own) m.out[(size_t)tok * a.ld_out + row] = res[e] + m.res[(size_t)tok * a.ld_res + row];
            } else if (epi == EPI_STORE) {
                if (own) m.out[(size_t)tok * a.ld_out + row] = res[e];
            } else if (epi == EPI_GELU) {
                if (own) m.out[(size_t)tok * a.ld_out + row] = f16_bits_to_f32(m.gelu_tab[f32_to_f16_bits(res[e])]);
            } else if (epi == EPI_ADD2) {
                if (own) m.out[(size_t)tok * a.ld_out + row] = (res[e] + m.res[(size_t)tok * a.ld_res + row]) + m.res2[(size_t)tok * a.ld_res + row];
            } else if (epi == EPI_V) {
                if (own) m.vcache[(size_t)row * m.v_stride + pos] = f32_to_f16_bits(res[e]);
            } else {   // RoPE, normal mode (ggml.c:12522-12539, the reference build's fma forms): rows 2i, 2i + 1 are neighbouring lanes
                const float other = lane_xor1(res[e]);
                if (own) {
                    const int ip = (row % m.head_dim) >> 1;
                    const float cs = m.rope_cs[((size_t)pos * (m.head_dim >> 1) + ip) * 2 + 0];
                    const float sn = m.rope_cs[((size_t)pos * (m.head_dim >> 1) + ip) * 2 + 1];
                    const float o = (row & 1) ? fmaf(res[e], cs, other * sn) : fmaf(res[e], cs, -(other * sn));
                    if (epi == EPI_ROPE_Q) m.q_f16[(size_t)tok * a.ld_q + row] = f32_to_f16_bits(o);
                    else m.kcache[kcache_off(pos, row, m.head_dim, m.n_ctx)] = f32_to_f16_bits(o);
                }
            }
        }
    }
}

// ---- prompt attention of the order-free form ------------------------------------------------------------------------------------------------------
// The reference's chain per (head, token) (llama.cpp:2352-2378): K.Q dots of fp16 rows with f32 sums, * 1/sqrt(head_dim) in f32,
// causal mask, softmax = max, table_exp_f16[fp16(s - max)], the sum in double, * (float)(1 / sum) in f32; the probabilities rounded to fp16, V.P dots of
// fp16 rows with f32 sums.  Kept here: every one of those roundings (the fp16 ones and the f32 scale / normalisation), the exact max, the exact double sum
// (fp16 addends are multiples of 2^-24: any order is exact).  Given up: the order of the f32 sums inside the two dot products, which run on
// v_mfma_f32_32x32x16_f16 (products of two halves are exact in f32).
// A workgroup = (head, 32 tokens), eight waves; wave w takes the position tiles w, w + 8, .. of 32 positions up to the tile's last visible position.
//   scores, transposed: D = K_tile (A operand: lane = position) x Q_tile (B operand: lane = token) — lane (token, c) then holds, in register i, the score
//   of position (i & 3) + 8 (i >> 2) + 4 c of the tile: sixteen probabilities of ITS token, which after rounding to fp16 ARE the A operand of the V.P
//   product (lane = token, eight k-slots per matrix instruction = registers 0..7 / 8..15) with no exchange between lanes; the V operand (lane = channel)
//   reads the matching positions as two 8-byte pieces per instruction.
// The probability rows are never stored: pass 1 computes the scores for the row maxima, pass 2 again for the sums, pass 3 again for the V.P product — the
// K.Q product is a quarter of a percent of the matrix cores' time at these sizes, the table look-ups of passes 2 and 3 are what the kernel costs.  No
// context limit (nothing in LDS grows with it).  Masked positions contribute p = 0 as in the reference (expf(-inf) entry); V of positions this request has
// not written yet is not read as numbers (zeroed in the operand: a stale Inf times 0 would poison the sum).
template <int HD>
__global__ void __launch_bounds__(512) attn_mm_kernel(const AttnArgsX a, int n_tok) {
    kernarg_touch<sizeof(AttnArgsX)>();
    constexpr int NS = HD / 16, NCT = HD / 32, NW = 8;
    CT_DYN_SMEM(smem);   // [NW][32] floats (maxima) | [NW][32] doubles (sums) | [NW][16][64] floats (one channel tile's partial results)
    float* s_max = (float*)smem;
    double* s_sum = (double*)(smem + NW * 32 * 4);
    float* s_red = (float*)(smem + NW * 32 * 12);
    const int lane = lane_id(), wv = uniform_int(wave_id()), tk = lane & 31, c = lane >> 5;
    const int h = (int)blockIdx.x, t0 = (int)blockIdx.y * 32;
    const int hk = h / (a.n_head / a.n_head_kv);
    const int pos0 = *a.pos;
    const int tok_ld = t0 + tk < n_tok ? t0 + tk : n_tok - 1;            // (a tile's surplus tokens compute the last token again; nothing of it is stored)
    const int my_last = pos0 + t0 + tk;                                   // the last position this lane's token sees
    const int last = pos0 + (t0 + 31 < n_tok ? t0 + 31 : n_tok - 1);      // the last position any token of the tile sees (written by this chunk's QKV launch)
    const int ntiles = last / 32 + 1;
    u32x4 Q[NS];
    {
        const uint16_t* qrow = a.q_f16 + (size_t)tok_ld * a.q_stride + (size_t)h * HD + 8 * c;
#pragma unroll
        for (int s = 0; s < NS; ++s) Q[s] = ld16(qrow + 16 * s);
    }
    const uint16_t* kb = a.kcache + (size_t)hk * a.n_ctx * HD;
    const float scale = a.kq_scale;
    auto load_k = [&](int pt, u32x4 (&Kf)[NS]) {
        int p = pt * 32 + tk;
        p = p < last ? p : last;                                          // (rows behind the last written one: read the last one, masked below)
        const uint16_t* kr = kb + (size_t)p * HD + 8 * c;
#pragma unroll
        for (int s = 0; s < NS; ++s) Kf[s] = ld16(kr + 16 * s);
    };
    auto scores = [&](int pt, const u32x4 (&Kf)[NS], float (&S)[16]) {
        f32x16 D = mfma_f16_32x32x16(Kf[0], Q[0]);
#pragma unroll
        for (int s = 1; s < NS; ++s) D = mfma_f16_32x32x16_acc(Kf[s], Q[s], D);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int p = pt * 32 + (i & 3) + 8 * (i >> 2) + 4 * c;
            S[i] = p <= my_last ? D[i] * scale : -INFINITY;
        }
    };
    // pass 1: the row maxima
    float mx = -INFINITY;
    {
        u32x4 Kc[NS], Kn[NS];
        if (wv < ntiles) load_k(wv, Kc);
        for (int pt = wv; pt < ntiles; pt += NW) {
            if (pt + NW < ntiles) load_k(pt + NW, Kn);
            float S[16];
            scores(pt, Kc, S);
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, S[i]);
#pragma unroll
            for (int s = 0; s < NS; ++s) Kc[s] = Kn[s];
        }
    }
    mx = fmaxf(mx, lane_xor32(mx));
    if (c == 0) s_max[wv * 32 + tk] = mx;
    __syncthreads();
    mx = s_max[tk];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, s_max[w * 32 + tk]);
    // pass 2: the sums of the exponentials (double: exact in any order)
    double sum = 0.0;
    {
        u32x4 Kc[NS], Kn[NS];
        if (wv < ntiles) load_k(wv, Kc);
        for (int pt = wv; pt < ntiles; pt += NW) {
            if (pt + NW < ntiles) load_k(pt + NW, Kn);
            float S[16];
            scores(pt, Kc, S);
            float e[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) e[i] = f16_bits_to_f32(a.exp_tab[f32_to_f16_bits(S[i] - mx)]);   // (masked: fp16(-inf) -> the table's 0)
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += (double)e[i];
#pragma unroll
            for (int s = 0; s < NS; ++s) Kc[s] = Kn[s];
        }
    }
    sum += lane_xor32(sum);
    if (c == 0) s_sum[wv * 32 + tk] = sum;
    __syncthreads();
    sum = s_sum[tk];
#pragma unroll
    for (int w = 1; w < NW; ++w) sum += s_sum[w * 32 + tk];
    const float inv = (float)(1.0 / sum);
    // pass 3: probabilities as fp16, V.P on the matrix cores
    f32x16 O[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) O[ct][i] = 0.0f;
    {
        const uint16_t* vb = a.vcache + ((size_t)hk * HD + tk) * a.v_stride + 4 * c;   // lane (channel tk of a tile, c): positions 4 c + {0..3, 8..11, 16..19, 24..27}
        u32x4 Kc[NS], Kn[NS];
        if (wv < ntiles) load_k(wv, Kc);
        for (int pt = wv; pt < ntiles; pt += NW) {
            if (pt + NW < ntiles) load_k(pt + NW, Kn);
            u32x2 V[NCT][4];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) V[ct][g] = ld8(vb + (size_t)ct * 32 * a.v_stride + pt * 32 + 8 * g);
            if (pt * 32 + 31 > last) {   // the tile holds positions nobody has written in this request: zero them in the operand (wave-uniform branch)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nv = last - (pt * 32 + 8 * g + 4 * c) + 1;   // valid halves of this piece
                    const uint32_t m0 = nv >= 2 ? 0xFFFFFFFFu : (nv == 1 ? 0xFFFFu : 0u), m1 = nv >= 4 ? 0xFFFFFFFFu : (nv == 3 ? 0xFFFFu : 0u);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) { V[ct][g][0] &= m0; V[ct][g][1] &= m1; }
                }
            }
            float S[16];
            scores(pt, Kc, S);
            uint32_t P[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float e0 = f16_bits_to_f32(a.exp_tab[f32_to_f16_bits(S[2 * i] - mx)]);
                const float e1 = f16_bits_to_f32(a.exp_tab[f32_to_f16_bits(S[2 * i + 1] - mx)]);
                P[i] = (uint32_t)f32_to_f16_bits(e0 * inv) | ((uint32_t)f32_to_f16_bits(e1 * inv) << 16);
            }
            const u32x4 PA0 = {P[0], P[1], P[2], P[3]}, PA1 = {P[4], P[5], P[6], P[7]};
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const u32x4 VB0 = {V[ct][0][0], V[ct][0][1], V[ct][1][0], V[ct][1][1]}, VB1 = {V[ct][2][0], V[ct][2][1], V[ct][3][0], V[ct][3][1]};
                O[ct] = mfma_f16_32x32x16_acc(PA0, VB0, O[ct]);
                O[ct] = mfma_f16_32x32x16_acc(PA1, VB1, O[ct]);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) Kc[s] = Kn[s];
        }
    }
    // the waves' partial results meet in LDS, one channel tile at a time: lane (channel, c) register i = token (i & 3) + 8 (i >> 2) + 4 c
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) s_red[(wv * 16 + i) * 64 + lane] = O[ct][i];
        __syncthreads();
        for (int idx = (int)threadIdx.x; idx < 16 * 64; idx += 512) {
            const int i = idx >> 6, ln = idx & 63;
            float r = s_red[idx];
#pragma unroll
            for (int w = 1; w < NW; ++w) r += s_red[w * 16 * 64 + idx];
            const int tok = t0 + (i & 3) + 8 * (i >> 2) + 4 * (ln >> 5);
            if (tok < n_tok) a.out[(size_t)tok * a.out_stride + (size_t)h * HD + ct * 32 + (ln & 31)] = r;
        }
    }
}
