// Prompt chunks, K-quant weights on the int8 matrix cores — still bit-identical to the reference CPU build.
// (Q4_K explained here; Q5_K and Q6_K differ in how the scale is split — see pfm_item_q45 and the Q6_K section.)
//
// What has to be reproduced per (row, token) and 256-block is the reference's eight int32 lane sums
//     sumi[l] = sum_{s<8} sc_s * sum_{e<4} q_s[4l+e] * a_s[4l+e]                  (k_quants.c:2651-2720; kernels_exact.h)
// followed by ONE f32 fma per lane and block.  The integer part is a matrix product with K = 32 per AVX lane l
// (k = (s, e)), except that the 6-bit sub-block scale sits between the 4-bit weight and the product.  Splitting the
// scale as sc = 8*hi3 + lo3 makes both halves fit the matrix core's int8 operands (q * 7 <= 105):
//     sumi[l] = 8 * sum_k a_k * (q_k * hi3_s(k)) + sum_k a_k * (q_k * lo3_s(k))
// i.e. two v_mfma_i32_16x16x32_i8 per l, exact in int32, combined as (hi << 3) + lo.
// The scaled weight bytes come from ONE packed 16-bit multiply per dword (a byte times 7 cannot carry into its neighbour).
// The f32 side — acc[l] = fma(y.d * fp16(x.d), (float)sumi[l], acc[l]), the four min-term accumulators, hsum_float_8 —
// is the decode kernels', now entirely in-lane (no cross-lane reduction is left: the matrix core did the sums).
//
// Shapes: a wave owns 16 weight rows (two 8-row tiles of the TILE8S layout) x 16 tokens.  The MFMA is issued
// transposed, D[token][row] = A[token][k] * B[k][row], so that a lane's results are four tokens of ITS row (lane & 15):
// everything per-row (d, dmin, scales, mins) is already in the lane that loaded the row's header.
//   weights   lane (row r16 = lane & 15, q = lane >> 4) loads the row's header (16 B) and the 32 bytes of nibble units
//             2q, 2q+1: dword l = elements 4l..4l+3 of sub-blocks 2q (low nibbles) and 2q+1 (high nibbles)
//   tokens    lane (token n = lane & 15, q) reads words [16q, 16q+16) of the token's block: word l / 8+l = the same
//             elements of sub-blocks 2q / 2q+1; operand k-order {sub-block 2q, sub-block 2q+1} on both sides
//   results   lane (r16, q), register j: token 4q + j
// Per block and wave (256 (row, token) pairs): 16 MFMA + ~250 VALU, against 8 x ~55 VALU for 64 pairs in kernels_pf.h.
#pragma once
#include "kernels_pf.h"

// All blocks of one 16-row item against 16 token images (K <= 8192; pfm_item_q45_t8 below is the form for fewer images).
// Q5_K (TYPE == GT_Q5_K) differs in three places: the fifth bit comes from the row's 32 qh bytes (bit 2q / 2q+1 of byte e =
// sub-block 2q / 2q+1), q5 * 7 would not fit int8 so the scale goes in three 2-bit digits (q5 * 3 <= 93, three MFMAs per
// l), and the min term is one scalar: summs = fma(-y.d * dmin, (float)(prod[0] + .. + prod[3]), summs) (k_quants.c:3183-3262).
template <int TYPE, int TOK>
DEV void pfm_item_q45(const uint8_t* __restrict__ w0, int item, int n_tiles, int nb, const int* __restrict__ lds, int act_words,
                      int K, int lane, float (&res)[4]) {
    constexpr bool Q5 = TYPE == GT_Q5_K;
    constexpr uint32_t REC = Q5 ? 1408 : 1152, QS0 = Q5 ? 384 : 128;
    const int r16 = lane & 15, q = lane >> 4;
    int tile = 2 * item + (r16 >> 3);
    tile = tile < n_tiles ? tile : n_tiles - 1;
    const uint8_t* base = w0 + (size_t)tile * nb * REC + (r16 & 7) * 16;
    const uint32_t qoff = QS0 - (uint32_t)(r16 & 7) * 16u + (uint32_t)(r16 & 7) * 128u + (uint32_t)q * 32u;
    const uint32_t hoff = 128u - (uint32_t)(r16 & 7) * 16u + (uint32_t)(r16 & 7) * 32u;   // Q5_K: the row's qh bytes
    const int nq = K >> 2, nbk = K >> 8;
    const int* imgA = lds + ((lane & 15) & (TOK - 1)) * act_words + 16 * q;
    const int* imgT[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) imgT[j] = lds + ((4 * q + j) & (TOK - 1)) * act_words + nq;
    float acc[4][8], accm[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int l = 0; l < 8; ++l) acc[j][l] = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) accm[j][t] = 0.0f;
    }
    // Two weight records in flight per lane, as a ring indexed by the unrolled loop position: the slot is refilled with
    // block b+2 right after its registers are read.  (Written as "load block b+1 at the top of iteration b" hipcc turns the
    // loop into "load block b, wait, use it": the memory latency of every block lands on the critical path.)
    constexpr int PF = 2;
    u32x4 rh[PF], ra[PF], rb[PF], rq0[PF], rq1[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const uint8_t* p = base + (size_t)(u < nb ? u : nb - 1) * REC;
        rh[u] = ld_stream16(p); ra[u] = ld_stream16(p + qoff); rb[u] = ld_stream16(p + qoff + 16);
        if constexpr (Q5) { rq0[u] = ld_stream16(p + hoff); rq1[u] = ld_stream16(p + hoff + 16); }
    }
    for (int b0 = 0; b0 < nb; b0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int b = b0 + u;
        const u32x4 H = rh[u], QA = ra[u], QB = rb[u];
        u32x4 QH0 = H, QH1 = H;
        if constexpr (Q5) { QH0 = rq0[u]; QH1 = rq1[u]; }
        {
            const uint8_t* p = base + (size_t)(b + PF < nb ? b + PF : nb - 1) * REC;
            rh[u] = ld_stream16(p); ra[u] = ld_stream16(p + qoff); rb[u] = ld_stream16(p + qoff + 16);
            if constexpr (Q5) { rq0[u] = ld_stream16(p + hoff); rq1[u] = ld_stream16(p + hoff + 16); }
        }
        if (b >= nb) continue;
        // the four 24-bit scale groups {sc[2c], sc[2c+1], m[2c], m[2c+1]} x 6 bit of this row (engine.cc:upload_matrix)
        const uint32_t x0 = H[1], x1 = alignbit32(H[2], H[1], 24), x2 = alignbit32(H[3], H[2], 16), x3 = H[3] >> 8;
        const uint32_t xq = q == 0 ? x0 : (q == 1 ? x1 : (q == 2 ? x2 : x3));
        const uint32_t sc_lo = xq & 63u, sc_hi = bfe32(xq, 6, 6);
        // scale digits, low to high: Q4_K 3 + 3 bits, Q5_K 2 + 2 + 2 bits
        constexpr int ND = Q5 ? 3 : 2, DB = Q5 ? 2 : 3;
        uint32_t sd[ND], td[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            sd[i] = ((sc_lo >> (DB * i)) & ((1u << DB) - 1u)) * 0x00010001u;
            td[i] = ((sc_hi >> (DB * i)) & ((1u << DB) - 1u)) * 0x00010001u;
        }
        const float dw = f16_bits_to_f32((uint16_t)(H[0] & 0xFFFF));
        const float dmw = f16_bits_to_f32((uint16_t)(H[0] >> 16));
        float D[4], DM[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float yd = bits_to_f32((uint32_t)imgT[j][b]);
            D[j] = yd * dw;
            DM[j] = -yd * dmw;
        }
        const u32x4 a0 = *(const u32x4*)(imgA + b * 64), a1 = *(const u32x4*)(imgA + b * 64 + 4);
        const u32x4 a2 = *(const u32x4*)(imgA + b * 64 + 8), a3 = *(const u32x4*)(imgA + b * 64 + 12);
        u32x4 s0[4], s1[4];   // q8s[0..7] of this lane's four tokens: requested together, consumed after the matrix work
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int* sb = imgT[j] + nbk + b * 8;
            s0[j] = *(const u32x4*)sb;
            s1[j] = *(const u32x4*)(sb + 4);
        }
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            const uint32_t qs = l < 4 ? QA[l & 3] : QB[l & 3];
            uint32_t wlo = qs & 0x0F0F0F0Fu, whi = (qs >> 4) & 0x0F0F0F0Fu;
            if constexpr (Q5) {
                const uint32_t qh = l < 4 ? QH0[l & 3] : QH1[l & 3];
                wlo |= ((qh >> (2 * q)) & 0x01010101u) << 4;
                whi |= ((qh >> (2 * q + 1)) & 0x01010101u) << 4;
            }
            const uint32_t alo = l < 4 ? a0[l & 3] : a1[l & 3], ahi = l < 4 ? a2[l & 3] : a3[l & 3];
            const uint64_t A = (uint64_t)alo | ((uint64_t)ahi << 32);
            if constexpr (ND == 2) {   // Q4_K: two independent MFMAs (they pipeline), combined (hi << 3) + lo
                const i32x4 zero = {0, 0, 0, 0};
                i32x4 pd[ND];
#pragma unroll
                for (int i = 0; i < ND; ++i) pd[i] = mfma_i8_16x16x32(A, (uint64_t)pk_mul_u16(wlo, sd[i]) | ((uint64_t)pk_mul_u16(whi, td[i]) << 32), zero);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j][l] = fmaf(D[j], (float)((int)((uint32_t)pd[1][j] << DB) + pd[0][j]), acc[j][l]);
            } else {                   // Q5_K: chained through the accumulator (three independent results cost too many registers)
                i32x4 c = {0, 0, 0, 0};
#pragma unroll
                for (int i = ND - 1; i >= 0; --i) {
                    c = mfma_i8_16x16x32(A, (uint64_t)pk_mul_u16(wlo, sd[i]) | ((uint64_t)pk_mul_u16(whi, td[i]) << 32), c);
                    if (i > 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) c[j] = (int)((uint32_t)c[j] << DB);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j][l] = fmaf(D[j], (float)c[j], acc[j][l]);
            }
        }
        // min term: prod[t] = m[2t] * q8s[2t] + m[2t+1] * q8s[2t+1];  Q4_K: acc_m[t] = fma(-y.d * dmin, (float)prod[t], acc_m[t]),
        // Q5_K: summs = fma(-y.d * dmin, (float)(prod[0] + prod[1] + prod[2] + prod[3]), summs)
        const int m0 = (int)bfe32(x0, 12, 6), m1 = (int)bfe32(x0, 18, 6), m2 = (int)bfe32(x1, 12, 6), m3 = (int)bfe32(x1, 18, 6);
        const int m4 = (int)bfe32(x2, 12, 6), m5 = (int)bfe32(x2, 18, 6), m6 = (int)bfe32(x3, 12, 6), m7 = (int)bfe32(x3, 18, 6);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p0 = mul24(m0, (int)s0[j][0]) + mul24(m1, (int)s0[j][1]), p1 = mul24(m2, (int)s0[j][2]) + mul24(m3, (int)s0[j][3]);
            const int p2 = mul24(m4, (int)s1[j][0]) + mul24(m5, (int)s1[j][1]), p3 = mul24(m6, (int)s1[j][2]) + mul24(m7, (int)s1[j][3]);
            if constexpr (Q5) {
                accm[j][0] = fmaf(DM[j], (float)((p0 + p1) + (p2 + p3)), accm[j][0]);
            } else {
                accm[j][0] = fmaf(DM[j], (float)p0, accm[j][0]);
                accm[j][1] = fmaf(DM[j], (float)p1, accm[j][1]);
                accm[j][2] = fmaf(DM[j], (float)p2, accm[j][2]);
                accm[j][3] = fmaf(DM[j], (float)p3, accm[j][3]);
            }
        }
    }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // hsum_float_8 (k_quants.c:90-97) and the min-term tree, in-lane
        const float tot = ((acc[j][0] + acc[j][4]) + (acc[j][2] + acc[j][6])) + ((acc[j][1] + acc[j][5]) + (acc[j][3] + acc[j][7]));
        const float am = Q5 ? accm[j][0] : (accm[j][0] + accm[j][2]) + (accm[j][1] + accm[j][3]);
        res[j] = tot + am;
    }
}

// the lane holding the other AVX-lane half of the same (row, token) in the 8- / 4-token forms below
template <int NT, class T> DEV T pf_partner(T v) {
    if constexpr (NT == 8) return lane_xor32(v);
    else return lane_xor16(v);
}

// The same for K > 8192, where LDS holds 8 token images only.  Running the 16-token form on 8 tokens would compute every
// result twice; instead the 16 token slots of the matrix product become (token, half of the AVX lanes): slot n < 8 is
// token n with lanes l' = 0..3, slot n >= 8 is token n - 8 with lanes l' + 4.  One v_mfma_i32_16x16x64_i8 carries both
// K sets — k = (half, sub-block, e) — and a slot's A operand is zero in the other half's bytes, so
//     D[(token, half)][row] = sumi[l' + 4 * half]      for l' = 0..3: 8 MFMA per block, 16 accumulators per lane.
// The min term splits the same way (half 0: acc_m[0..1], half 1: acc_m[2..3]); the two halves of a (row, token) sit in
// lanes 32 apart and meet once per item for the final hsum_float_8 tree.
// NT = 8 token images (8192 < K <= 12288), or 4 (K up to 32768: the 70B-class ffn_down) — then only slots 0..7 are distinct
// (token = slot & 3, half = bit 2 of the slot), the partner half sits 16 lanes away and lanes 32..63 repeat lanes 0..31.
template <int TYPE, int NT>
DEV void pfm_item_q45_t8(const uint8_t* __restrict__ w0, int item, int n_tiles, int nb, const int* __restrict__ lds, int act_words,
                         int K, int lane, float (&res)[4]) {
    constexpr bool Q5 = TYPE == GT_Q5_K;
    constexpr uint32_t REC = Q5 ? 1408 : 1152, QS0 = Q5 ? 384 : 128;
    const int r16 = lane & 15, q = lane >> 4, half = NT == 8 ? q >> 1 : q & 1;
    int tile = 2 * item + (r16 >> 3);
    tile = tile < n_tiles ? tile : n_tiles - 1;
    const uint8_t* base = w0 + (size_t)tile * nb * REC + (r16 & 7) * 16;
    const uint32_t qoff = QS0 - (uint32_t)(r16 & 7) * 16u + (uint32_t)(r16 & 7) * 128u + (uint32_t)q * 32u;
    const uint32_t hoff = 128u - (uint32_t)(r16 & 7) * 16u + (uint32_t)(r16 & 7) * 32u;
    const int nq = K >> 2, nbk = K >> 8;
    const bool a_hi_half = (lane & NT) != 0;   // A side: slot n = lane & 15 -> token n & (NT - 1), half = the next bit
    const int* imgA = lds + (lane & (NT - 1)) * act_words + 16 * q + (a_hi_half ? 4 : 0);
    const int* imgT[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) imgT[j] = lds + (NT == 8 ? 4 * (q & 1) + j : j) * act_words + nq;
    float acc[4][4], accm[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int l = 0; l < 4; ++l) acc[j][l] = 0.0f;
        accm[j][0] = accm[j][1] = 0.0f;
    }
    constexpr int PF = 2;
    u32x4 rh[PF], ra[PF], rb[PF], rq0[PF], rq1[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const uint8_t* p = base + (size_t)(u < nb ? u : nb - 1) * REC;
        rh[u] = ld_stream16(p); ra[u] = ld_stream16(p + qoff); rb[u] = ld_stream16(p + qoff + 16);
        if constexpr (Q5) { rq0[u] = ld_stream16(p + hoff); rq1[u] = ld_stream16(p + hoff + 16); }
    }
    for (int b0 = 0; b0 < nb; b0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int b = b0 + u;
        const u32x4 H = rh[u], QA = ra[u], QB = rb[u];
        u32x4 QH0 = H, QH1 = H;
        if constexpr (Q5) { QH0 = rq0[u]; QH1 = rq1[u]; }
        {
            const uint8_t* p = base + (size_t)(b + PF < nb ? b + PF : nb - 1) * REC;
            rh[u] = ld_stream16(p); ra[u] = ld_stream16(p + qoff); rb[u] = ld_stream16(p + qoff + 16);
            if constexpr (Q5) { rq0[u] = ld_stream16(p + hoff); rq1[u] = ld_stream16(p + hoff + 16); }
        }
        if (b >= nb) continue;
        const uint32_t x0 = H[1], x1 = alignbit32(H[2], H[1], 24), x2 = alignbit32(H[3], H[2], 16), x3 = H[3] >> 8;
        const uint32_t xq = q == 0 ? x0 : (q == 1 ? x1 : (q == 2 ? x2 : x3));
        const uint32_t sc_lo = xq & 63u, sc_hi = bfe32(xq, 6, 6);
        constexpr int ND = Q5 ? 3 : 2, DB = Q5 ? 2 : 3;
        uint32_t sd[ND], td[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            sd[i] = ((sc_lo >> (DB * i)) & ((1u << DB) - 1u)) * 0x00010001u;
            td[i] = ((sc_hi >> (DB * i)) & ((1u << DB) - 1u)) * 0x00010001u;
        }
        const float dw = f16_bits_to_f32((uint16_t)(H[0] & 0xFFFF));
        const float dmw = f16_bits_to_f32((uint16_t)(H[0] >> 16));
        float D[4], DM[4];
        u32x4 sbv[4], sbo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float yd = bits_to_f32((uint32_t)imgT[j][b]);
            D[j] = yd * dw;
            DM[j] = -yd * dmw;
            sbv[j] = *(const u32x4*)(imgT[j] + nbk + b * 8 + 4 * half);   // q8s[4 * half .. + 3]
            if constexpr (Q5) sbo[j] = *(const u32x4*)(imgT[j] + nbk + b * 8 + 4 * (1 - half));   // the other four: Q5_K sums all eight
        }
        const u32x4 alo = *(const u32x4*)(imgA + b * 64), ahi = *(const u32x4*)(imgA + b * 64 + 8);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            uint32_t w0lo = QA[l] & 0x0F0F0F0Fu, w0hi = (QA[l] >> 4) & 0x0F0F0F0Fu;   // AVX lane l
            uint32_t w1lo = QB[l] & 0x0F0F0F0Fu, w1hi = (QB[l] >> 4) & 0x0F0F0F0Fu;   // AVX lane l + 4
            if constexpr (Q5) {
                w0lo |= ((QH0[l] >> (2 * q)) & 0x01010101u) << 4; w0hi |= ((QH0[l] >> (2 * q + 1)) & 0x01010101u) << 4;
                w1lo |= ((QH1[l] >> (2 * q)) & 0x01010101u) << 4; w1hi |= ((QH1[l] >> (2 * q + 1)) & 0x01010101u) << 4;
            }
            const uint32_t al = alo[l], ah = ahi[l];
            const u32x4 A = a_hi_half ? u32x4{0u, 0u, al, ah} : u32x4{al, ah, 0u, 0u};
            if constexpr (ND == 2) {   // Q4_K: two independent MFMAs (they pipeline), combined (hi << 3) + lo
                const i32x4 zero = {0, 0, 0, 0};
                i32x4 pd[ND];
#pragma unroll
                for (int i = 0; i < ND; ++i) pd[i] = mfma_i8_16x16x64(A, u32x4{pk_mul_u16(w0lo, sd[i]), pk_mul_u16(w0hi, td[i]), pk_mul_u16(w1lo, sd[i]), pk_mul_u16(w1hi, td[i])}, zero);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j][l] = fmaf(D[j], (float)((int)((uint32_t)pd[1][j] << DB) + pd[0][j]), acc[j][l]);
            } else {                   // Q5_K: chained through the accumulator (three independent results cost too many registers)
                i32x4 c = {0, 0, 0, 0};
#pragma unroll
                for (int i = ND - 1; i >= 0; --i) {
                    c = mfma_i8_16x16x64(A, u32x4{pk_mul_u16(w0lo, sd[i]), pk_mul_u16(w0hi, td[i]), pk_mul_u16(w1lo, sd[i]), pk_mul_u16(w1hi, td[i])}, c);
                    if (i > 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) c[j] = (int)((uint32_t)c[j] << DB);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j][l] = fmaf(D[j], (float)c[j], acc[j][l]);
            }
        }
        const uint32_t xa = half ? x2 : x0, xb = half ? x3 : x1;   // scale groups 2 * half, 2 * half + 1
        const int ma0 = (int)bfe32(xa, 12, 6), ma1 = (int)bfe32(xa, 18, 6), mb0 = (int)bfe32(xb, 12, 6), mb1 = (int)bfe32(xb, 18, 6);
        const uint32_t xc = half ? x0 : x2, xd = half ? x1 : x3;   // Q5_K: the other half's groups
        const int mc0 = (int)bfe32(xc, 12, 6), mc1 = (int)bfe32(xc, 18, 6), md0 = (int)bfe32(xd, 12, 6), md1 = (int)bfe32(xd, 18, 6);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pa = mul24(ma0, (int)sbv[j][0]) + mul24(ma1, (int)sbv[j][1]), pb = mul24(mb0, (int)sbv[j][2]) + mul24(mb1, (int)sbv[j][3]);
            if constexpr (Q5) {   // one scalar term: both halves of the lane pair compute the whole sum (integers: any order)
                const int pc = mul24(mc0, (int)sbo[j][0]) + mul24(mc1, (int)sbo[j][1]), pd = mul24(md0, (int)sbo[j][2]) + mul24(md1, (int)sbo[j][3]);
                accm[j][0] = fmaf(DM[j], (float)((pa + pb) + (pc + pd)), accm[j][0]);
            } else {
                accm[j][0] = fmaf(DM[j], (float)pa, accm[j][0]);
                accm[j][1] = fmaf(DM[j], (float)pb, accm[j][1]);
            }
        }
    }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // x_l + x_{l+4}: one operand is this lane's, the other the partner half's (fp addition commutes bit for bit)
        const float p0 = acc[j][0] + pf_partner<NT>(acc[j][0]), p1 = acc[j][1] + pf_partner<NT>(acc[j][1]);
        const float p2 = acc[j][2] + pf_partner<NT>(acc[j][2]), p3 = acc[j][3] + pf_partner<NT>(acc[j][3]);
        const float tot = (p0 + p2) + (p1 + p3);
        const float am = Q5 ? accm[j][0] : (accm[j][0] + pf_partner<NT>(accm[j][0])) + (accm[j][1] + pf_partner<NT>(accm[j][1]));   // Q4_K: (m0 + m2) + (m1 + m3)
        res[j] = tot + am;
    }
}

// ---- Q6_K --------------------------------------------------------------------------------------------------------------
// Reference (k_quants.c:3800-3872): sumi[l] = sum over the eight 32-element vectors v of
//     scales[2v + (l >> 2)] * sum_{e<4} (q6_v[4l+e] - 32) * a_v[4l+e],        scales int8, q6 in [0, 63]
// — the Q4_K shape with a signed 8-bit scale on a signed 6-bit weight.  With u = scale + 128 = u0 + 4 u1 + 16 u2 + 64 u3
// (2-bit digits) every operand byte (q6 - 32) * u_i lies in [-96, 93], and
//     sumi[l] = P0 + 4 P1 + 16 P2 + 64 P3 + 128 Pn,   P_i = sum a * (q6 - 32) * u_i,   Pn = sum a * (32 - q6)
// is five chained MFMAs (shift the accumulator between them).  A signed byte product without a packed byte multiplier:
// q6 * u_i by the packed 16-bit multiply (<= 189, no carry), plus (128 - 32 u_i) per byte (still < 256), xor 0x80.
struct Q6Scale {            // one 16-element scale, prepared for four weight dwords: multiplier and byte offset per digit
    uint32_t cu[4], off[4];
};
DEV Q6Scale q6_scale(uint32_t sc_byte) {
    const uint32_t u = sc_byte ^ 0x80u;
    Q6Scale S;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        S.cu[i] = ((u >> (2 * i)) & 3u) * 0x00010001u;
        S.off[i] = 0x80808080u - pk_mul_u16(0x20202020u, S.cu[i]);
    }
    return S;
}
DEV uint32_t q6_piece(uint32_t q6, const Q6Scale& S, int i) { return (pk_mul_u16(q6, S.cu[i]) + S.off[i]) ^ 0x80808080u; }
DEV uint32_t q6_neg(uint32_t q6) { return (0xA0A0A0A0u - q6) ^ 0x80808080u; }   // bytes 32 - q6

struct Q6Rec { u32x4 sc, ql0, ql1, qh0, qh1; uint32_t d; };
// lane (row r16 & 7 of its tile, q): scales of the row, the 32 bytes of ql units 2q, 2q+1 and of qh units (q >> 1, 0..1)
DEV Q6Rec q6_load(const uint8_t* rec, int r, int q) {
    Q6Rec R;
    R.d = *(const uint16_t*)(rec + r * 2);
    R.sc = ld_stream16(rec + 16 + r * 16);
    R.qh0 = ld_stream16(rec + 144 + r * 64 + (q >> 1) * 32);
    R.qh1 = ld_stream16(rec + 144 + r * 64 + (q >> 1) * 32 + 16);
    R.ql0 = ld_stream16(rec + 656 + r * 128 + q * 32);
    R.ql1 = ld_stream16(rec + 656 + r * 128 + q * 32 + 16);
    return R;
}
// the two 6-bit weight dwords of AVX lane l: vector 4n + kq (low nibbles) and vector 4n + 2 + kq (high nibbles)
DEV void q6_unpack(const Q6Rec& R, int l, int kq, uint32_t& lo6, uint32_t& hi6) {
    const uint32_t qlw = l < 4 ? R.ql0[l & 3] : R.ql1[l & 3], qhw = l < 4 ? R.qh0[l & 3] : R.qh1[l & 3];
    lo6 = (qlw & 0x0F0F0F0Fu) | (((qhw >> (2 * kq)) & 0x03030303u) << 4);
    hi6 = ((qlw >> 4) & 0x0F0F0F0Fu) | (((qhw >> (4 + 2 * kq)) & 0x03030303u) << 4);
}

template <int TOK>
DEV void pfm_item_q6k(const uint8_t* __restrict__ w0, int item, int n_tiles, int nb, const int* __restrict__ lds, int act_words,
                      int K, int lane, float (&res)[4]) {
    constexpr uint32_t REC = 1680;
    const int r16 = lane & 15, q = lane >> 4, n = q >> 1, kq = q & 1;
    int tile = 2 * item + (r16 >> 3);
    tile = tile < n_tiles ? tile : n_tiles - 1;
    const uint8_t* base = w0 + (size_t)tile * nb * REC;
    const int nq = K >> 2;
    const int* imgA = lds + ((lane & 15) & (TOK - 1)) * act_words + 32 * n + 8 * kq;   // words l / 16 + l of the block
    const int* imgT[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) imgT[j] = lds + ((4 * q + j) & (TOK - 1)) * act_words + nq;
    float acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int l = 0; l < 8; ++l) acc[j][l] = 0.0f;
    }
    constexpr int PF = 2;
    Q6Rec ring[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) ring[u] = q6_load(base + (size_t)(u < nb ? u : nb - 1) * REC, r16 & 7, q);
    for (int b0 = 0; b0 < nb; b0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int b = b0 + u;
        const Q6Rec R = ring[u];
        ring[u] = q6_load(base + (size_t)(b + PF < nb ? b + PF : nb - 1) * REC, r16 & 7, q);
        if (b >= nb) continue;
        const float dw = f16_bits_to_f32((uint16_t)(R.d & 0xFFFF));
        float D[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) D[j] = bits_to_f32((uint32_t)imgT[j][b]) * dw;
        const uint32_t w_lo = n ? R.sc[2] : R.sc[0], w_hi = n ? R.sc[3] : R.sc[1];   // scales 8n.., 8n+4..
        const u32x4 a0 = *(const u32x4*)(imgA + b * 64), a1 = *(const u32x4*)(imgA + b * 64 + 4);
        const u32x4 a2 = *(const u32x4*)(imgA + b * 64 + 16), a3 = *(const u32x4*)(imgA + b * 64 + 20);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const Q6Scale SL = q6_scale((w_lo >> (8 * (2 * kq + h))) & 0xFFu), SH = q6_scale((w_hi >> (8 * (2 * kq + h))) & 0xFFu);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int l = 4 * h + k;
                uint32_t lo6, hi6;
                q6_unpack(R, l, kq, lo6, hi6);
                const uint64_t A = (uint64_t)(h ? a1[k] : a0[k]) | ((uint64_t)(h ? a3[k] : a2[k]) << 32);
                i32x4 c = {0, 0, 0, 0};
                c = mfma_i8_16x16x32(A, (uint64_t)q6_neg(lo6) | ((uint64_t)q6_neg(hi6) << 32), c);
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j] = (int)((uint32_t)c[j] << 1);
#pragma unroll
                for (int i = 3; i >= 0; --i) {
                    c = mfma_i8_16x16x32(A, (uint64_t)q6_piece(lo6, SL, i) | ((uint64_t)q6_piece(hi6, SH, i) << 32), c);
                    if (i > 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) c[j] = (int)((uint32_t)c[j] << 2);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j][l] = fmaf(D[j], (float)c[j], acc[j][l]);
            }
        }
    }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        res[j] = ((acc[j][0] + acc[j][4]) + (acc[j][2] + acc[j][6])) + ((acc[j][1] + acc[j][5]) + (acc[j][3] + acc[j][7]));
}

// K > 8192 form (see pfm_item_q45_t8): token slots = (token, AVX-lane half), one 16x16x64 MFMA carries lanes l' and l' + 4.
template <int NT>
DEV void pfm_item_q6k_t8(const uint8_t* __restrict__ w0, int item, int n_tiles, int nb, const int* __restrict__ lds, int act_words,
                         int K, int lane, float (&res)[4]) {
    constexpr uint32_t REC = 1680;
    const int r16 = lane & 15, q = lane >> 4, n = q >> 1, kq = q & 1;
    int tile = 2 * item + (r16 >> 3);
    tile = tile < n_tiles ? tile : n_tiles - 1;
    const uint8_t* base = w0 + (size_t)tile * nb * REC;
    const int nq = K >> 2;
    const bool a_hi_half = (lane & NT) != 0;
    const int* imgA = lds + (lane & (NT - 1)) * act_words + 32 * n + 8 * kq + (a_hi_half ? 4 : 0);
    const int* imgT[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) imgT[j] = lds + (NT == 8 ? 4 * (q & 1) + j : j) * act_words + nq;
    float acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int l = 0; l < 4; ++l) acc[j][l] = 0.0f;
    }
    constexpr int PF = 2;
    Q6Rec ring[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) ring[u] = q6_load(base + (size_t)(u < nb ? u : nb - 1) * REC, r16 & 7, q);
    for (int b0 = 0; b0 < nb; b0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int b = b0 + u;
        const Q6Rec R = ring[u];
        ring[u] = q6_load(base + (size_t)(b + PF < nb ? b + PF : nb - 1) * REC, r16 & 7, q);
        if (b >= nb) continue;
        const float dw = f16_bits_to_f32((uint16_t)(R.d & 0xFFFF));
        float D[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) D[j] = bits_to_f32((uint32_t)imgT[j][b]) * dw;
        const uint32_t w_lo = n ? R.sc[2] : R.sc[0], w_hi = n ? R.sc[3] : R.sc[1];
        const Q6Scale SL0 = q6_scale((w_lo >> (16 * kq)) & 0xFFu), SH0 = q6_scale((w_hi >> (16 * kq)) & 0xFFu);           // lanes 0..3
        const Q6Scale SL1 = q6_scale((w_lo >> (16 * kq + 8)) & 0xFFu), SH1 = q6_scale((w_hi >> (16 * kq + 8)) & 0xFFu);   // lanes 4..7
        const u32x4 alo = *(const u32x4*)(imgA + b * 64), ahi = *(const u32x4*)(imgA + b * 64 + 16);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            uint32_t lo0, hi0, lo1, hi1;
            q6_unpack(R, l, kq, lo0, hi0);
            q6_unpack(R, l + 4, kq, lo1, hi1);
            const uint32_t al = alo[l], ah = ahi[l];
            const u32x4 A = a_hi_half ? u32x4{0u, 0u, al, ah} : u32x4{al, ah, 0u, 0u};
            i32x4 c = {0, 0, 0, 0};
            c = mfma_i8_16x16x64(A, u32x4{q6_neg(lo0), q6_neg(hi0), q6_neg(lo1), q6_neg(hi1)}, c);
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = (int)((uint32_t)c[j] << 1);
#pragma unroll
            for (int i = 3; i >= 0; --i) {
                c = mfma_i8_16x16x64(A, u32x4{q6_piece(lo0, SL0, i), q6_piece(hi0, SH0, i), q6_piece(lo1, SL1, i), q6_piece(hi1, SH1, i)}, c);
                if (i > 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) c[j] = (int)((uint32_t)c[j] << 2);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][l] = fmaf(D[j], (float)c[j], acc[j][l]);
        }
    }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float p0 = acc[j][0] + pf_partner<NT>(acc[j][0]), p1 = acc[j][1] + pf_partner<NT>(acc[j][1]);
        const float p2 = acc[j][2] + pf_partner<NT>(acc[j][2]), p3 = acc[j][3] + pf_partner<NT>(acc[j][3]);
        res[j] = (p0 + p2) + (p1 + p3);
    }
}

template <int TYPE, int TOK>
DEV void pfm_item(const uint8_t* __restrict__ w0, int item, int n_tiles, int nb, const int* __restrict__ lds, int act_words,
                  int K, int lane, float (&res)[4]) {
    if constexpr (TYPE == GT_Q6_K) {
        if constexpr (TOK == 16) pfm_item_q6k<16>(w0, item, n_tiles, nb, lds, act_words, K, lane, res);
        else pfm_item_q6k_t8<TOK>(w0, item, n_tiles, nb, lds, act_words, K, lane, res);
    } else {
        if constexpr (TOK == 16) pfm_item_q45<TYPE, 16>(w0, item, n_tiles, nb, lds, act_words, K, lane, res);
        else pfm_item_q45_t8<TYPE, TOK>(w0, item, n_tiles, nb, lds, act_words, K, lane, res);
    }
}

// Launch over the jobs of a site that have weight type TYPE (one launch per type: a kernel holding all three forms needs
// more than 256 VGPRs, and uniform items balance better — a Q6_K item costs about twice a Q4_K item): items are 16-row
// pairs of tiles, job after job; 512 threads, wave w takes items w * gridDim.x + blockIdx.x + k * 8 * gridDim.x.
template <int TYPE, int TOK, bool GU>
__global__ void __launch_bounds__(512) matvec_pfm_kernel(const PfArgs a) {
    CT_DYN_SMEM(smem_raw);
    int* lds = reinterpret_cast<int*>(smem_raw);
    const MatvecArgs& m = a.m;
    const int tid = (int)threadIdx.x, lane = lane_id();
    const int wv = uniform_int(wave_id());
    const int t0 = (int)blockIdx.y * TOK;
    const int nt = a.n_tok - t0 < TOK ? a.n_tok - t0 : TOK;
    {
        // All TOK images of the group are copied (the scratch buffer always holds kPfChunk of them; images past n_tok are stale
        // and their results are dropped at the store): no predicates, eight loads in flight per thread, tail indices clamped.
        const u32x4* src = (const u32x4*)(a.acts + (size_t)t0 * a.act_words);
        const int n16 = TOK * (a.act_words >> 2);
        for (int i0 = 0; i0 < n16; i0 += 8 * 512) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * 512 + tid; v[u] = ld16(src + (i < n16 ? i : n16 - 1)); }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * 512 + tid; ((u32x4*)lds)[i < n16 ? i : n16 - 1] = v[u]; }
        }
    }
    __syncthreads();
    const int pos0 = (m.pos ? *m.pos : 0) + t0;
    const int GX = (int)gridDim.x, nb = m.K >> 8;
    const int r16 = lane & 15, q = lane >> 4;
    for (int item = wv * GX + (int)blockIdx.x; item < m.n_pairs; item += 8 * GX) {
        if constexpr (GU) {
            const int n_tiles = (m.job[0].w.M + 7) / 8;
            float gate[4], up[4];
            pfm_item<TYPE, TOK>(m.job[0].w.p[0], item, n_tiles, nb, lds, a.act_words, m.K, lane, gate);
            pfm_item<TYPE, TOK>(m.job[1].w.p[0], item, n_tiles, nb, lds, a.act_words, m.K, lane, up);
            const int row = item * 16 + r16;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = TOK == 4 ? (q == 0 ? j : TOK) : 4 * q + j;   // 4-token form: lanes 0..15 hold the results
                if (t < nt && row < m.job[0].w.M)
                    m.out[(size_t)(t0 + t) * a.ld_out + row] = f16_bits_to_f32(m.silu_tab[f32_to_f16_bits(gate[j])]) * up[j];
            }
        } else {
            int jb = 0;
            if (m.njobs > 1 && item >= m.job[1].pair0) jb = 1;
            if (m.njobs > 2 && item >= m.job[2].pair0) jb = 2;
            const int it = item - m.job[jb].pair0;
            float res[4];
            pfm_item<TYPE, TOK>(m.job[jb].w.p[0], it, (m.job[jb].w.M + 7) / 8, nb, lds, a.act_words, m.K, lane, res);
            const int row = it * 16 + r16;
            const bool row_ok = row < m.job[jb].w.M;
            const int epi = m.job[jb].epi;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = TOK == 4 ? (q == 0 ? j : TOK) : 4 * q + j;
                const bool own = row_ok && t < nt;
                const int tok = t0 + t, pos = pos0 + t;
                if (epi == EPI_ADD) {
                    if (own) m.out[(size_t)tok * a.ld_out + row] = res[j] + m.res[(size_t)tok * a.ld_res + row];
                } else if (epi == EPI_STORE) {
                    if (own) m.out[(size_t)tok * a.ld_out + row] = res[j];
                } else if (epi == EPI_GELU) {
                    if (own) m.out[(size_t)tok * a.ld_out + row] = f16_bits_to_f32(m.gelu_tab[f32_to_f16_bits(res[j])]);
                } else if (epi == EPI_ADD2) {
                    if (own) m.out[(size_t)tok * a.ld_out + row] = (res[j] + m.res[(size_t)tok * a.ld_res + row]) + m.res2[(size_t)tok * a.ld_res + row];
                } else if (epi == EPI_V) {
                    if (own) m.vcache[(size_t)row * m.v_stride + pos] = f32_to_f16_bits(res[j]);
                } else {   // RoPE, normal mode (ggml.c:12522-12539): rows 2i, 2i+1 are neighbouring lanes
                    const float other = lane_xor1(res[j]);
                    if (own) {
                        const int ip = (row % m.head_dim) >> 1;
                        const float cs = m.rope_cs[((size_t)pos * (m.head_dim >> 1) + ip) * 2 + 0];
                        const float sn = m.rope_cs[((size_t)pos * (m.head_dim >> 1) + ip) * 2 + 1];
                        const float o = (row & 1) ? fmaf(res[j], cs, other * sn) : fmaf(res[j], cs, -(other * sn));
                        if (epi == EPI_ROPE_Q) m.q_f16[(size_t)tok * a.ld_q + row] = f32_to_f16_bits(o);
                        else m.kcache[kcache_off(pos, row, m.head_dim, m.n_ctx)] = f32_to_f16_bits(o);
                    }
                }
            }
        }
    }
}
