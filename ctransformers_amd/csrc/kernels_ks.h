// K-quant mat-vec for wide rows (K > 12288: ffn_down of Llama-2-70B, K = 28672, and of Falcon-40B, K = 32768), systolic
// like kernels_q32.h: the blocks of every 8-row tile are split over the 16 waves, a wave turns its blocks into register
// operands of the reference's f32 chain — (d_b, (float)sumi_b[l]) and, for Q4_K/Q5_K, (-dmin_b, (float)prod_b) — with
// the integer arithmetic of img_to_chain (kernels_v5.h), and the two accumulators per lane travel from wave to wave
// through an LDS mailbox; the last wave finishes with the AVX reduction tree and the epilogue.  No chain storage, so K
// is bounded only by the activation vector in LDS (45 KB at K = 32768).  Replaces the wave-per-tile fallback
// (matvec_exact_kernel) that streamed these matrices at 1.7 TB/s.
#pragma once
#include "kernels_v6.h"

constexpr int kKsSlots = 8;
template <int MAXK> struct SmemKS {
    ActLdsX<MAXK> L;
    float mail[kKsSlots][2][64];
    unsigned ctr[kKsSlots];
};

// Integer work of one block -> this lane's chain operands (same arithmetic as img_to_chain<TYPE>).
// b: block of the activation image (may differ per lane), q8w: word offset of that block's 64 quant words in L.q8.
template <int TYPE, class ACT>
DEV void img_to_regs(const BlkImg<TYPE>& R, int b, int q8w, const ACT& L, const LaneGeom& G, float& sv, float& dv, float& mv,
                     float& pv) {
    const int c = G.c;
    const float yd = L.yd[b];
    if constexpr (TYPE == GT_Q4_K || TYPE == GT_Q5_K) {
        const int* alo = &L.q8[q8w + G.a45];
        const int* ahi = alo + 8;
        const uint32_t lo_w = c < 2 ? R.hdr[1] : (c == 2 ? R.hdr[2] : R.hdr[3]);
        const uint32_t hi_w = c < 2 ? R.hdr[2] : R.hdr[3];
        const uint32_t x = alignbit32(hi_w, lo_w, (uint32_t)((24 * c) & 31));
        const int sc_lo = (int)(x & 63u), sc_hi = (int)bfe32(x, 6, 6), m_lo = (int)bfe32(x, 12, 6), m_hi = (int)bfe32(x, 18, 6);
        int w8[8], a8[8], d8[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t lo = R.qs[k] & 0x0F0F0F0Fu;
            uint32_t hi = (R.qs[k] >> 4) & 0x0F0F0F0Fu;
            if constexpr (TYPE == GT_Q5_K) {
                lo |= ((R.qh[k] >> (2 * c)) & 0x01010101u) << 4;
                hi |= ((R.qh[k] >> (2 * c + 1)) & 0x01010101u) << 4;
            }
            w8[k] = (int)lo; w8[4 + k] = (int)hi;
            a8[k] = alo[k]; a8[4 + k] = ahi[k];
        }
        dot4x8(d8, w8, a8);
        int part[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) part[k] = mul24(sc_lo, d8[k]) + mul24(sc_hi, d8[4 + k]);
        sv = (float)quad_transpose_reduce_dpp(part[0], part[1], part[2], part[3], c);
        int prod = mul24(m_lo, L.sb[b * 8 + 2 * c]) + mul24(m_hi, L.sb[b * 8 + 2 * c + 1]);
        if constexpr (TYPE == GT_Q5_K) {
            if (G.h != 0) prod = 0;
            prod += lane_xor2(prod);
            prod += lane_xor4(prod);
        }
        pv = (float)prod;
        dv = yd * f16_bits_to_f32((uint16_t)(R.hdr[0] & 0xFFFF));
        mv = -yd * f16_bits_to_f32((uint16_t)(R.hdr[0] >> 16));
    } else {
        const int n = G.g >> 2;
        const int* alo = &L.q8[q8w + G.a6];
        const int* ahi = alo + 16;
        const uint32_t w_lo = n ? R.sc[2] : R.sc[0];
        const uint32_t w_hi = n ? R.sc[3] : R.sc[1];
        const int sc_lo = (int)(int8_t)((w_lo >> G.sc_sh6) & 0xFF);
        const int sc_hi = (int)(int8_t)((w_hi >> G.sc_sh6) & 0xFF);
        int w8[8], a8[8], d8[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            w8[k] = (int)((R.ql[k] & 0x0F0F0F0Fu) | (((R.qh[k] >> G.s_lo6) & 0x03030303u) << 4));
            w8[4 + k] = (int)(((R.ql[k] >> 4) & 0x0F0F0F0Fu) | (((R.qh[k] >> G.s_hi6) & 0x03030303u) << 4));
            a8[k] = alo[k]; a8[4 + k] = ahi[k];
        }
        dot4x8_bias(d8, w8, a8, (int)0xE0E0E0E0u);   // (q6 - 32) . a = q6 . a + (-32,-32,-32,-32) . a
        int part[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) part[k] = mul24(sc_lo, d8[k]) + mul24(sc_hi, d8[4 + k]);
        sv = (float)quad_transpose_reduce_dpp(part[0], part[1], part[2], part[3], c);
        dv = yd * f16_bits_to_f32((uint16_t)(R.d & 0xFFFF));
        mv = 0.0f;
        pv = 0.0f;
    }
}

template <int TYPE, int MAXK, int MAXB>   // the wide-K launches (ffn_down) never carry a norm: PLAIN prologue only
__global__ void __launch_bounds__(1024) matvec_ks_kernel(const MatvecArgs a) {
    static_assert(MAXB % 4 == 0, "blocks are processed four images at a time");
    __shared__ SmemKS<MAXK> SM;
    constexpr bool mins = TYPE != GT_Q6_K;
    const int lane = lane_id();
    const int wv = uniform_int(wave_id());
    const LaneGeom G = lane_geom(lane);
    const int nb = a.K >> 8;
    const int NA = nb < 16 ? nb : 16;
    if (threadIdx.x < kKsSlots) SM.ctr[threadIdx.x] = 0u;
    prologue_q8k_exact16<1024, MAXK>(SM.L, a.x, a.norm_w, a.K, a.pro, a.eps);
    if (wv >= NA) return;
    const int base = nb / NA, rem = nb % NA;
    const int bcnt = base + (wv < rem ? 1 : 0);
    const int bbeg = wv * base + (wv < rem ? wv : rem);
    const int stride = (int)gridDim.x, first = (int)blockIdx.x;
    const int n_seq = first < a.n_pairs ? (a.n_pairs - first + stride - 1) / stride : 0;
    auto tile_of = [&](int seq, int& j, int& tile) __attribute__((always_inline)) {
        const int it = first + seq * stride;
        j = 0;
        if (a.njobs > 1 && it >= a.job[1].pair0) j = 1;
        if (a.njobs > 2 && it >= a.job[2].pair0) j = 2;
        tile = it - a.job[j].pair0;
    };
    BlkImg<TYPE> R[4];
    auto load_chunk = [&](int seq, int ch) __attribute__((always_inline)) {   // images of blocks bbeg + 4*ch .. +3 of tile `seq`
        int j, tile;
        tile_of(seq, j, tile);
        const uint8_t* tp = a.job[j].w.p[0] + ((size_t)tile * nb + bbeg) * rec_bytes<TYPE>();
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (4 * ch + u < bcnt) R[u] = img_load<TYPE>(tp + (size_t)(4 * ch + u) * rec_bytes<TYPE>(), G);
    };
    if (n_seq > 0) load_chunk(0, 0);
    for (int seq = 0; seq < n_seq; ++seq) {
        float sv[MAXB], dv[MAXB], mv[MAXB], pv[MAXB];
#pragma unroll
        for (int ch = 0; ch < MAXB / 4; ++ch) {
            BlkImg<TYPE> Q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) Q[u] = R[u];
            // request what comes next before the arithmetic on what is here
            if (4 * (ch + 1) < bcnt) load_chunk(seq, ch + 1);
            else if (ch == (bcnt - 1) / 4 && seq + 1 < n_seq) load_chunk(seq + 1, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 4 * ch + u;
                if (k < bcnt) img_to_regs<TYPE>(Q[u], bbeg + k, (bbeg + k) * 64, SM.L, G, sv[k], dv[k], mv[k], pv[k]);
            }
        }
        const int slot = seq % kKsSlots;
        lds_wait_ge(&SM.ctr[slot], (unsigned)(seq / kKsSlots) * (unsigned)NA + (unsigned)wv);
        float acc = 0.0f, accm = 0.0f;
        if (wv > 0) { acc = SM.mail[slot][0][lane]; if (mins) accm = SM.mail[slot][1][lane]; }
#pragma unroll
        for (int k = 0; k < MAXB; ++k) {
            if (k < bcnt) {
                acc = fmaf(dv[k], sv[k], acc);
                if (mins) accm = fmaf(mv[k], pv[k], accm);
            }
        }
        if (wv < NA - 1) {
            SM.mail[slot][0][lane] = acc;
            if (mins) SM.mail[slot][1][lane] = accm;
            lds_signal(&SM.ctr[slot], lane, 1u);
            continue;
        }
        lds_signal(&SM.ctr[slot], lane, 1u);
        float res = hsum8_exact_dpp(acc);
        if constexpr (mins) {
            if constexpr (TYPE == GT_Q4_K) {
                const float wsum = accm + lane_xor4(accm);
                accm = wsum + lane_xor2(wsum);
            }
            accm = __shfl(accm, lane & ~7);
            res = res + accm;
        }
        int j, tile;
        tile_of(seq, j, tile);
        const int row = tile * 8 + G.r;
        const bool own = G.g == 0 && row < a.job[j].w.M;
        const int epi = a.job[j].epi;
        if (epi == EPI_ADD) {
            if (own) a.out[row] = res + a.res[row];
        } else if (epi == EPI_ADD2) {
            if (own) a.out[row] = (res + a.res[row]) + a.res2[row];
        } else if (epi == EPI_GELU) {
            if (own) a.out[row] = f16_bits_to_f32(a.gelu_tab[f32_to_f16_bits(res)]);
        } else {   // EPI_STORE (the wide-K launches are ffn_down / lm_head style: no RoPE / cache epilogues)
            if (own) a.out[row] = res;
        }
    }
}
