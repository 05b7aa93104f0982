// K-quant building blocks shared by the decode mat-vec (kernels_v7.h), the Q8_0 / Q4_0 kernels (kernels_q32.h) and the prompt-chunk
// kernels: the per-lane register image of one weight block in the record field order (quant.h: a LAYOUT_R2C4 record has the
// field order of an 8-row tile record, slot p in place of row p), the integer work of one block -> the operands of the reference's
// f32 chain, and the LDS counters of the systolic kernels.  (The mat-vec generations 5 / 6 and the wide-K kernel these pieces
// came from were retired in round 2: generation 7 is faster on every shape, profiles/r02_generations_ab_wide_shapes.txt.)
#pragma once
#include "kernels_exact.h"

template <int TYPE> struct BlkImg;
template <> struct BlkImg<GT_Q4_K> { u32x4 hdr, qs; };
template <> struct BlkImg<GT_Q5_K> { u32x4 hdr, qs, qh; };
template <> struct BlkImg<GT_Q6_K> { u32x4 sc, ql, qh; uint32_t d; };

template <int TYPE> DEV BlkImg<TYPE> img_load(const uint8_t* rec, const LaneGeom& G);
template <> DEV BlkImg<GT_Q4_K> img_load<GT_Q4_K>(const uint8_t* rec, const LaneGeom& G) {
    BlkImg<GT_Q4_K> R;
    R.hdr = ld_stream16(rec + G.off_hdr);
    R.qs = ld_stream16(rec + 128 + G.off_qs);
    return R;
}
template <> DEV BlkImg<GT_Q5_K> img_load<GT_Q5_K>(const uint8_t* rec, const LaneGeom& G) {
    BlkImg<GT_Q5_K> R;
    R.hdr = ld_stream16(rec + G.off_hdr);
    R.qh = ld_stream16(rec + G.off_qh5);
    R.qs = ld_stream16(rec + 384 + G.off_qs);
    return R;
}
template <> DEV BlkImg<GT_Q6_K> img_load<GT_Q6_K>(const uint8_t* rec, const LaneGeom& G) {
    BlkImg<GT_Q6_K> R;
    R.d = *(const uint16_t*)(rec + G.off6_d);
    R.sc = ld_stream16(rec + G.off6_sc);
    R.qh = ld_stream16(rec + G.off6_qh);
    R.ql = ld_stream16(rec + G.off6_ql);
    return R;
}

template <int TYPE> DEV constexpr uint32_t rec_bytes() { return TYPE == GT_Q4_K ? 1152u : (TYPE == GT_Q5_K ? 1408u : 1680u); }

#ifdef CT_EMU
DEV void lds_signal(unsigned* ctr, int lane, unsigned inc) {
    if (lane == 0) *ctr += inc;
}
DEV void lds_wait_ge(const unsigned* ctr, unsigned target) {
    while (*(const volatile unsigned*)ctr < target) emu::spin_yield();
}
#else
DEV void lds_signal(unsigned* ctr, int lane, unsigned inc) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's LDS writes are visible before the count moves
    if (lane == 0) __hip_atomic_fetch_add(ctr, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
DEV void lds_wait_ge(const unsigned* ctr, unsigned target) {
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
#endif

// Integer work of one block -> this lane's chain operands (the reference's integer arithmetic per 256-block, citations in kernels_exact.h).
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
