// Prompt chunks, K-quant weights on the f16 matrix cores — bit-identical to the reference CPU build.
//
// What the reference computes per (row, token) and 256-block (k_quants.c:2651-2720 Q4_K, :3183-3262 Q5_K, :3800-3872 Q6_K) is
// eight int32 lane sums
//     sumi[l] = sum over the block's eight 32-element vectors v of  scale_v(l) * sum_{e<4} w_v[4l+e] * a_v[4l+e]
// and then ONE f32 fma per lane and block, acc[l] = fma(y.d * fp16(x.d), (float)sumi[l], acc[l]) (plus the min term).  Each sumi[l] is
// a K = 32 contraction (8 vectors x 4 elements) of integers: |scale * w| <= 63 * 31 (Q4_K / Q5_K), <= 128 * 32 (Q6_K), |a| <= 128.
// fp16 holds every integer up to 2048 and every even integer up to 4096, and the matrix core accumulates in f32, so with
//     B[k][row] = (half)(scale * w)          (Q6_K: two operands, scale = (scale & ~1) + (scale & 1), chained through C)
//     A[token][k] = (half)a
// ONE v_mfma_f32_16x16x32_f16 per AVX lane l returns (float)sumi[l] for 16 rows x 16 tokens EXACTLY (all partial sums are
// integers below 2^24: tools/experiments/mfma_f16_exact.cpp) — no scale splitting, no integer->float conversion, no shifts; the
// f32 chain, the min-term accumulators and hsum_float_8 stay the decode kernels', in-lane.  The min term is a matrix product as
// well: prod[t] = sum_{i<4} m[2t + i/2] * bsum16[4t + i] on v_mfma_f32_16x16x16_f16 (sums of 16 quants are <= 2048).
// (Round 1 did this on the int8 cores with the scale split into 2-5 digit products per lane: 9.5k prompt tok/s against 16.5k here.)
//
// Shapes.  A wave owns 16 weight rows (8 row pairs of the LAYOUT_R2C4 arena the decode kernels read: one copy of the weights
// serves both) x TG tokens (TG / 16 matrix products per unpacked weight operand: the nibble -> fp16 unpack is amortised over
// them).  The MFMA is issued transposed, D[token][row]: a lane's four results are four tokens of ITS row (lane & 15), so d, dmin,
// scales and mins are already in the lane that loaded the row's header.  Activations are streamed: the quantize kernel writes
// per (token group, block) one STAGE image (fp16 quants in operand order, fp16 sums of 16, f32 y.d); the NW waves of a workgroup
// copy stage b + 1 into one half of LDS while computing on stage b in the other (one barrier per block).
//   weights   lane (row r16 = lane & 15, p = lane >> 4): header (16 B) + the 32 bytes that hold vectors va(p), vb(p) of the block
//   tokens    lane (token n = lane & 15, p): 16 B = halves [va: e0 e2 e1 e3 | vb: e0 e2 e1 e3] of elements 4l + e — the order
//             the packed nibble extraction yields (w & 0x000F000F pairs bytes 0 and 2)
//   results   lane (r16, p), register j: token 4p + j
#pragma once
#include "kernels_pf.h"

// Byte offsets inside one (token group, block) stage image.  Values [l][p][tok] x 16 B: the 16 lanes of a ds_read_b128 phase
// hold 16 different tokens -> conflict-free.  Sums [kg][tok] x 8 B with the kg stride = 128 mod 256 (kg 0 / 1 on different banks).
// The kPgWaves waves of a workgroup copy a stage by LDS-DMA without a branch: the values as whole 1 KB pieces (16 B per lane),
// TG / 16 per wave; the tail (sums, y.d; padded) as 256 B pieces (4 B per lane), TAILP per wave.
#ifndef PG_WAVES
#define PG_WAVES 8   // waves (= 16-row items) per workgroup; 4: two workgroups per CU, their per-step barriers independent (A/B switch)
#endif
constexpr int kPgWaves = PG_WAVES;
template <int TG> struct PgStage {
    static constexpr int VALS = 0;
    static constexpr int SUMS = 512 * TG;
    static constexpr int SUM_STRIDE = 8 * TG + ((8 * TG) % 256 == 128 ? 0 : 128);
    static constexpr int YD = SUMS + 4 * SUM_STRIDE;
    static constexpr int TAILP = (YD + 4 * TG - SUMS + kPgWaves * 256 - 1) / (kPgWaves * 256);
    static constexpr int BYTES = SUMS + TAILP * kPgWaves * 256;
};
constexpr int pg_stage_bytes(int tg) { return tg == 16 ? PgStage<16>::BYTES : PgStage<32>::BYTES; }

struct PgArgs {
    MatvecArgs m;          // jobs (w.r2 = LAYOUT_R2C4 records) and epilogue operands
    const uint8_t* acts;   // stage images [token group][block] of the layout this weight type reads
    int n_tok;             // tokens in this chunk
    int n_items;           // 16-row items over all jobs of the launch
    int ld_out, ld_res, ld_q;
};

// One workgroup per token: (RMSNorm / LayerNorm ->) Q8_K exactly as the decode prologue does it, written as stage images.
// img45: pieces (l, p) hold vectors 2p, 2p+1 (Q4_K / Q5_K: low / high nibbles of qs bytes 32p..32p+31); img6: vectors
// 4(p >> 1) + (p & 1) and + 2 (Q6_K: low / high nibbles of ql bytes 32p..32p+31).  Either may be null.
// NT threads.  (Round 5: workgroups of only as many waves as own blocks — 256 threads for K <= 4096, 768 for K <= 12288 — measured the same
// prompt rate as 1024: 17 130-17 340 against 17 180-17 300 tok/s, alternating on one box; the launch is not bound by its wave launches.)
template <int MAXK, bool LN, int NT = 1024>
__global__ void __launch_bounds__(NT) pg_quantize_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ nw, int K,
                                                           int pro, float eps, uint8_t* __restrict__ img45, uint8_t* __restrict__ img6,
                                                           int tg45, int tg6, const float* __restrict__ nb_, int n_tok) {
    __shared__ ActLdsX<MAXK> L;
    // Workgroup -> token: the eight tokens whose 16-byte pieces fill one 128-byte line of a stage image go to workgroups on ONE XCD
    // (linear ids equal mod 8 when the chunk has 16 lines of tokens), so that its L2 writes the line back whole instead of eight
    // XCDs writing 16 bytes each.  grid = 8 * ceil(n_tok / 8); n_tok < 0: the identity (A/B switch CT_AMD_PGQ_REMAP=0).
    // tg45 / tg6: tokens per group of the two image layouts (the launches of the two weight-type families choose their own: engine.cc).
    int t = (int)blockIdx.x;
    if (n_tok >= 0) {
        const int G = (int)gridDim.x >> 3;
        t = (t % G) * 8 + t / G;
        if (t >= n_tok) return;
    }
    const int tid = (int)threadIdx.x;
    if constexpr (LN) prologue_q8k_exact16_ln<NT, MAXK>(L, x + (size_t)t * ldx, nw, K, pro, eps, nb_);
    else prologue_q8k_exact16<NT, MAXK>(L, x + (size_t)t * ldx, nw, K, pro, eps);
    const int nb = K >> 8;
#pragma unroll
    for (int lay = 0; lay < 2; ++lay) {
        uint8_t* img = lay ? img6 : img45;
        if (!img) continue;
        const int tg = lay ? tg6 : tg45;
        const int g = t / tg, tt = t - g * tg;
        const int sums = 512 * tg, sum_stride = 8 * tg + ((8 * tg) % 256 == 128 ? 0 : 128), yd_off = sums + 4 * sum_stride;
        const size_t sbytes = (size_t)pg_stage_bytes(tg);
        for (int i = tid; i < nb * 32; i += NT) {
            const int b = i >> 5, l = (i >> 2) & 7, p = i & 3;
            const int va = lay ? 4 * (p >> 1) + (p & 1) : 2 * p, vb = lay ? va + 2 : va + 1;
            const uint32_t wa = (uint32_t)L.q8[b * 64 + va * 8 + l], wb = (uint32_t)L.q8[b * 64 + vb * 8 + l];
            uint32_t o[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t w = h ? wb : wa;
                const uint32_t e0 = f32_to_f16_bits((float)(int8_t)(w & 0xFF)), e1 = f32_to_f16_bits((float)(int8_t)((w >> 8) & 0xFF));
                const uint32_t e2 = f32_to_f16_bits((float)(int8_t)((w >> 16) & 0xFF)), e3 = f32_to_f16_bits((float)(int8_t)(w >> 24));
                o[2 * h] = e0 | (e2 << 16);
                o[2 * h + 1] = e1 | (e3 << 16);
            }
            uint32_t* dst = (uint32_t*)(img + ((size_t)g * nb + b) * sbytes + (size_t)((l * 4 + p) * tg + tt) * 16);
            dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];
        }
        if (!lay) {   // the sums of 16 feed the min term: Q4_K / Q5_K images only
            for (int i = tid; i < nb * 4; i += NT) {
                const int b = i >> 2, kg = i & 3;
                const uint32_t s0 = f32_to_f16_bits((float)L.bsums[b * 16 + 4 * kg]), s1 = f32_to_f16_bits((float)L.bsums[b * 16 + 4 * kg + 1]);
                const uint32_t s2 = f32_to_f16_bits((float)L.bsums[b * 16 + 4 * kg + 2]), s3 = f32_to_f16_bits((float)L.bsums[b * 16 + 4 * kg + 3]);
                uint32_t* dst = (uint32_t*)(img + ((size_t)g * nb + b) * sbytes + sums + kg * sum_stride + tt * 8);
                dst[0] = s0 | (s1 << 16); dst[1] = s2 | (s3 << 16);
            }
        }
        for (int b = tid; b < nb; b += NT) *(float*)(img + ((size_t)g * nb + b) * sbytes + yd_off + tt * 4) = L.yd[b];
    }
}

// (w & 0x000F000F) | 0x64006400 is one v_and_or_b32 only with the second constant in a VECTOR register (a gfx9 VOP3 instruction
// takes one scalar-or-literal operand); per-lane masks that pick the header words of scale group p.
struct PgConst { uint32_t one, one6, sel01, sel2, sel3, gsh; };
DEV PgConst pg_consts(int p) {
    PgConst C;
    C.one = vgpr_const(0x64006400u);
    C.one6 = vgpr_const(0x58005800u);
    C.sel01 = p < 2 ? 0xFFFFFFFFu : 0u; C.sel2 = p == 2 ? 0xFFFFFFFFu : 0u; C.sel3 = p == 3 ? 0xFFFFFFFFu : 0u;   // lane masks: header word of scale group p
    C.gsh = (uint32_t)((24 * p) & 31);
    return C;
}

// ---- per-lane weight registers of one block ---------------------------------------------------------------------------------
template <int TYPE> struct PgRec;
template <> struct PgRec<GT_Q4_K> { u32x4 h, qa, qb; };
template <> struct PgRec<GT_Q5_K> { u32x4 h, qa, qb, ha, hb; };
template <> struct PgRec<GT_Q6_K> { u32x4 sc, q[4]; uint32_t d; };   // q[l >> 1] = {A(l), B(l), A(l + 1), B(l + 1)} (quant.h:r2c4_record_bytes)

// Block 4 * RD + CB of the lane's row, counted from the record `rec0` points at: record +RD, slot 4 * (row & 1) + CB (quant.h
// LAYOUT_R2C4).  RD and CB are compile-time, the lane-dependent parts sit in three 32-bit offsets: every load is (advancing
// pointer + lane offset) with an immediate — nothing for the loop optimiser to turn into a dozen 64-bit induction pointers.
struct PgLane { uint32_t h, x, q; };
template <int TYPE> DEV PgLane pg_lane(int rr, int p) {
    PgLane L;
    if constexpr (TYPE == GT_Q4_K) { L.h = (uint32_t)rr * 64u; L.x = 0u; L.q = (uint32_t)rr * 512u + (uint32_t)p * 32u; }
    else if constexpr (TYPE == GT_Q5_K) { L.h = (uint32_t)rr * 64u; L.x = (uint32_t)rr * 128u; L.q = (uint32_t)rr * 512u + (uint32_t)p * 32u; }
    else { L.h = (uint32_t)rr * 64u; L.x = 0u; L.q = (uint32_t)rr * 1024u + (uint32_t)p * 64u; }
    return L;
}
template <int TYPE, int RD, int CB>
DEV PgRec<TYPE> pg_load(const uint8_t* __restrict__ rec0, const PgLane& L) {
    constexpr uint32_t REC = TYPE == GT_Q4_K ? 1152u : (TYPE == GT_Q5_K ? 1408u : 2192u);   // r2c4_record_bytes(TYPE)
    const uint8_t* rec = rec0 + RD * REC;
    PgRec<TYPE> R;
    if constexpr (TYPE == GT_Q4_K) {
        R.h = ld16(rec + L.h + CB * 16);
        R.qa = ld16(rec + L.q + 128 + CB * 128);
        R.qb = ld16(rec + L.q + 128 + CB * 128 + 16);
    } else if constexpr (TYPE == GT_Q5_K) {
        R.h = ld16(rec + L.h + CB * 16);
        R.ha = ld16(rec + L.x + 128 + CB * 32);
        R.hb = ld16(rec + L.x + 128 + CB * 32 + 16);
        R.qa = ld16(rec + L.q + 384 + CB * 128);
        R.qb = ld16(rec + L.q + 384 + CB * 128 + 16);
    } else {
        R.d = *(const uint16_t*)(rec + (L.h >> 3) + CB * 2);
        R.sc = ld16(rec + L.h + 16 + CB * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) R.q[k] = ld16(rec + L.q + 144 + CB * 256 + k * 16);
    }
    return R;
}

template <int TYPE> struct PgAcc {   // accumulators of one 16-token group: [AVX lane][token j], min term [t][token j]
    static constexpr int NM = TYPE == GT_Q4_K ? 4 : (TYPE == GT_Q5_K ? 1 : 0);
    float a[8][4];
    float m[NM ? NM : 1][4];
};

// One block of the lane's row against the TG tokens of the stage in `buf`.  Software-pipelined by hand: the matrix products of
// AVX lane l are issued, then the f32 chain steps of lane l - 1 consume the previous results, with a scheduling barrier per
// lane and a register fence on the accumulators just stepped — left to itself hipcc places every chain step of the (branch-free)
// loop body at its very end and spills the matrix products of four blocks on the way there; and a reload from scratch is an
// `s_waitcnt vmcnt(0)` on the prefetched weight loads.
// What a step requests besides computing: the copy of stage b + 2 (DMA pieces of this wave) and the refill of the weight ring slot.
// Issued from inside the block's lane loop, the refill at AVX lane 0 and two DMA pieces per lane from lane 4: a burst of six memory
// instructions at the top of the step costs the wave ~400 cycles of issue stalls (in-kernel stamps), spread out they hide behind the
// unpack and the matrix products.
struct PgFeed { const uint8_t* src; uint8_t* dst; const uint8_t* rec0; int wv, lane16, lane4; };
template <int TYPE, int TG, int RD, int CB>
DEV void pg_feed(int l, const PgFeed& F, const PgLane& LN, PgRec<TYPE>& ring) {
    using ST = PgStage<TG>;
    constexpr int NW = kPgWaves, G = TG / 2 / NW;   // 1 KB value pieces per wave (TG x 512 bytes over NW waves)
    static_assert(G + ST::TAILP <= 8, "the step issues at most eight DMA pieces per wave");
    // The refill goes first: hipcc does not see the DMA and counts its own loads only, so its `s_waitcnt vmcnt(N)` for an older ring
    // slot lets exactly its N youngest loads stay in flight — with the DMA pieces younger than the refill they stay in flight too,
    // issued before it they would have to land first.
    if (l == 0) ring = pg_load<TYPE, RD, CB>(F.rec0, LN);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = 2 * (l - 4) + k;   // after the step's barrier (behind lane 3): every wave has left the previous step, whose stage
        if (i < 0) continue;              // buffer this copy overwrites
        if (i < G) glds16(F.src + (size_t)(i * NW + F.wv) * 1024 + F.lane16, F.dst + (size_t)(i * NW + F.wv) * 1024);
        else if (i < G + ST::TAILP) glds4(F.src + ST::SUMS + (size_t)((i - G) * NW + F.wv) * 256 + F.lane4, F.dst + ST::SUMS + (size_t)((i - G) * NW + F.wv) * 256);
    }
}

// measurement only (CT_AMD_PG_TRACE): s_memtime stamps of wave 0 of workgroup (0, 0), kept in LDS and written out at the end
template <bool TRACE> DEV void pg_stamp(unsigned long long* tr, int idx) {
    if constexpr (TRACE) { if (tr && idx < 120) tr[idx] = clock64_dev(); }
}

template <int TYPE, int TG, int MIDWAIT, bool TRACE, int RD, int CB>
DEV void pg_block(const PgRec<TYPE>& R, PgRec<TYPE>& ring, const PgFeed& F, const PgLane& LN, const uint8_t* __restrict__ buf, int r16, int p,
                  PgAcc<TYPE> (&acc)[TG / 16], const PgConst& C, bool live, unsigned long long* tr, int ti) {
    using ST = PgStage<TG>;
    constexpr int G = TG / 16;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (TYPE == GT_Q4_K || TYPE == GT_Q5_K) {
        constexpr bool Q5 = TYPE == GT_Q5_K;
        constexpr int NMT = Q5 ? 1 : 4;   // min-term products per block
        const u32x4 H = R.h;
        // the four 24-bit scale groups {sc[2c], sc[2c+1], m[2c], m[2c+1]} x 6 bit of this row (engine.cc:place_kblock)
        // group p = bits [24p, 24p + 24) of words 1..3: funnel shift over the two words that hold it, picked by per-lane masks
        // (written as a ?: chain hipcc turns the selection into divergent branches with the waits on the header load inside)
        const uint32_t lo = (H[1] & C.sel01) | (H[2] & C.sel2) | (H[3] & C.sel3), hi = (H[2] & C.sel01) | (H[3] & C.sel2);
        const uint32_t xp = alignbit32(hi, lo, C.gsh);
        const uint32_t s_lo = h2_from_int((int)(xp & 63u)), s_hi = h2_from_int((int)bfe32(xp, 6, 6));
        const uint32_t c_lo = pk_mul_f16(s_lo, 0xE400E400u), c_hi = pk_mul_f16(s_hi, 0xE400E400u);   // -1024 * scale
        const uint32_t s_hi16 = pk_mul_f16(s_hi, 0x2C002C00u), c_hi16 = pk_mul_f16(s_hi, 0xD400D400u);   // Q4_K high nibbles in place: scale / 16, -64 * scale
        const uint32_t m_lo = h2_from_int((int)bfe32(xp, 12, 6)), m_hi = h2_from_int((int)bfe32(xp, 18, 6));
        const float dw = live ? f16_bits_to_f32((uint16_t)(H[0] & 0xFFFF)) : 0.0f, dmw = live ? f16_bits_to_f32((uint16_t)(H[0] >> 16)) : 0.0f;
        float D[G][4], DM[G][4];
        uint64_t AS[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const f32x4 yd = *(const f32x4*)(buf + ST::YD + (g * 16 + 4 * p) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { D[g][j] = yd[j] * dw; DM[g][j] = -yd[j] * dmw; }
            AS[g] = *(const uint64_t*)(buf + ST::SUMS + p * ST::SUM_STRIDE + (g * 16 + r16) * 8);
        }
        // min term: prod[t] = m[2t] * (bsum16[4t] + bsum16[4t+1]) + m[2t+1] * (bsum16[4t+2] + bsum16[4t+3]); k-group p of the product
        // carries exactly the lane's own pair of mins.  Q4_K: acc_m[t] = fma(-y.d * dmin, (float)prod[t], acc_m[t]) (one product per
        // t: the other k-groups zeroed); Q5_K: summs = fma(-y.d * dmin, (float)(prod[0] + .. + prod[3]), summs).
        const uint64_t MB = (uint64_t)m_lo | ((uint64_t)m_hi << 32);
        f32x4 pend[G], pendm[G];
        u32x4 An[G];   // token operands are fetched one AVX lane ahead: the LDS latency hides behind the previous lane's unpack and products
#pragma unroll
        for (int g = 0; g < G; ++g) An[g] = *(const u32x4*)(buf + ST::VALS + (p * TG + g * 16 + r16) * 16);
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            u32x4 Ac[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                Ac[g] = An[g];
                if (l < 7) An[g] = *(const u32x4*)(buf + ST::VALS + (((l + 1) * 4 + p) * TG + g * 16 + r16) * 16);
            }
            const uint32_t w = l < 4 ? R.qa[l & 3] : R.qb[l & 3];
            constexpr uint32_t NIB = 0x000F000Fu;   // C.one = 0x64006400: 0x6400 | n = 1024 + n
            u32x4 W;
            if constexpr (Q5) {
                uint32_t p0 = (w & NIB) | C.one, p1 = ((w >> 8) & NIB) | C.one, p2 = ((w >> 4) & NIB) | C.one, p3 = ((w >> 12) & NIB) | C.one;
                // fifth bit: qh byte e, bit 2p / 2p+1 = element e of vector 2p / 2p+1
                const uint32_t hq = (l < 4 ? R.ha[l & 3] : R.hb[l & 3]) >> (2 * p);
                p0 |= (hq << 4) & 0x00100010u; p1 |= (hq >> 4) & 0x00100010u;
                p2 |= (hq << 3) & 0x00100010u; p3 |= (hq >> 5) & 0x00100010u;
                // (1024 + n) * sc - 1024 * sc = n * sc: one rounding of an exactly representable integer
                W = u32x4{pk_fma_f16(p0, s_lo, c_lo), pk_fma_f16(p1, s_lo, c_lo), pk_fma_f16(p2, s_hi, c_hi), pk_fma_f16(p3, s_hi, c_hi)};
            } else {
                // the high nibbles stay where they are: 0x6400 | (n << 4) = 1024 + 16 n, and (1024 + 16 n) * (sc / 16) - 64 sc = n * sc
                // (sc / 16 and 64 sc are exact in fp16) — one shift per dword instead of three
                constexpr uint32_t NIBH = 0x00F000F0u;
                const uint32_t w8 = w >> 8;
                const uint32_t p0 = (w & NIB) | C.one, p1 = (w8 & NIB) | C.one, p2 = (w & NIBH) | C.one, p3 = (w8 & NIBH) | C.one;
                W = u32x4{pk_fma_f16(p0, s_lo, c_lo), pk_fma_f16(p1, s_lo, c_lo), pk_fma_f16(p2, s_hi16, c_hi16), pk_fma_f16(p3, s_hi16, c_hi16)};
            }
            f32x4 cur[G], curm[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                cur[g] = mfma_f16_16x16x32(Ac[g], W, zero);
                if (l < NMT) curm[g] = mfma_f16_16x16x16(AS[g], (Q5 || p == l) ? MB : (uint64_t)0, zero);
                else curm[g] = zero;
            }
            pg_feed<TYPE, TG, RD, CB>(l, F, LN, ring);
            if (l > 0) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    chain4(acc[g].a[l - 1], D[g], pend[g]);   // acc[l-1] = fma(y.d * d, (float)sumi[l-1], acc[l-1]) for four tokens, HERE
                    if (l - 1 < NMT) chain4(acc[g].m[l - 1], DM[g], pendm[g]);
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) { pend[g] = cur[g]; pendm[g] = curm[g]; }
            sched_fence();
            if (l == 3) { pg_stamp<TRACE>(tr, ti + 2); vm_wait<MIDWAIT>(); __syncthreads(); pg_stamp<TRACE>(tr, ti + 3); sched_fence(); }   // the copy issued ONE step ago has landed: see the kernel
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            chain4(acc[g].a[7], D[g], pend[g]);
        }
    } else {
        // Q6_K: lane p = (n = p >> 1, kq = p & 1) holds vectors 4n + kq (low nibbles, qh bits 2kq..) and 4n + 2 + kq (high nibbles,
        // qh bits 4 + 2kq..); scale of vector v for lanes l < 4 / >= 4: scales[2v], scales[2v + 1].
        const int n = p >> 1, kq = p & 1;
        const float dw = live ? f16_bits_to_f32((uint16_t)(R.d & 0xFFFF)) : 0.0f;
        float D[G][4];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const f32x4 yd = *(const f32x4*)(buf + ST::YD + (g * 16 + 4 * p) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) D[g][j] = yd[j] * dw;
        }
        const uint32_t w_a = n ? R.sc[2] : R.sc[0], w_b = n ? R.sc[3] : R.sc[1];   // scales 8n .. 8n+3 / 8n+4 .. 8n+7
        f32x4 pend[G];
        u32x4 An[G];   // token operands one AVX lane ahead (see the Q4_K branch)
#pragma unroll
        for (int g = 0; g < G; ++g) An[g] = *(const u32x4*)(buf + ST::VALS + (p * TG + g * 16 + r16) * 16);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int sa = (int)(int8_t)((w_a >> (8 * (2 * kq + h))) & 0xFFu), sb = (int)(int8_t)((w_b >> (8 * (2 * kq + h))) & 0xFFu);
            // scale = even part + low bit: (q6 - 32) * even is an even integer <= 4096, (q6 - 32) * bit is tiny: both exact in fp16
            const uint32_t ea = h2_from_int(sa & ~1), eb = h2_from_int(sb & ~1), ba = h2_from_int(sa & 1), bb = h2_from_int(sb & 1);
            const uint32_t cea = pk_mul_f16(ea, 0xD900D900u), ceb = pk_mul_f16(eb, 0xD900D900u);   // -160 * scale part
            const uint32_t cba = pk_mul_f16(ba, 0xD900D900u), cbb = pk_mul_f16(bb, 0xD900D900u);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int l = 4 * h + k;
                u32x4 Ac[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    Ac[g] = An[g];
                    if (l < 7) An[g] = *(const u32x4*)(buf + ST::VALS + (((l + 1) * 4 + p) * TG + g * 16 + r16) * 16);
                }
                const uint32_t A = R.q[l >> 1][2 * (l & 1)], B = R.q[l >> 1][2 * (l & 1) + 1];   // the record keeps the 6-bit values operand-ready
                constexpr uint32_t Q6M = 0x01F801F8u;   // C.one6 = 0x58005800: 0x5800 | (q6 << 3) = 128 + q6
                const uint32_t v0 = (A & Q6M) | C.one6, v1 = ((A >> 6) & Q6M) | C.one6;   // vector va: elements (0, 2), (1, 3)
                const uint32_t v2 = (B & Q6M) | C.one6, v3 = ((B >> 6) & Q6M) | C.one6;   // vector vb
                // (128 + q6) * s - 160 * s = (q6 - 32) * s
                const u32x4 WE = {pk_fma_f16(v0, ea, cea), pk_fma_f16(v1, ea, cea), pk_fma_f16(v2, eb, ceb), pk_fma_f16(v3, eb, ceb)};
                const u32x4 WB = {pk_fma_f16(v0, ba, cba), pk_fma_f16(v1, ba, cba), pk_fma_f16(v2, bb, cbb), pk_fma_f16(v3, bb, cbb)};
                f32x4 cur[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    cur[g] = mfma_f16_16x16x32(Ac[g], WE, mfma_f16_16x16x32(Ac[g], WB, zero));
                }
                pg_feed<TYPE, TG, RD, CB>(l, F, LN, ring);
                if (l > 0) {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        chain4(acc[g].a[l - 1], D[g], pend[g]);
                    }
                }
#pragma unroll
                for (int g = 0; g < G; ++g) pend[g] = cur[g];
                sched_fence();
                if (l == 3) { pg_stamp<TRACE>(tr, ti + 2); vm_wait<MIDWAIT>(); __syncthreads(); pg_stamp<TRACE>(tr, ti + 3); sched_fence(); }
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            chain4(acc[g].a[7], D[g], pend[g]);
        }
    }
}

// Launch over the jobs of a site that have weight type TYPE.  grid (ceil(n_items / NW) rounded up to a multiple of 8, token
// groups): the workgroups of one row range differ by a multiple of 8 in their linear id, i.e. run on the same XCD, whose L2
// then serves the re-reads of the range's weights by the other token groups.
template <int TYPE, int TG, int NW, bool GU, bool TRACE = false>
__global__ void __launch_bounds__(NW * 64, 2) matmul_pg_kernel(const uint8_t* acts0, int K0, int n_items0, const PgArgs a) {
    // (acts0 / K0 / n_items0 repeat a.acts, a.m.K, a.n_items as leading scalars: preloaded into SGPRs at wave launch — kernels_v9.h:matvec_v9_kernel —
    // so the first stage copies wait for no kernel-argument fetch)
#ifndef CT_EMU
    __builtin_assume(a.acts == acts0); __builtin_assume(a.m.K == K0); __builtin_assume(a.n_items == n_items0);
#else
    (void)acts0; (void)K0; (void)n_items0;
#endif
    kernarg_touch<16 + sizeof(PgArgs)>();   // (gpu.h) one round trip for every argument line
    CT_DYN_SMEM(smem);
    using ST = PgStage<TG>;
    constexpr int G = TG / 16, SB = ST::BYTES;
    constexpr uint32_t REC = TYPE == GT_Q4_K ? 1152u : (TYPE == GT_Q5_K ? 1408u : 2192u);   // r2c4_record_bytes(TYPE)
    const MatvecArgs& m = a.m;
    // (row block, token group) = (blockIdx.x, blockIdx.y).  Measured in round 3: a remapping that makes the token groups of one row
    // block neighbours in dispatch order on one XCD (so that they run together and share the XCD's L2) changes nothing (15 377 vs
    // 15 345 tok/s): the re-reads of a matrix by the later token groups are served by the 256 MB infinity cache, the step is bound by
    // instruction issue (in-kernel stamps: 2 x (181 VALU x 4 + 24 MFMA x 16) cycles per pair of waves and step).
    const int bx = (int)blockIdx.x, grp = (int)blockIdx.y;
    if (bx * NW >= a.n_items) return;   // padding workgroups (whole workgroups: no barrier is left waiting)
    const int lane = lane_id();
    const int wv = uniform_int(wave_id());
    const int r16 = lane & 15, p = lane >> 4, rr = r16 & 1;
    const int t0 = grp * TG;
    const int nb = m.K >> 8, nrec = (nb + 3) >> 2;
    int item = bx * NW + wv;
    const bool item_ok = item < a.n_items;   // a surplus wave of the last workgroup walks the last item again (it takes part in the
    item = item_ok ? item : a.n_items - 1;   // copies and barriers) and stores nothing
    // job of this item (explicit selects: indexing the kernel argument with a run-time value goes through scratch)
    int jb = 0;
    const uint8_t* r2 = m.job[0].w.r2;
    int M = m.job[0].w.M, epi = m.job[0].epi, it = item;
    if constexpr (!GU) {
        if (m.njobs > 1 && item >= m.job[1].pair0) { jb = 1; r2 = m.job[1].w.r2; M = m.job[1].w.M; epi = m.job[1].epi; it = item - m.job[1].pair0; }
        if (m.njobs > 2 && item >= m.job[2].pair0) { jb = 2; r2 = m.job[2].w.r2; M = m.job[2].w.M; epi = m.job[2].epi; it = item - m.job[2].pair0; }
    }
    (void)jb;
    const int n_units = GU ? M : (M + 1) >> 1;
    int unit = it * 8 + (r16 >> 1);
    unit = unit < n_units ? unit : n_units - 1;
    const uint8_t* ub = r2 + (size_t)unit * nrec * REC;
    const uint8_t* src = a.acts + (size_t)grp * nb * SB;

    const PgConst C = pg_consts(p);
    PgAcc<TYPE> acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int l = 0; l < 8; ++l)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[g].a[l][j] = 0.0f;
#pragma unroll
        for (int t = 0; t < (PgAcc<TYPE>::NM ? PgAcc<TYPE>::NM : 1); ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[g].m[t][j] = 0.0f;
    }
    // Stage b + 1 goes into the other LDS half by LDS-DMA while the waves compute on stage b; two weight blocks in flight per lane.
    // Per block: the DMA pieces, then the ring refill (younger: NRING loads may stay in flight when the stage must have landed),
    // compute, wait, barrier.  No branch anywhere in the loop body (see the note in the loop).
    static_assert(NW == kPgWaves, "the stage layout is cut for kPgWaves waves");
    constexpr int NRING = TYPE == GT_Q4_K ? 3 : (TYPE == GT_Q5_K ? 5 : 6);   // loads of one pg_load
    const int lane16 = lane * 16, lane4 = lane * 4;
    constexpr int VP = TG / 2 / NW;   // 1 KB value pieces of a stage per wave
#define PG_STAGE(SRC, DST) do { \
        _Pragma("unroll") \
        for (int c = 0; c < VP; ++c) glds16((SRC) + (size_t)(c * NW + wv) * 1024 + lane16, (DST) + (size_t)(c * NW + wv) * 1024); \
        _Pragma("unroll") \
        for (int c = 0; c < ST::TAILP; ++c) glds4((SRC) + ST::SUMS + (size_t)(c * NW + wv) * 256 + lane4, (DST) + ST::SUMS + (size_t)(c * NW + wv) * 256); } while (0)
    // Three LDS buffers (rotating byte offsets; all below 64 KB — round 6: M0 reaches all 160 KB, tools/experiments/lds_dma_reach.cpp; kernels_mm8.h uses it) and ONE barrier per step, in its
    // middle (behind AVX lane 3).  Stage b + 2 is requested in the second half of step b, waited for (vmcnt) right before the barrier
    // of step b + 1 and first read at the top of step b + 2:
    //   landing  an LDS-DMA takes ~1.1 us under load, about a step: waited for inside its own step it IS the step time;
    //   RAW      the barrier after the wait is not enough by itself on this hardware — the wave's vmcnt retires an LDS-DMA slightly
    //            before the bytes are visible to another wave's ds_read (seen as rare wrong stage data on the 32-layer model; guide:
    //            "read a staged buffer one phase AFTER the wait that retires it") — here half a step lies between that barrier and the read;
    //   WAR      the copy overwrites the buffer of stage b - 1; it is issued behind the barrier of step b, which a wave reaches only after
    //            it has finished step b - 1.
    // The wait leaves the younger requests in flight: this step's ring refill (NRING; memory operations retire in order).
    unsigned long long* tr = nullptr;
    if constexpr (TRACE) { if ((m.dbg & 32) && bx == 0 && grp == 0 && wv == 0 && lane == 0) tr = (unsigned long long*)(smem + 3 * SB); }
    pg_stamp<TRACE>(tr, 0);
    PG_STAGE(src, smem);
    PG_STAGE(src + (size_t)(1 < nb ? 1 : nb - 1) * SB, smem + (size_t)SB);
    // Weight ring: block b + 2 is requested while block b is used.  The requests run past the row's last block by up to two slots
    // (into the next unit's record; the arena is padded, engine.cc:upload_r2c4): no clamps, no guards — see the note in the loop.
    const PgLane LN = pg_lane<TYPE>(rr, p);
    PgRec<TYPE> ring0 = pg_load<TYPE, 0, 0>(ub, LN), ring1 = pg_load<TYPE, 0, 1>(ub, LN);
    vm_wait<2 * NRING>();
    __syncthreads();
    sleep_cycles<8>();   // the prologue's stand-in for the step between wait and first read
    __syncthreads();
    pg_stamp<TRACE>(tr, 1);
    const uint8_t* rec0 = ub;
    uint32_t o_cur = 0, o_nxt = SB, o_dma = 2 * SB;
    for (int b0 = 0; b0 < nb; b0 += 4) {
        // One basic block, no guard around a step or a copy.  With `if (b < nb)` the ring registers become phi nodes and hipcc copies the
        // freshly loaded slot at the loop end — behind an `s_waitcnt vmcnt(0)`; with any branch between the steps it SINKS the f32 chain
        // steps of all four blocks to the end of the loop body and spills every matrix product on the way.  So the last stage is copied
        // again instead of `if (b + 2 < nb)`, and the up to three surplus steps of the last round (K / 256 not a multiple
        // of 4) run with y.d = 0 on whatever the ring holds: acc + 0 * finite = acc bit for bit (acc is never -0).
#define PG_STEP(U, RING, RD, CB) do { \
            const int b = b0 + U; \
            const PgRec<TYPE> R = RING; \
            pg_stamp<TRACE>(tr, 8 + 6 * b); \
            const PgFeed F = {src + (size_t)(b + 2 < nb ? b + 2 : nb - 1) * SB, smem + o_dma, rec0, wv, lane16, lane4}; \
            pg_stamp<TRACE>(tr, 8 + 6 * b + 1); \
            pg_block<TYPE, TG, NRING, TRACE, RD, CB>(R, RING, F, LN, smem + o_cur, r16, p, acc, C, b < nb, tr, 8 + 6 * b); \
            { const uint32_t t = o_cur; o_cur = o_nxt; o_nxt = o_dma; o_dma = t; } \
            pg_stamp<TRACE>(tr, 8 + 6 * b + 4); \
            pg_stamp<TRACE>(tr, 8 + 6 * b + 5); } while (0)
        PG_STEP(0, ring0, 0, 2);
        PG_STEP(1, ring1, 0, 3);
        PG_STEP(2, ring0, 1, 0);
        PG_STEP(3, ring1, 1, 1);
#undef PG_STEP
#undef PG_STAGE
        rec0 += REC;
    }
    if constexpr (TRACE) { if (tr) { tr[2] = clock64_dev(); for (int i = 0; i < 120; ++i) ((unsigned long long*)m.dbg_sink)[i] = tr[i]; } }
    // hsum_float_8 (k_quants.c:90-97) and the min-term tree, in-lane; then the epilogues of the decode kernels, per token
    const int nt = a.n_tok - t0 < TG ? a.n_tok - t0 : TG;
    const int pos0 = (m.pos ? *m.pos : 0) + t0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float res[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float tot = ((acc[g].a[0][j] + acc[g].a[4][j]) + (acc[g].a[2][j] + acc[g].a[6][j])) +
                              ((acc[g].a[1][j] + acc[g].a[5][j]) + (acc[g].a[3][j] + acc[g].a[7][j]));
            if constexpr (TYPE == GT_Q4_K) res[j] = tot + ((acc[g].m[0][j] + acc[g].m[2][j]) + (acc[g].m[1][j] + acc[g].m[3][j]));
            else if constexpr (TYPE == GT_Q5_K) res[j] = tot + acc[g].m[0][j];
            else res[j] = tot;
        }
        if constexpr (GU) {   // fused matrix: even lane = gate row `unit`, odd lane = up row `unit`
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float up = lane_xor1(res[j]);
                const int t = g * 16 + 4 * p + j, row = it * 8 + (r16 >> 1);
                if (item_ok && rr == 0 && t < nt && row < M)
                    m.out[(size_t)(t0 + t) * a.ld_out + row] = f16_bits_to_f32(m.silu_tab[f32_to_f16_bits(res[j])]) * up;
            }
        } else {
            const int row = it * 16 + r16;
            const bool row_ok = item_ok && row < M;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = g * 16 + 4 * p + j;
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
