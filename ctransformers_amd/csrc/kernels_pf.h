// Prompt chunks, shared pieces and the Q8_0 / Q4_0 form: the mat-vec of the decode kernels applied to a chunk of tokens per launch,
// so that a weight tile is fetched and unpacked once for 8 tokens instead of once per token.  (K-quant weights take the f16
// matrix-core kernels of kernels_pg.h; their round-1 forms — dot4 and int8 matrix cores — were retired in round 2.)
//
// The arithmetic per (row, token) is the decode kernels' — the reference's per-block integer sums and its per-block
// fma in block order, then hsum_float_8 (kernels_q32.h has the citations) — so a token evaluated inside a chunk produces the
// same bits as the token evaluated alone.  A launch has tiles x token-groups units of work, enough for every wave to own
// (tile, 8 tokens) outright: it walks the tile's blocks in order with the accumulators in registers, no chain storage and no
// intra-workgroup synchronisation after the activations are in LDS.
//
//   pf_quantize_q80_kernel  one workgroup per token: (RMSNorm / LayerNorm ->) Q8_0 exactly as the decode prologue does it, written
//                           out as compact images — one per token, q8[K/4] | yd[K/32] words, for the matrix-core kernel; per token
//                           PAIR, q8 even | q8 odd | {yd even, yd odd} per block, for the dot4 kernel's two-wide steps
//   matvec_pfm_kernel       K <= 16384.  grid (x, token-groups of 8): copies its 8 images into LDS, then wave w takes tiles
//                           w * gridDim.x + blockIdx.x + k * 16 * gridDim.x (low tile counts spread over the CUs first, then over
//                           the SIMDs of a CU); the lane sums on v_mfma_i32_4x4x4_16b_i8 (K = 4 per block IS a lane sum: 256 of
//                           them per instruction); epilogues as in the decode kernels, per token
//   matvec_pf_kernel        wider rows (K <= 32768, 4 tokens per workgroup): the same walk with v_dot4_i32_i8
//   embed / attention / falcon RoPE store   the decode kernels with a token index in blockIdx.y / blockIdx.z
#pragma once
#include "kernels_q32.h"

constexpr int kPfTokens = 8;    // tokens per workgroup (register accumulators per lane: 2 x 8)
constexpr int kPfChunkFast = 512;   // chunk size where every site runs the order-free kernels (kernels_mm8.h): more token tiles per launch
constexpr int kPfChunk = 128;   // tokens per chunk_step (8 groups of 16 on the matrix-core path): the per-chunk launches of Wo,
                                // the quantize kernels and attention are shared by more tokens (64 -> 128: +11 %)

struct PfArgs {
    MatvecArgs m;        // jobs and epilogue operands; x / norm_w / pro are the quantize kernel's business
    const int* acts;     // n_tok activation images (pf_quantize_q80_kernel)
    int act_words;       // words per image (multiple of 4)
    int n_tok;           // tokens in this chunk
    int ld_out, ld_res, ld_q;   // element strides between the tokens' rows of m.out, m.res, m.q_f16
};

// ---- Q8_0 / Q4_0 weights (LAYOUT_G4, kernels_q32.h): Q8_0 activation images  q8[K/4] ([group][l][i] order) | yd[K/32] -----------
// Words per token image, padded to 16 (mod 64): the 16 lanes one LDS cycle of a ds_read_b128 serves are four AVX lanes l x the four tokens
// m of a token group (lane = 32 rg + 4 l + m), each reading the four words at m * act_words + 4 (8 g + l) — with images a multiple of 64
// words apart (K = 4096: 1152) the four tokens share their banks (a 4-way conflict on every read: a third of the kernel's time,
// profiles/r04_q80_chunk_ablations.txt); 16 (mod 64) gives the sixteen lanes sixteen different 4-bank slots.
constexpr int pf_act_words_q32(int K) {
    const int w = ((K >> 2) + (K >> 5) + 3) & ~3;
    return w + ((16 - (w & 63)) & 63);
}

template <int MAXK>
__global__ void __launch_bounds__(1024) pf_quantize_q80_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ nw, int K,
                                                               int pro, float eps, int* __restrict__ acts, int act_words,
                                                               const float* __restrict__ nb_ = nullptr, int paired = 1) {
    __shared__ ActLdsQ32<MAXK> L;
    const int t = (int)blockIdx.x, tid = (int)threadIdx.x;
    prologue_q8_0<MAXK>(L, x + (size_t)t * ldx, nw, K, pro, eps, nb_);   // nb_: LayerNorm bias (gpt2)
    // the images cover whole groups of four blocks: a row whose last group is incomplete (K % 128 != 0) gets zero blocks there, as the
    // weight records do (engine_load.h:upload_matrix) — the kernels then see a row of Kp elements whose tail contributes fma(0, 0, acc)
    const int Kp = (K + 127) & ~127;
    if (!paired) {   // one image per token, q8[Kp/4] | yd[Kp/32]: the matrix-core form (matvec_pfm_kernel) reads a token per lane
        int* o1 = acts + (size_t)t * act_words;
        for (int i = tid; i < (Kp >> 2); i += 1024) o1[i] = L.q8[i];
        for (int i = tid; i < (Kp >> 5); i += 1024) o1[(Kp >> 2) + i] = (int)f32_to_bits(L.yd[i]);
        return;
    }
    // tokens are stored in pairs (the chunk kernel's two-wide float steps): q8 of the even token | q8 of the odd token | block scales
    // interleaved {even, odd} per block — 2 * act_words words per pair
    int* o = acts + (size_t)(t >> 1) * 2 * act_words;
    const int nq = Kp >> 2, nb = Kp >> 5, odd = t & 1;
    for (int i = tid; i < nq; i += 1024) o[odd * nq + i] = L.q8[i];
    for (int i = tid; i < nb; i += 1024) o[2 * nq + 2 * i + odd] = (int)f32_to_bits(L.yd[i]);
}

// q32_tile_dot for the nt tokens in LDS: one fma per 32-block, AVX lane and token (ggml.c:3321 / :2428, AVX2 forms), the
// weight group fetched (and its nibbles turned into signed bytes) once for all tokens.  res[t] valid in every lane of the row.
// The chunk kernel is VALU-bound (one (dot4, int->float, scale product, fma) per 4 multiply-adds), so the step is written for the
// fewest instructions, two tokens at a time:
//   * the dot carries the addend 0x4B400000 (VOP3P form, scalar operand): its result, read as a float, IS 12582912 + sumi
//     (|sumi| <= 65024 < 2^22 keeps the sum in the binade whose ulp is 1), and one v_pk_add_f32 of -12582912 gives the two exact
//     (float)sumi — no v_cvt, no zero-initialised accumulator;
//   * Q4_0: nibble - 8 as a signed byte = ((nibble + 0x78) ^ 0x80) per byte, two operations per dword ONCE per weight group instead
//     of a second dot per token (the reference's sum(nib * y) - 8 sum(y) is this integer);
//   * v_pk_mul_f32 for the two block-scale products fp16(x.d) * fp16(y.d), v_pk_fma_f32 for the two chain steps.
// Every float operation is the reference's, on the reference's operands, in its order: per token, block after block.
template <int TYPE, int TB>
DEV void pf_tile_q32(const uint8_t* __restrict__ tile, int ng, const int* __restrict__ lds, int act_words, int K, int nt, int lane,
                     float (&res)[TB]) {
    static_assert(TB % 2 == 0, "tokens are taken in pairs");
    constexpr int REC = TYPE == GT_Q8_0 ? kRecQ8_0 : kRecQ4_0;
    constexpr int PF = 4;
    const int r = lane >> 3, p3 = lane & 7, l = ((p3 & 1) << 2) | (p3 & 2) | (p3 >> 2);   // AVX lane = bitrev3(position), see q32_tile_dot
    const uint32_t qoff = TYPE == GT_Q8_0 ? (uint32_t)(r * 8 + l) * 16u : (uint32_t)(r * 4 + (l & 3)) * 16u;
    const uint32_t doff = (TYPE == GT_Q8_0 ? 1024u : 512u) + (uint32_t)r * 8u;
    const int sh = (TYPE == GT_Q4_0 && l >= 4) ? 4 : 0;
    const int nq = K >> 2;
    const int kMagic = 0x4B400000;                       // 12582912.0f = 1.5 * 2^23
    const F32x2 unmagic = pk2(-12582912.0f, -12582912.0f);
    F32x2 acc[TB / 2];
#pragma unroll
    for (int t = 0; t < TB / 2; ++t) acc[t] = pk2(0.0f, 0.0f);
    u32x4 qv[PF];
    uint64_t dv[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int g = u < ng ? u : ng - 1;
        qv[u] = ld_stream16(tile + (size_t)g * REC + qoff);
        dv[u] = *(const uint64_t*)(tile + (size_t)g * REC + doff);
    }
    for (int g0 = 0; g0 < ng; g0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int g = g0 + u;
            const u32x4 q = qv[u];
            const uint64_t dd = dv[u];
            {
                const int gn = (g + PF < ng) ? g + PF : ng - 1;
                qv[u] = ld_stream16(tile + (size_t)gn * REC + qoff);
                dv[u] = *(const uint64_t*)(tile + (size_t)gn * REC + doff);
            }
            if (g >= ng) continue;
            int w[8];
            F32x2 dw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w[i] = TYPE == GT_Q8_0 ? (int)q[i] : (int)((((q[i] >> sh) & 0x0F0F0F0Fu) + 0x78787878u) ^ 0x80808080u);
                w[4 + i] = w[i];
                const float d = f16_bits_to_f32((uint16_t)((dd >> (16 * i)) & 0xFFFFu));
                dw[i] = pk2(d, d);
            }
#pragma unroll
            for (int tp = 0; tp < TB / 2; ++tp) {
                if (2 * tp < nt) {   // an odd chunk tail computes its unused second token on the (stale) image behind it: never stored
                    const int* img = lds + tp * 2 * act_words;   // pair image: q8 even | q8 odd | {yd even, yd odd} per block
                    const u32x4 y0 = *(const u32x4*)(img + (g * 8 + l) * 4), y1 = *(const u32x4*)(img + nq + (g * 8 + l) * 4);
                    const u32x4 yda = *(const u32x4*)(img + 2 * nq + g * 8), ydb = *(const u32x4*)(img + 2 * nq + g * 8 + 4);
                    const int y[8] = {(int)y0[0], (int)y0[1], (int)y0[2], (int)y0[3], (int)y1[0], (int)y1[1], (int)y1[2], (int)y1[3]};
                    const F32x2 ydp[4] = {pk2(bits_to_f32(yda[0]), bits_to_f32(yda[1])), pk2(bits_to_f32(yda[2]), bits_to_f32(yda[3])),
                                          pk2(bits_to_f32(ydb[0]), bits_to_f32(ydb[1])), pk2(bits_to_f32(ydb[2]), bits_to_f32(ydb[3]))};
                    int s[8];
                    dot4x8_acc(s, w, y, kMagic);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const F32x2 f = pk_add_f32(pk2(bits_to_f32((uint32_t)s[i]), bits_to_f32((uint32_t)s[4 + i])), unmagic);
                        acc[tp] = pk_fma_f32(pk_mul_f32(dw[i], ydp[i]), f, acc[tp]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TB / 2; ++t) {
#ifdef CT_EMU
        res[2 * t] = hsum8_exact_dpp(acc[t].x);
        res[2 * t + 1] = hsum8_exact_dpp(acc[t].y);
#else
        res[2 * t] = hsum8_exact_dpp(acc[t][0]);
        res[2 * t + 1] = hsum8_exact_dpp(acc[t][1]);
#endif
    }
}

// ---- matrix-core form: the lane sums on v_mfma_i32_4x4x4_16b_i8 -------------------------------------------------------------------
// K = 4 per block is exactly the reference's lane sum (four products of one AVX lane), and the instruction runs sixteen independent
// 4x4 blocks: block b = (AVX lane l, row group rg), A = the four rows 4rg..4rg+3 of the tile (bytes 4l..4l+3 of the 32-block),
// B = four tokens, C = 0x4B400000 so that the result bits read as floats are 1.5 * 2^23 + sumi.  One instruction = 256 lane sums
// (8 rows x 4 tokens x 8 AVX lanes of one 32-block); lane 32rg + 4l + m then owns the accumulators of rows 4rg..4rg+3, token
// 4tg + m (tg = 0, 1: two instructions per 32-block for the 8 tokens of the workgroup) and AVX lane l, and runs the chain two rows
// at a time: v_pk_add_f32 (-1.5 * 2^23), v_pk_mul_f32 (the two scale products), v_pk_fma_f32.  hsum_float_8 over l = lane bits
// 2..4, in the reference's order (l^4, l^2, l^1).  Same floats, same order as pf_tile_q32: per token, block after block.
// NTG token groups of four per workgroup (TB = 4 NTG tokens; 2 in the product: 16 and 32 tokens per workgroup — 512 threads for
// the register budget — measured no faster on MI355X, the kernel is bound by instruction issue, not by re-reading the weights).
template <int NTG> struct PfmAcc { F32x2 a[NTG][2]; };   // [token group][row pair]

template <int TYPE, int NTG, int PF = 3>
DEV void pf_tile_q32m(const uint8_t* __restrict__ tile, int ng, const int* __restrict__ lds, int act_words, int K, int nt, int lane,
                      float (&res)[NTG][4]) {
    constexpr int REC = TYPE == GT_Q8_0 ? kRecQ8_0 : kRecQ4_0;
    const int m = lane & 3, l = (lane >> 2) & 7, rg = lane >> 5;
    const int ra = 4 * rg + m;                                           // the row this lane feeds as A
    const uint32_t qoff = TYPE == GT_Q8_0 ? (uint32_t)(ra * 8 + l) * 16u : (uint32_t)(ra * 4 + (l & 3)) * 16u;
    const uint32_t doff = (TYPE == GT_Q8_0 ? 1024u : 512u) + (uint32_t)ra * 8u;   // the four fp16 block scales of row `ra` in this group
    const int sh = (TYPE == GT_Q4_0 && l >= 4) ? 4 : 0;
    const int nq = K >> 2;
    i32x4 magic;
    magic[0] = magic[1] = magic[2] = magic[3] = 0x4B400000;
    f32x4 zero4;
    zero4[0] = zero4[1] = zero4[2] = zero4[3] = 0.0f;
    const F32x2 unmagic = pk2(-12582912.0f, -12582912.0f);
    PfmAcc<NTG> acc;
#pragma unroll
    for (int tg = 0; tg < NTG; ++tg) acc.a[tg][0] = acc.a[tg][1] = pk2(0.0f, 0.0f);
    // weight dwords: a ring PF groups deep (HBM latency); this lane's row scales (8 bytes) share the record's last cache line with the
    // other rows' and are requested one group ahead
    u32x4 qv[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int g = u < ng ? u : ng - 1;
        qv[u] = ld_stream16(tile + (size_t)g * REC + qoff);
    }
    u32x2 sc_n = ld_stream8(tile + doff);
    const int* img0 = lds + m * act_words;          // token m of group 0; group tg is 4 tg images further
    (void)nt;   // chunk tails run all eight token slots (the images behind the tail are stale, their results are never stored): no
                // branch in the block step, the two groups' matrix instructions interleave
    for (int g0 = 0; g0 < ng; g0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int g = g0 + u;
            const u32x4 q = qv[u];
            const u32x2 sc = sc_n;
            {   // the next group's scales BEFORE the ring refill (memory operations retire in order: requested behind the refill, waiting
                // for them would drain the refill too)
                const int g1 = (g + 1 < ng) ? g + 1 : ng - 1;
                sc_n = ld_stream8(tile + (size_t)g1 * REC + doff);
                const int gn = (g + PF < ng) ? g + PF : ng - 1;
                qv[u] = ld_stream16(tile + (size_t)gn * REC + qoff);
            }
            if (g >= ng) continue;
            // The block-scale products fp16(x.d) * fp16(y.d) of rows 4rg .. 4rg+3 x token m come out of the f32 matrix core (K = 1: a
            // rank-1 update with C = 0; the product of two fp16 values is exact in f32, so these ARE the reference's products): this lane
            // gives the scale of ITS row as A and its token's y.d as B and receives the four rows' products for its token — no broadcast
            // of y.d, four conversions per group instead of sixteen, and the packed multiplies leave the vector pipe (round 3: 10.5 vector
            // instructions per matrix instruction, of which 6 the chain; here 5).
            float xd[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xd[i] = f16_bits_to_f32((uint16_t)(sc[i >> 1] >> ((i & 1) * 16)));
            int w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                w[i] = TYPE == GT_Q8_0 ? (int)q[i] : (int)((((q[i] >> sh) & 0x0F0F0F0Fu) + 0x78787878u) ^ 0x80808080u);
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) {
                // two token groups between scheduling fences: their matrix instructions interleave, and no more than two groups' operands
                // and products are live at a time (left alone hipcc interleaves all NTG groups and spills: 67 registers at NTG = 8)
                if (NTG > 2 && (tg & 1) == 0 && tg > 0) sched_fence();
                const int* img = img0 + 4 * tg * act_words;
                const u32x4 y = *(const u32x4*)(img + (g * 8 + l) * 4);
                const u32x4 yd = *(const u32x4*)(img + nq + g * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const i32x4 d = mfma_i8_4x4x4(w[i], (int)y[i], magic);
                    const f32x4 pr = mfma_f32_4x4x1(xd[i], bits_to_f32(yd[i]), zero4);
                    const F32x2 f01 = pk_add_f32(pk2(bits_to_f32((uint32_t)d[0]), bits_to_f32((uint32_t)d[1])), unmagic);
                    const F32x2 f23 = pk_add_f32(pk2(bits_to_f32((uint32_t)d[2]), bits_to_f32((uint32_t)d[3])), unmagic);
                    acc.a[tg][0] = pk_fma_f32(pk2(pr[0], pr[1]), f01, acc.a[tg][0]);
                    acc.a[tg][1] = pk_fma_f32(pk2(pr[2], pr[3]), f23, acc.a[tg][1]);
                }
            }
            if (NTG > 2) sched_fence();
        }
    }
#pragma unroll
    for (int tg = 0; tg < NTG; ++tg)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#ifdef CT_EMU
            float x = (i & 1) ? acc.a[tg][i >> 1].y : acc.a[tg][i >> 1].x;
#else
            float x = acc.a[tg][i >> 1][i & 1];
#endif
            x = x + __shfl_xor(x, 16);   // (x0 + x4), (x1 + x5), (x2 + x6), (x3 + x7)      hsum_float_8, k_quants.c:90-97
            x = x + __shfl_xor(x, 8);    // (x0 + x4) + (x2 + x6), (x1 + x5) + (x3 + x7)
            x = x + __shfl_xor(x, 4);    // the sum
            res[tg][i] = x;
        }
}

// grid (x, token-groups of 8), 1024 threads: as matvec_pf_kernel, the tile product on the matrix cores.  A lane with l == 0 owns
// rows 4rg..4rg+3 of the tile for tokens m, 4 + m, ... of the group.
// NTG token groups of four per workgroup: what a CU loads per pass over the weights is shared by 4 NTG tokens (the host picks 4, 3 or 2 by
// what fits the LDS).  Where the kernel's time goes (round 4, profiles/r04_q80_chunk_ablations.txt, tools/experiments/mfma_4x4_rate.cpp):
// a block step of a wave — two matrix instructions + the four packed chain instructions — issues in 36.6 cycles per SIMD at four waves
// (the 4x4 matrix instructions cost 6.2 cycles each and do not hide behind the vector pipe), 806 000 steps per SIMD for a 128-token
// chunk of the 7B = 14.7 ms of the 24 ms; the two ds_read_b128 per four steps, the s_nop behind each matrix result and the weight ring
// are the rest.  Without the block math the chunk takes 15 ms, without weight reloads 22 ms.
template <bool GU, int NTG, int NT = 1024>
__global__ void __launch_bounds__(NT) matvec_pfm_kernel(const PfArgs a) {
    constexpr int TB = 4 * NTG, NW = NT / 64;
    CT_DYN_SMEM(smem_raw);
    int* lds = reinterpret_cast<int*>(smem_raw);
    const MatvecArgs& m = a.m;
    const int tid = (int)threadIdx.x, lane = lane_id();
    const int wv = uniform_int(wave_id());
    const int t0 = (int)blockIdx.y * TB;
    const int nt = a.n_tok - t0 < TB ? a.n_tok - t0 : TB;
    {
        const u32x4* src = (const u32x4*)(a.acts + (size_t)t0 * a.act_words);
        const int n16 = TB * (a.act_words >> 2);
        for (int i0 = 0; i0 < n16; i0 += 4 * NT) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * NT + tid; v[u] = ld16(src + (i < n16 ? i : n16 - 1)); }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * NT + tid; ((u32x4*)lds)[i < n16 ? i : n16 - 1] = v[u]; }
        }
    }
    __syncthreads();
    const int pos0 = (m.pos ? *m.pos : 0) + t0;
    const int GX = (int)gridDim.x;
    const int ng = m.K >> 7;
    const int tm = lane & 3, rg = lane >> 5;
    const bool own_lane = ((lane >> 2) & 7) == 0;
    for (int item = wv * GX + (int)blockIdx.x; item < m.n_pairs; item += NW * GX) {
        int j = 0;
        if (!GU) {
            if (m.njobs > 1 && item >= m.job[1].pair0) j = 1;
            if (m.njobs > 2 && item >= m.job[2].pair0) j = 2;
        }
        const int tile = GU ? item : item - m.job[j].pair0;
        const DevMat& wj = m.job[j].w;
        const size_t rec = wj.type == GT_Q8_0 ? kRecQ8_0 : kRecQ4_0;
        float res[NTG][4];
        constexpr int PFD = (NT == 1024 && NTG >= 4 && !GU) ? 1 : ((NT == 1024 && NTG >= 4) || (GU && NTG == 2) ? 2 : 3);   // the register budget of sixteen waves
        if (wj.type == GT_Q8_0) pf_tile_q32m<GT_Q8_0, NTG, PFD>(wj.p[0] + (size_t)tile * ng * rec, ng, lds, a.act_words, m.K, nt, lane, res);
        else pf_tile_q32m<GT_Q4_0, NTG, PFD>(wj.p[0] + (size_t)tile * ng * rec, ng, lds, a.act_words, m.K, nt, lane, res);
        if constexpr (GU) {
            // gate + up: SiLU(gate) of the owning lanes waits in LDS while the wave walks the up tile (held in registers, the two result
            // sets and the accumulators do not fit 128 registers beyond 8 tokens); the epilogue below then sees `res` = the up rows
            float* mine = reinterpret_cast<float*>(lds + (size_t)TB * a.act_words) + ((size_t)wv * 8 + (size_t)(4 * rg + tm)) * (NTG * 4);
            if (own_lane) {
#pragma unroll
                for (int tg = 0; tg < NTG; ++tg)
#pragma unroll
                    for (int i = 0; i < 4; ++i) mine[tg * 4 + i] = f16_bits_to_f32(m.silu_tab[f32_to_f16_bits(res[tg][i])]);
            }
            const DevMat& wu = m.job[1].w;
            const size_t recu = wu.type == GT_Q8_0 ? kRecQ8_0 : kRecQ4_0;
            if (wu.type == GT_Q8_0) pf_tile_q32m<GT_Q8_0, NTG, PFD>(wu.p[0] + (size_t)tile * ng * recu, ng, lds, a.act_words, m.K, nt, lane, res);
            else pf_tile_q32m<GT_Q4_0, NTG, PFD>(wu.p[0] + (size_t)tile * ng * recu, ng, lds, a.act_words, m.K, nt, lane, res);
            if (own_lane) {
#pragma unroll
                for (int tg = 0; tg < NTG; ++tg)
#pragma unroll
                    for (int i = 0; i < 4; ++i) res[tg][i] = mine[tg * 4 + i] * res[tg][i];   // silu_table[fp16(gate)] * up
            }
        }
        const int epi = m.job[j].epi;
#pragma unroll
        for (int tg = 0; tg < NTG; ++tg) {
            const int t = 4 * tg + tm;
            if (t >= nt || !own_lane) continue;
            int tok = t0 + t;
            const int pos = pos0 + t;
#ifndef CT_EMU
            asm volatile("" : "+v"(tok));   // the row addresses are formed here, not hoisted above the tile loop (where they would be spilled)
#endif
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = tile * 8 + 4 * rg + i;
                if (row >= wj.M) continue;
                const float v = res[tg][i];
                if constexpr (GU) {
                    m.out[(size_t)tok * a.ld_out + row] = v;
                } else if (epi == EPI_ADD) {
                    m.out[(size_t)tok * a.ld_out + row] = v + m.res[(size_t)tok * a.ld_res + row];
                } else if (epi == EPI_STORE) {
                    m.out[(size_t)tok * a.ld_out + row] = v;
                } else if (epi == EPI_GELU) {
                    m.out[(size_t)tok * a.ld_out + row] = f16_bits_to_f32(m.gelu_tab[f32_to_f16_bits(v)]);
                } else if (epi == EPI_ADD2) {
                    m.out[(size_t)tok * a.ld_out + row] = (v + m.res[(size_t)tok * a.ld_res + row]) + m.res2[(size_t)tok * a.ld_res + row];
                } else if (epi == EPI_BIAS_STORE) {
                    m.out[(size_t)tok * a.ld_out + row] = m.bias[row] + v;
                } else if (epi == EPI_BIAS_ADD) {
                    m.out[(size_t)tok * a.ld_out + row] = (m.bias[row] + v) + m.res[(size_t)tok * a.ld_res + row];
                } else if (epi == EPI_BIAS_GELU) {
                    m.out[(size_t)tok * a.ld_out + row] = f16_bits_to_f32(m.gelu_tab[f32_to_f16_bits(m.bias[row] + v)]);
                } else if (epi == EPI_V) {
                    m.vcache[(size_t)row * m.v_stride + pos] = f32_to_f16_bits(v);
                } else {   // EPI_ROPE_Q / EPI_ROPE_K (normal mode, ggml.c:12522-12539): rows 2k, 2k+1 are registers i, i ^ 1 of this lane
                    const float other = res[tg][i ^ 1];
                    const int ip = (row % m.head_dim) >> 1;
                    const float cs = m.rope_cs[((size_t)pos * (m.head_dim >> 1) + ip) * 2 + 0];
                    const float sn = m.rope_cs[((size_t)pos * (m.head_dim >> 1) + ip) * 2 + 1];
                    const float o = (i & 1) ? fmaf(v, cs, other * sn) : fmaf(v, cs, -(other * sn));
                    if (epi == EPI_ROPE_Q) m.q_f16[(size_t)tok * a.ld_q + row] = f32_to_f16_bits(o);
                    else m.kcache[kcache_off(pos, row, m.head_dim, m.n_ctx)] = f32_to_f16_bits(o);
                }
            }
        }
    }
}

template <int TB>
DEV void pf_tile_any(int type, const uint8_t* __restrict__ w0, int tile, const int* __restrict__ lds, int act_words, int K,
                     int nt, const LaneGeom& G, float (&res)[TB]) {
    const int ng = K >> 7;
    const int lane = G.r * 8 + G.g;
    if (type == GT_Q8_0) pf_tile_q32<GT_Q8_0, TB>(w0 + (size_t)tile * ng * kRecQ8_0, ng, lds, act_words, K, nt, lane, res);
    else pf_tile_q32<GT_Q4_0, TB>(w0 + (size_t)tile * ng * kRecQ4_0, ng, lds, act_words, K, nt, lane, res);
}

template <int TB, bool GU>
__global__ void __launch_bounds__(1024) matvec_pf_kernel(const PfArgs a) {
    CT_DYN_SMEM(smem_raw);
    int* lds = reinterpret_cast<int*>(smem_raw);
    const MatvecArgs& m = a.m;
    const int tid = (int)threadIdx.x, lane = lane_id();
    const int wv = uniform_int(wave_id());
    const LaneGeom G = lane_geom(lane);
    const int t0 = (int)blockIdx.y * TB;
    const int nt = a.n_tok - t0 < TB ? a.n_tok - t0 : TB;
    {   // this token group's images, contiguous in global memory
        // all TB images (the scratch buffer always holds kPfChunk; images past n_tok are stale and never stored from)
        const u32x4* src = (const u32x4*)(a.acts + (size_t)t0 * a.act_words);
        const int n16 = TB * (a.act_words >> 2);
        for (int i0 = 0; i0 < n16; i0 += 4 * 1024) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * 1024 + tid; v[u] = ld16(src + (i < n16 ? i : n16 - 1)); }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * 1024 + tid; ((u32x4*)lds)[i < n16 ? i : n16 - 1] = v[u]; }
        }
    }
    __syncthreads();
    const int pos0 = (m.pos ? *m.pos : 0) + t0;
    const int GX = (int)gridDim.x;
    const bool own_lane = G.g == 0;
    for (int item = wv * GX + (int)blockIdx.x; item < m.n_pairs; item += 16 * GX) {
        if constexpr (GU) {
            float gate[TB], up[TB];
            pf_tile_any<TB>(m.job[0].w.type, m.job[0].w.p[0], item, lds, a.act_words, m.K, nt, G, gate);
            pf_tile_any<TB>(m.job[1].w.type, m.job[1].w.p[0], item, lds, a.act_words, m.K, nt, G, up);
            const int row = item * 8 + G.r;
            const bool own = own_lane && row < m.job[0].w.M;
#pragma unroll
            for (int t = 0; t < TB; ++t)
                if (t < nt && own) m.out[(size_t)(t0 + t) * a.ld_out + row] = f16_bits_to_f32(m.silu_tab[f32_to_f16_bits(gate[t])]) * up[t];
        } else {
            int j = 0;
            if (m.njobs > 1 && item >= m.job[1].pair0) j = 1;
            if (m.njobs > 2 && item >= m.job[2].pair0) j = 2;
            const int tile = item - m.job[j].pair0;
            float res[TB];
            pf_tile_any<TB>(m.job[j].w.type, m.job[j].w.p[0], tile, lds, a.act_words, m.K, nt, G, res);
            const int row = tile * 8 + G.r;
            const bool own = own_lane && row < m.job[j].w.M;
            const int epi = m.job[j].epi;
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                if (t >= nt) continue;
                const int tok = t0 + t, pos = pos0 + t;
                if (epi == EPI_ADD) {
                    if (own) m.out[(size_t)tok * a.ld_out + row] = res[t] + m.res[(size_t)tok * a.ld_res + row];
                } else if (epi == EPI_STORE) {
                    if (own) m.out[(size_t)tok * a.ld_out + row] = res[t];
                } else if (epi == EPI_GELU) {         // falcon
                    if (own) m.out[(size_t)tok * a.ld_out + row] = f16_bits_to_f32(m.gelu_tab[f32_to_f16_bits(res[t])]);
                } else if (epi == EPI_ADD2) {
                    if (own) m.out[(size_t)tok * a.ld_out + row] = (res[t] + m.res[(size_t)tok * a.ld_res + row]) + m.res2[(size_t)tok * a.ld_res + row];
                } else if (epi == EPI_BIAS_STORE) {   // gpt2 (row biases): same operand order as the decode kernels
                    if (own) m.out[(size_t)tok * a.ld_out + row] = m.bias[row] + res[t];
                } else if (epi == EPI_BIAS_ADD) {
                    if (own) m.out[(size_t)tok * a.ld_out + row] = (m.bias[row] + res[t]) + m.res[(size_t)tok * a.ld_res + row];
                } else if (epi == EPI_BIAS_GELU) {
                    if (own) m.out[(size_t)tok * a.ld_out + row] = f16_bits_to_f32(m.gelu_tab[f32_to_f16_bits(m.bias[row] + res[t])]);
                } else if (epi == EPI_V) {
                    if (own) m.vcache[(size_t)row * m.v_stride + pos] = f32_to_f16_bits(res[t]);
                } else {   // EPI_ROPE_Q / EPI_ROPE_K (normal mode, ggml.c:12522-12539): rows 2i, 2i+1 are adjacent rows of the tile
                    const float other = lane_xor8(res[t]);
                    const int ip = (row % m.head_dim) >> 1;
                    const float cs = m.rope_cs[((size_t)pos * (m.head_dim >> 1) + ip) * 2 + 0];
                    const float sn = m.rope_cs[((size_t)pos * (m.head_dim >> 1) + ip) * 2 + 1];
                    const float o = (G.r & 1) ? fmaf(res[t], cs, other * sn) : fmaf(res[t], cs, -(other * sn));
                    if (own) {
                        if (epi == EPI_ROPE_Q) m.q_f16[(size_t)tok * a.ld_q + row] = f32_to_f16_bits(o);
                        else m.kcache[kcache_off(pos, row, m.head_dim, m.n_ctx)] = f32_to_f16_bits(o);
                    }
                }
            }
        }
    }
}

__global__ void advance_state_n_kernel(int* state, int n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        state[0] += n;
        state[1] += n;
    }
}

