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
//                           out as compact images, two tokens per image:  q8 even[K/4] | q8 odd[K/4] | {yd even, yd odd}[K/32]  (words)
//   matvec_pf_kernel        grid (x, token-groups of 8): copies its 8 images into LDS, then wave w takes tiles
//                           w * gridDim.x + blockIdx.x + k * 16 * gridDim.x (low tile counts spread over the CUs first,
//                           then over the SIMDs of a CU); epilogues as in the decode kernels, per token.  Q8_0 / Q4_0 lane sums are
//                           4-element dots — nothing for a matrix core to contract.
//   embed / attention / falcon RoPE store   the decode kernels with a token index in blockIdx.y / blockIdx.z
#pragma once
#include "kernels_q32.h"

constexpr int kPfTokens = 8;    // tokens per workgroup (register accumulators per lane: 2 x 8)
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
constexpr int pf_act_words_q32(int K) { return ((K >> 2) + (K >> 5) + 3) & ~3; }

template <int MAXK>
__global__ void __launch_bounds__(1024) pf_quantize_q80_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ nw, int K,
                                                               int pro, float eps, int* __restrict__ acts, int act_words,
                                                               const float* __restrict__ nb_ = nullptr) {
    __shared__ ActLdsQ32<MAXK> L;
    const int t = (int)blockIdx.x, tid = (int)threadIdx.x;
    prologue_q8_0<MAXK>(L, x + (size_t)t * ldx, nw, K, pro, eps, nb_);   // nb_: LayerNorm bias (gpt2)
    // tokens are stored in pairs (the chunk kernel's two-wide float steps): q8 of the even token | q8 of the odd token | block scales
    // interleaved {even, odd} per block — 2 * act_words words per pair
    int* o = acts + (size_t)(t >> 1) * 2 * act_words;
    const int nq = K >> 2, nb = K >> 5, odd = t & 1;
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
