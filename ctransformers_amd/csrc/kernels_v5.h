// Mat-vec, generation 5 (and the arithmetic building blocks shared with generation 6, kernels_v6.h, and the wide-K kernel,
// kernels_ks.h): one 1024-thread workgroup per CU, the K-blocks of an 8-row tile split over the 16 waves, weight TYPE a
// compile-time parameter.
//   round = T tiles; per wave T*S block steps -> chain storage in LDS (tail blocks are clamped: recomputing the last
//   block writes identical values), one barrier, then T waves replay one f32 chain each (rotating) while the rest start
//   the next round from the other half of the chain storage.
// Today it serves the launches with at most two tiles per workgroup (one round: Wo); everything else runs generation 6.
#pragma once
#include "kernels_exact.h"

struct UnitInfo {            // one 8-row tile of a launch, wave-uniform
    int valid, type, nb, M, tile, j;
    const uint8_t* base;     // first record of the tile
    uint32_t rec;
};

// Chain storage of generation 5: per (block, lane) `(float)sumi[l]`, per (block, row) the block scales already multiplied out
// (D = y.d * fp16(x.d), DM = -y.d * fp16(x.dmin)): those products are order-free, so the block step that holds the
// header does them and the replay — the serial part of a round — is left with LDS fetches and the fma chain only.
template <int MAXNB> struct ChainBuf5 {
    float S[MAXNB][64];
    float D[MAXNB][8];
    float DM[MAXNB][8];
    float PM[MAXNB][32];
};

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

// Integer work of one block -> chain storage (see block_to_chain4 for the arithmetic; this is its per-type form).
template <int TYPE, int MAXK, int MAXNB>
DEV void img_to_chain(const BlkImg<TYPE>& R, int b, const ActLdsX<MAXK>& L, ChainBuf5<MAXNB>& C, int lane, const LaneGeom& G) {
    const int c = G.c;
    if constexpr (TYPE == GT_Q4_K || TYPE == GT_Q5_K) {
        const int* alo = &L.q8[b * 64 + G.a45];
        const int* ahi = alo + 8;
        const uint32_t lo_w = c < 2 ? R.hdr[1] : (c == 2 ? R.hdr[2] : R.hdr[3]);
        const uint32_t hi_w = c < 2 ? R.hdr[2] : R.hdr[3];
        const uint32_t x = alignbit32(hi_w, lo_w, (uint32_t)((24 * c) & 31));
        const int sc_lo = (int)(x & 63u), sc_hi = (int)bfe32(x, 6, 6), m_lo = (int)bfe32(x, 12, 6), m_hi = (int)bfe32(x, 18, 6);
        int part[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t lo = R.qs[k] & 0x0F0F0F0Fu;
            uint32_t hi = (R.qs[k] >> 4) & 0x0F0F0F0Fu;
            if constexpr (TYPE == GT_Q5_K) {
                lo |= ((R.qh[k] >> (2 * c)) & 0x01010101u) << 4;
                hi |= ((R.qh[k] >> (2 * c + 1)) & 0x01010101u) << 4;
            }
            part[k] = mul24(sc_lo, sdot4((int)lo, alo[k], 0)) + mul24(sc_hi, sdot4((int)hi, ahi[k], 0));
        }
        C.S[b][lane] = (float)quad_transpose_reduce_dpp(part[0], part[1], part[2], part[3], c);
        int prod = mul24(m_lo, L.sb[b * 8 + 2 * c]) + mul24(m_hi, L.sb[b * 8 + 2 * c + 1]);
        if constexpr (TYPE == GT_Q5_K) {
            if (G.h != 0) prod = 0;
            prod += lane_xor2(prod);
            prod += lane_xor4(prod);
        }
        if (G.h == 0) C.PM[b][G.r * 4 + c] = (float)prod;
        if (G.g == 0) {
            const float yd = L.yd[b];
            C.D[b][G.r] = yd * f16_bits_to_f32((uint16_t)(R.hdr[0] & 0xFFFF));
            C.DM[b][G.r] = -yd * f16_bits_to_f32((uint16_t)(R.hdr[0] >> 16));
        }
    } else {
        const int n = G.g >> 2;
        const int* alo = &L.q8[b * 64 + G.a6];
        const int* ahi = alo + 16;
        const uint32_t w_lo = n ? R.sc[2] : R.sc[0];
        const uint32_t w_hi = n ? R.sc[3] : R.sc[1];
        const int sc_lo = (int)(int8_t)((w_lo >> G.sc_sh6) & 0xFF);
        const int sc_hi = (int)(int8_t)((w_hi >> G.sc_sh6) & 0xFF);
        int part[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = (R.ql[k] & 0x0F0F0F0Fu) | (((R.qh[k] >> G.s_lo6) & 0x03030303u) << 4);
            const uint32_t hi = ((R.ql[k] >> 4) & 0x0F0F0F0Fu) | (((R.qh[k] >> G.s_hi6) & 0x03030303u) << 4);
            const int dl = sdot4((int)lo, alo[k], sdot4((int)0xE0E0E0E0u, alo[k], 0));
            const int dh = sdot4((int)hi, ahi[k], sdot4((int)0xE0E0E0E0u, ahi[k], 0));
            part[k] = mul24(sc_lo, dl) + mul24(sc_hi, dh);
        }
        C.S[b][lane] = (float)quad_transpose_reduce_dpp(part[0], part[1], part[2], part[3], c);
        if (G.g == 0) C.D[b][G.r] = L.yd[b] * f16_bits_to_f32((uint16_t)(R.d & 0xFFFF));
    }
}

template <int TYPE, int MAXK, int MAXNB>
DEV float chain_typed(int nb, const ActLdsX<MAXK>& L, const ChainBuf5<MAXNB>& C, int lane, const LaneGeom& G) {
    float acc = 0.0f, accm = 0.0f;
    constexpr bool mins = TYPE != GT_Q6_K;
    constexpr int CH = 8;   // operands of CH blocks are fetched before the dependent fma chain starts
    for (int b0 = 0; b0 < nb; b0 += CH) {
        float dv[CH], sv[CH], mv[CH], pv[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int b = (b0 + u < nb) ? b0 + u : nb - 1;
            dv[u] = C.D[b][G.r];
            sv[u] = C.S[b][lane];
            if constexpr (mins) {
                mv[u] = C.DM[b][G.r];
                pv[u] = C.PM[b][G.r * 4 + G.c];
            }
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (b0 + u < nb) {
                acc = fmaf(dv[u], sv[u], acc);
                if constexpr (mins) accm = fmaf(mv[u], pv[u], accm);   // lanes with h == 1 carry garbage here; never read
            }
        }
    }
    const float tot = hsum8_exact_dpp(acc);
    if constexpr (!mins) return tot;
    if constexpr (TYPE == GT_Q4_K) {
        const float wsum = accm + lane_xor4(accm);
        accm = wsum + lane_xor2(wsum);
    }
    accm = __shfl(accm, lane & ~7);   // lane g == 0 of the row (h == 0, c == 0)
    return tot + accm;
}

// Two independent chains (the gate and the up tile of one item) replayed together: same arithmetic per chain as
// chain_typed, but the LDS fetches and the two dependent fma sequences interleave instead of running back to back.
template <int TYPE, int MAXK, int MAXNB>
DEV void chain_typed2(int nb, const ActLdsX<MAXK>& L, const ChainBuf5<MAXNB>& C0, const ChainBuf5<MAXNB>& C1, int lane,
                      const LaneGeom& G, float& out0, float& out1) {
    float acc0 = 0.0f, accm0 = 0.0f, acc1 = 0.0f, accm1 = 0.0f;
    constexpr bool mins = TYPE != GT_Q6_K;
    constexpr int CH = 8;
    for (int b0 = 0; b0 < nb; b0 += CH) {
        float dv0[CH], sv0[CH], mv0[CH], pv0[CH], dv1[CH], sv1[CH], mv1[CH], pv1[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int b = (b0 + u < nb) ? b0 + u : nb - 1;
            dv0[u] = C0.D[b][G.r];
            dv1[u] = C1.D[b][G.r];
            sv0[u] = C0.S[b][lane];
            sv1[u] = C1.S[b][lane];
            if constexpr (mins) {
                mv0[u] = C0.DM[b][G.r];
                mv1[u] = C1.DM[b][G.r];
                pv0[u] = C0.PM[b][G.r * 4 + G.c];
                pv1[u] = C1.PM[b][G.r * 4 + G.c];
            }
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (b0 + u < nb) {
                acc0 = fmaf(dv0[u], sv0[u], acc0);
                acc1 = fmaf(dv1[u], sv1[u], acc1);
                if constexpr (mins) {
                    accm0 = fmaf(mv0[u], pv0[u], accm0);
                    accm1 = fmaf(mv1[u], pv1[u], accm1);
                }
            }
        }
    }
    const float tot0 = hsum8_exact_dpp(acc0), tot1 = hsum8_exact_dpp(acc1);
    if constexpr (!mins) { out0 = tot0; out1 = tot1; return; }
    if constexpr (TYPE == GT_Q4_K) {
        const float w0 = accm0 + lane_xor4(accm0), w1 = accm1 + lane_xor4(accm1);
        accm0 = w0 + lane_xor2(w0);
        accm1 = w1 + lane_xor2(w1);
    }
    accm0 = __shfl(accm0, lane & ~7);
    accm1 = __shfl(accm1, lane & ~7);
    out0 = tot0 + accm0;
    out1 = tot1 + accm1;
}

struct GroupInfo {           // one type-homogeneous group of jobs of a launch
    int item0, n_items;      // item range of the group inside the launch's concatenated item list
};

// One group: every wave of the workgroup walks the group's units in rounds of T.
template <int TYPE, int MAXK, int S, int T, int NBUF, bool WITH_PROLOGUE, bool GU, bool LN>
DEV void run_group(const MatvecArgs& a, int item0, int n_items, ActLdsX<MAXK>& L,
                   ChainBuf5<MAXK / 256> (&CB)[NBUF][T], int lane, int wv, const LaneGeom& G, int pos, int& round_seq) {
    constexpr int NW = 16, MAXNB = MAXK / 256;
    const int stride = (int)gridDim.x, first = (int)blockIdx.x;
    const int upi = GU ? 2 : 1;
    // units of this workgroup inside the group: local items k = 0.. with item = item0 + first + k*stride
    const int n_loc = first < n_items ? (n_items - first + stride - 1) / stride : 0;
    const int n_units = n_loc * upi;
    if (n_units == 0) {
        if constexpr (WITH_PROLOGUE) {
            if constexpr (LN) prologue_q8k_exact16_ln<1024, MAXK>(L, a.x, a.norm_w, a.K, a.pro, a.eps, a.norm_b);
            else prologue_q8k_exact16<1024, MAXK>(L, a.x, a.norm_w, a.K, a.pro, a.eps);
        }
        return;
    }
    const uint32_t rec = (uint32_t)tile8_record_bytes(TYPE);

    auto unit_of = [&](int u, UnitInfo& U) __attribute__((always_inline)) {   // u is clamped to the last valid unit
        const int uu = u < n_units ? u : n_units - 1;
        const int k = GU ? (uu >> 1) : uu;
        const int part = GU ? (uu & 1) : 0;
        const int it = item0 + first + k * stride;
        int j = 0;
        if (!GU) {
            if (a.njobs > 1 && it >= a.job[1].pair0) j = 1;
            if (a.njobs > 2 && it >= a.job[2].pair0) j = 2;
        }
        const DevMat& w = GU ? a.job[part].w : a.job[j].w;
        U.valid = u < n_units;
        U.j = j;
        U.tile = it - (GU ? 0 : a.job[j].pair0);
        U.type = TYPE; U.nb = w.nb; U.M = w.M; U.rec = rec;
        U.base = w.p[0] + (size_t)U.tile * w.nb * rec;
    };
    auto load_img = [&](const UnitInfo& U, int i) __attribute__((always_inline)) -> BlkImg<TYPE> {
        int b = wv + i * NW;
        b = b < U.nb ? b : U.nb - 1;
        return img_load<TYPE>(U.base + (size_t)b * rec, G);
    };

    UnitInfo cur[T], nxt[T];
    BlkImg<TYPE> R[T][S];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        unit_of(t, cur[t]);
#pragma unroll
        for (int i = 0; i < S; ++i) R[t][i] = load_img(cur[t], i);
    }
    const bool trace = (a.dbg & 32) && blockIdx.x == 0 && lane == 0;
    unsigned long long* tr = (unsigned long long*)a.dbg_sink + 16 * wv + (WITH_PROLOGUE ? 0 : 6);
    if (trace) tr[1] = clock64_dev();
    // the first weight loads are in flight while the activation vector is normalised / quantized
    if constexpr (WITH_PROLOGUE) {
        if constexpr (LN) prologue_q8k_exact16_ln<1024, MAXK>(L, a.x, a.norm_w, a.K, a.pro, a.eps, a.norm_b);
        else prologue_q8k_exact16<1024, MAXK>(L, a.x, a.norm_w, a.K, a.pro, a.eps);
    }
    if (trace) tr[2] = clock64_dev();
    const int n_rounds = (n_units + T - 1) / T;
    for (int rd = 0; rd < n_rounds; ++rd, ++round_seq) {
        const int par = round_seq % NBUF;
        // the wave that will replay unit t's chain starts the dependent residual load now
        float res_in = 0.0f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (wv == ((round_seq * T + t) & (NW - 1)) && cur[t].valid && !GU && (a.job[cur[t].j].epi == EPI_ADD || a.job[cur[t].j].epi == EPI_ADD2)) {
                const int row = cur[t].tile * 8 + G.r;
                if (row < cur[t].M) res_in = a.res[row];
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            unit_of((rd + 1) * T + t, nxt[t]);
#pragma unroll
            for (int i = 0; i < S; ++i) {
                int b = wv + i * NW;
                b = b < cur[t].nb ? b : cur[t].nb - 1;
                if (cur[t].valid) img_to_chain<TYPE, MAXK, MAXNB>(R[t][i], b, L, CB[par][t], lane, G);   // tail rounds: skip
                if (nxt[t].valid) R[t][i] = load_img(nxt[t], i);                                        // the padding units
            }
        }
        if (trace && rd == 0) tr[3] = clock64_dev();
        __syncthreads();
        if (trace && rd == 0) tr[4] = clock64_dev();
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (wv != ((round_seq * T + t) & (NW - 1)) || !cur[t].valid) continue;
            if (GU && (t & 1)) continue;                       // the gate wave also replays the up chain
            const UnitInfo& U = cur[t];
            const int row = U.tile * 8 + G.r;
            const bool own = G.g == 0 && row < U.M;
            if (GU) {
                float res, up;
                chain_typed2<TYPE, MAXK, MAXNB>(U.nb, L, CB[par][t], CB[par][t + 1 < T ? t + 1 : t], lane, G, res, up);
                if (own) a.out[row] = f16_bits_to_f32(a.silu_tab[f32_to_f16_bits(res)]) * up;
            } else {
                const float res = chain_typed<TYPE, MAXK, MAXNB>(U.nb, L, CB[par][t], lane, G);
                const int epi = a.job[U.j].epi;
                if (epi == EPI_ADD) {
                    if (own) a.out[row] = res + res_in;
                } else if (epi == EPI_STORE) {
                    if (own) a.out[row] = res;
                } else if (epi == EPI_V) {
                    if (own) a.vcache[(size_t)row * a.v_stride + pos] = f32_to_f16_bits(res);
                } else if (epi == EPI_GELU) {
                    if (own) a.out[row] = f16_bits_to_f32(a.gelu_tab[f32_to_f16_bits(res)]);
                } else if (epi == EPI_ADD2) {
                    if (own) a.out[row] = (res + res_in) + a.res2[row];
                } else {
                    const float other = lane_xor8(res);
                    const int ip = (row % a.head_dim) >> 1;
                    const float cs = a.rope_cs[((size_t)pos * (a.head_dim >> 1) + ip) * 2 + 0];
                    const float sn = a.rope_cs[((size_t)pos * (a.head_dim >> 1) + ip) * 2 + 1];
                    const float o = (G.r & 1) ? fmaf(res, cs, other * sn) : fmaf(res, cs, -(other * sn));
                    if (own) {
                        if (epi == EPI_ROPE_Q) a.q_f16[row] = f32_to_f16_bits(o);
                        else a.kcache[kcache_off(pos, row, a.head_dim, a.n_ctx)] = f32_to_f16_bits(o);
                    }
                }
            }
        }
        if (trace && rd == 0) tr[5] = clock64_dev();
        if (NBUF == 1) __syncthreads();
#pragma unroll
        for (int t = 0; t < T; ++t) cur[t] = nxt[t];
    }
}

// TA: weight type of the launch (one type; mixed-type launches run on generation 6).  GU: gate/up pairs.  LN: LayerNorm
// prologue (falcon / gpt2) — a template parameter because merged into one body it cost every instantiation registers.
template <int MAXK, int S, int T, int NBUF, int TA, bool GU, bool LN>
__global__ void __launch_bounds__(1024) matvec_v5_kernel(const MatvecArgs a) {
    constexpr int MAXNB = MAXK / 256;
    __shared__ ActLdsX<MAXK> L;
    __shared__ ChainBuf5<MAXNB> CB[NBUF][T];
    const int lane = lane_id();
    const int wv = uniform_int(wave_id());
    const LaneGeom G = lane_geom(lane);
    const bool trace = (a.dbg & 32) && blockIdx.x == 0 && lane == 0;
    unsigned long long* tr = (unsigned long long*)a.dbg_sink + 16 * wv;
    if (trace) tr[0] = clock64_dev();
    const int pos = a.pos ? *a.pos : 0;
    int round_seq = 0;
    run_group<TA, MAXK, S, T, NBUF, true, GU, LN>(a, 0, a.n_groupA, L, CB, lane, wv, G, pos, round_seq);
    if (trace) tr[6] = clock64_dev();
}
