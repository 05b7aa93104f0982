// Mat-vec, generation 4 ("one workgroup per CU"): same bit-exact arithmetic as kernels_exact.h, reorganised after the
// round-1 PMC profile (profiles/r01_pmc_matvec_v3.txt) showed the 256-thread design to be instruction-issue bound:
//   * 1024-thread persistent workgroups, one per CU: the activation prologue runs once per CU (not once per 8-row
//     tile group) and costs 4 elements per thread; 16 waves per CU keep >=64 KB of weight loads in flight;
//   * K is split over the 16 waves (block b -> wave b mod 16); two 8-row tiles are processed per barrier round, their
//     f32 fma chains are replayed by two different waves (rotating) while the other 14 start the next round;
//   * the 6-bit scale/min field of Q4_K/Q5_K headers is re-encoded at load (LAYOUT_TILE8S, same 12 bytes) as four
//     24-bit groups {sc[2c], sc[2c+1], m[2c], m[2c+1]} so a lane extracts its four values with one alignbit + 4 bfe;
//   * d / dmin conversion and the y.d products moved out of the per-block path into the chain wave;
//   * all work bookkeeping is wave-uniform and per round (not per block step).
#pragma once
#include "kernels_exact.h"

template <int MAXNB> struct ChainBuf4 {
    float S[MAXNB][64];       // (float)sumi[l(g)] per (block, lane)
    uint32_t H[MAXNB][8];     // raw scale word per (block, row): Q4_K/Q5_K d | dmin<<16 (fp16 bits), Q6_K d
    float PM[MAXNB][32];      // (float)prod[t] per (block, row, t); Q5_K: [row*4+0] holds the summed product
};

struct UnitInfo {
    int valid, type, nb, M, tile, j;
    const uint8_t* base;      // first record of the tile
    uint32_t rec;
};

DEV UnitInfo unit_make(const MatvecArgs& a, int item, int part) {
    UnitInfo u;
    u.valid = item < a.n_pairs;
    const int it = u.valid ? item : 0;
    int j = 0;
    if (!a.gateup) {
        if (a.njobs > 1 && it >= a.job[1].pair0) j = 1;
        if (a.njobs > 2 && it >= a.job[2].pair0) j = 2;
    }
    const DevMat& w = a.gateup ? a.job[part].w : a.job[j].w;
    u.j = j;
    u.tile = it - (a.gateup ? 0 : a.job[j].pair0);
    u.type = w.type; u.nb = w.nb; u.M = w.M;
    u.rec = (uint32_t)tile8_record_bytes(w.type);
    u.base = w.p[0] + (size_t)u.tile * w.nb * u.rec;
    return u;
}

// Integer work of one block; scale field in the LAYOUT_TILE8S encoding.
template <int MAXK, int MAXNB>
DEV void block_to_chain4(int type, int b, const ActLdsX<MAXK>& L, ChainBuf4<MAXNB>& C, int lane, const LaneGeom& G,
                         const BlockRegs& R) {
    const int c = G.c;
    if (type == GT_Q4_K || type == GT_Q5_K) {
        const bool q5 = type == GT_Q5_K;
        const int* alo = &L.q8[b * 64 + G.a45];
        const int* ahi = alo + 8;
        // 24-bit group c of the re-encoded scale field (bits 24c.. of header bytes 4..15)
        const uint32_t lo_w = c < 2 ? R.v0[1] : (c == 2 ? R.v0[2] : R.v0[3]);
        const uint32_t hi_w = c < 2 ? R.v0[2] : R.v0[3];
        const uint32_t x = alignbit32(hi_w, lo_w, (uint32_t)((24 * c) & 31));
        const int sc_lo = (int)(x & 63u), sc_hi = (int)bfe32(x, 6, 6), m_lo = (int)bfe32(x, 12, 6), m_hi = (int)bfe32(x, 18, 6);
        int part[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t lo = R.v1[k] & 0x0F0F0F0Fu;
            uint32_t hi = (R.v1[k] >> 4) & 0x0F0F0F0Fu;
            if (q5) {
                lo |= ((R.v2[k] >> (2 * c)) & 0x01010101u) << 4;
                hi |= ((R.v2[k] >> (2 * c + 1)) & 0x01010101u) << 4;
            }
            part[k] = mul24(sc_lo, sdot4((int)lo, alo[k], 0)) + mul24(sc_hi, sdot4((int)hi, ahi[k], 0));
        }
        const int sumi = quad_transpose_reduce_dpp(part[0], part[1], part[2], part[3], c);
        C.S[b][lane] = (float)sumi;
        int prod = mul24(m_lo, L.sb[b * 8 + 2 * c]) + mul24(m_hi, L.sb[b * 8 + 2 * c + 1]);
        if (q5) {
            if (G.h != 0) prod = 0;
            prod += lane_xor2(prod);
            prod += lane_xor4(prod);
        }
        if (G.h == 0) C.PM[b][G.r * 4 + c] = (float)prod;
        if (G.g == 0) C.H[b][G.r] = R.v0[0];
    } else {
        const int n = G.g >> 2;
        const int* alo = &L.q8[b * 64 + G.a6];
        const int* ahi = alo + 16;
        const uint32_t w_lo = n ? R.v0[2] : R.v0[0];
        const uint32_t w_hi = n ? R.v0[3] : R.v0[1];
        const int sc_lo = (int)(int8_t)((w_lo >> G.sc_sh6) & 0xFF);
        const int sc_hi = (int)(int8_t)((w_hi >> G.sc_sh6) & 0xFF);
        int part[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = (R.v1[k] & 0x0F0F0F0Fu) | (((R.v2[k] >> G.s_lo6) & 0x03030303u) << 4);
            const uint32_t hi = ((R.v1[k] >> 4) & 0x0F0F0F0Fu) | (((R.v2[k] >> G.s_hi6) & 0x03030303u) << 4);
            const int dl = sdot4((int)lo, alo[k], sdot4((int)0xE0E0E0E0u, alo[k], 0));
            const int dh = sdot4((int)hi, ahi[k], sdot4((int)0xE0E0E0E0u, ahi[k], 0));
            part[k] = mul24(sc_lo, dl) + mul24(sc_hi, dh);
        }
        const int sumi = quad_transpose_reduce_dpp(part[0], part[1], part[2], part[3], c);
        C.S[b][lane] = (float)sumi;
        if (G.g == 0) C.H[b][G.r] = R.dd;
    }
}

// One wave: d = y.d * fp16(x.d) etc. per block, then the reference's sequential fma chain and reduction tree.
template <int MAXK, int MAXNB>
DEV float chain_reduce4(int type, int nb, const ActLdsX<MAXK>& L, const ChainBuf4<MAXNB>& C, int lane, const LaneGeom& G) {
    float acc = 0.0f, accm = 0.0f;
    const bool mins = type != GT_Q6_K;
    for (int b0 = 0; b0 < nb; b0 += 8) {
        float dv[8], sv[8], mv[8], pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = (b0 + u < nb) ? b0 + u : nb - 1;
            const uint32_t hw = C.H[b][G.r];
            const float yd = L.yd[b];
            dv[u] = yd * f16_bits_to_f32((uint16_t)(hw & 0xFFFF));
            mv[u] = mins ? -yd * f16_bits_to_f32((uint16_t)(hw >> 16)) : 0.0f;
            sv[u] = C.S[b][lane];
            pv[u] = (mins && G.h == 0) ? C.PM[b][G.r * 4 + G.c] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (b0 + u < nb) {
                acc = fmaf(dv[u], sv[u], acc);
                accm = fmaf(mv[u], pv[u], accm);
            }
        }
    }
    const float tot = hsum8_exact_dpp(acc);
    if (!mins) return tot;
    if (type == GT_Q4_K) {
        const float wsum = accm + lane_xor4(accm);
        accm = wsum + lane_xor2(wsum);
    }
    accm = __shfl(accm, lane & ~7);
    return tot + accm;
}

DEV void epilogue4(const MatvecArgs& a, const UnitInfo& u, float res, int lane, const LaneGeom& G, int pos) {
    const int row = u.tile * 8 + G.r;
    const bool own = G.g == 0 && row < u.M;
    const int epi = a.job[u.j].epi;
    if (epi == EPI_STORE) {
        if (own) a.out[row] = res;
    } else if (epi == EPI_ADD) {
        if (own) a.out[row] = res + a.res[row];
    } else if (epi == EPI_V) {
        if (own) a.vcache[(size_t)row * a.v_stride + pos] = f32_to_f16_bits(res);
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

// S = block steps per wave per tile (ceil(nb/16)); NBUF = chain-buffer parity depth (2 = double buffered).
template <int MAXK, int S, int NBUF>
__global__ void __launch_bounds__(1024) matvec_v4_kernel(const MatvecArgs a) {
    constexpr int NT = 1024, NW = 16;
    constexpr int MAXNB = MAXK / 256;
    __shared__ ActLdsX<MAXK> L;
    __shared__ ChainBuf4<MAXNB> CB[NBUF][2];
    if (a.dbg & 16) return;
    const int lane = lane_id();
    const int wv = uniform_int(wave_id());
    const LaneGeom G = lane_geom(lane);
    const int stride = (int)gridDim.x;
    const int first = (int)blockIdx.x;

    // round p of this workgroup handles units A,B: gate/up mode -> (gate tile, up tile) of item first + p*stride;
    // otherwise the items first + 2p*stride and first + (2p+1)*stride.
    auto pair_units = [&](int p, UnitInfo& A, UnitInfo& B) __attribute__((always_inline)) {
        if (a.gateup) {
            const int it = first + p * stride;
            A = unit_make(a, it, 0);
            B = unit_make(a, it, 1);
        } else {
            A = unit_make(a, first + (2 * p) * stride, 0);
            B = unit_make(a, first + (2 * p + 1) * stride, 0);
        }
    };
    auto load_blk = [&](const UnitInfo& u, int i) __attribute__((always_inline)) -> BlockRegs {
        if (a.dbg & 8) { BlockRegs Z; Z.v0 = Z.v1 = Z.v2 = u32x4{1, 2, 3, 4}; Z.dd = 0; return Z; }
        int b = wv + i * NW;
        b = b < u.nb ? b : u.nb - 1;
        return block_load2(u.type, u.base + (size_t)b * u.rec, G);
    };

    UnitInfo cA, cB, nA, nB;
    pair_units(0, cA, cB);
    BlockRegs RA[S], RB[S];
#pragma unroll
    for (int i = 0; i < S; ++i) {
        RA[i] = load_blk(cA, i);   // cA.valid is guaranteed by the launch (grid <= items)
        RB[i] = load_blk(cB.valid ? cB : cA, i);
    }
    if (!(a.dbg & 1)) prologue_q8k_exact16<NT, MAXK>(L, a.x, a.norm_w, a.K, a.pro, a.eps);
    const int pos = a.pos ? *a.pos : 0;

    for (int p = 0; cA.valid; ++p) {
        pair_units(p + 1, nA, nB);
        ChainBuf4<MAXNB>& CA = CB[p % NBUF][0];
        ChainBuf4<MAXNB>& CBb = CB[p % NBUF][1];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const int b = wv + i * NW;
            if (b < cA.nb && !(a.dbg & 2)) block_to_chain4<MAXK, MAXNB>(cA.type, b, L, CA, lane, G, RA[i]);
            if (nA.valid) RA[i] = load_blk(nA, i);
            if (cB.valid && b < cB.nb && !(a.dbg & 2)) block_to_chain4<MAXK, MAXNB>(cB.type, b, L, CBb, lane, G, RB[i]);
            if (nB.valid) RB[i] = load_blk(nB, i);
        }
        __syncthreads();
        const int wA = (2 * p) & (NW - 1), wB = (2 * p + 1) & (NW - 1);
        if (a.dbg & 4) {
            if ((a.dbg & 2) && lane == 0 && wv == 0) a.dbg_sink[0] = (float)(RA[0].v1[0] + RB[0].v1[1]);  // keep the loads alive
        } else if (a.gateup) {
            if (wv == wA) {
                const float gate = chain_reduce4<MAXK, MAXNB>(cA.type, cA.nb, L, CA, lane, G);
                const float up = chain_reduce4<MAXK, MAXNB>(cB.type, cB.nb, L, CBb, lane, G);
                const int row = cA.tile * 8 + G.r;
                if (G.g == 0 && row < cA.M) a.out[row] = f16_bits_to_f32(a.silu_tab[f32_to_f16_bits(gate)]) * up;
            }
        } else {
            if (wv == wA) epilogue4(a, cA, chain_reduce4<MAXK, MAXNB>(cA.type, cA.nb, L, CA, lane, G), lane, G, pos);
            if (cB.valid && wv == wB) epilogue4(a, cB, chain_reduce4<MAXK, MAXNB>(cB.type, cB.nb, L, CBb, lane, G), lane, G, pos);
        }
        if (NBUF == 1 && nA.valid) __syncthreads();  // single-buffered chain storage: readers done before the next round writes
        cA = nA;
        cB = nB;
    }
}
