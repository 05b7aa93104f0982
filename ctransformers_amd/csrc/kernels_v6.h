// Mat-vec, generation 6: generation 5's arithmetic and data flow, with the per-round workgroup barrier replaced by
// two sets of LDS counters so that the serial part of a round — the f32 chain replay — leaves the critical path.
//
// Generation 5 timeline of one round (in-kernel s_memtime trace, gate+up launch): block math 4400 cycles, barrier,
// chain replay 5000-6700 cycles on two duty waves, and the NEXT barrier waits for those duty waves to redo their
// own block math: round time ~ math + chain ~ 11000 cycles although the VALU work of a round is ~4500 cycles.
//
// Here a round r uses chain-storage slot r % NBUF (NBUF >= 3) and
//   * every wave: waits until slot r is free (freed[slot] — the chains of round r-NBUF are replayed), does its block
//     math of round r, signals arrive[slot], and only THEN performs the chain duty it may have for round r-1;
//   * a duty wave waits for arrive[slot(r-1)] == all 16 waves before replaying, and signals freed[slot(r-1)] after.
// A duty wave therefore reaches round r+1 one chain-time late, but nobody waits for it before round r+1's replay,
// which is another wave's job one round later; the lateness is absorbed by the NBUF-deep slot ring and every wave
// pays one replay per 16/T rounds instead of the whole workgroup paying one per round.
#pragma once
#include "kernels_v5.h"

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

template <int MAXK, int T, int NBUF> struct SmemV6 {
    ActLdsX<MAXK> L;
    ChainBuf5<MAXK / 256> CB[NBUF][T];
    unsigned arrive[NBUF];   // waves that finished the block math of the round using the slot (monotonic)
    unsigned freed[NBUF];    // replay progress of the rounds that used the slot: every round adds 4 in total (monotonic)
};

struct UnitV6 {             // wave-uniform
    const uint8_t* base;    // first record of the tile
    int tile, j;            // tile index inside its matrix, job index (gate/up: 0 / 1)
    bool valid;
};

template <int TYPE, int S, int T, bool GU> struct GroupV6 {
    int item0, first, stride, n_units;
    UnitV6 cur[T];
    BlkImg<TYPE> R[T][S];
};

template <int TYPE> DEV constexpr uint32_t rec_bytes() { return TYPE == GT_Q4_K ? 1152u : (TYPE == GT_Q5_K ? 1408u : 1680u); }

template <int TYPE, int S, int T, bool GU>
DEV void v6_unit_of(const MatvecArgs& a, const GroupV6<TYPE, S, T, GU>& g, int u, UnitV6& U) {
    const int uu = u < g.n_units ? u : g.n_units - 1;
    const int k = GU ? (uu >> 1) : uu;
    const int it = g.item0 + g.first + k * g.stride;
    int j = 0;
    if (GU) {
        j = uu & 1;
    } else {
        if (a.njobs > 1 && it >= a.job[1].pair0) j = 1;
        if (a.njobs > 2 && it >= a.job[2].pair0) j = 2;
    }
    U.valid = u < g.n_units;
    U.j = j;
    U.tile = it - (GU ? 0 : a.job[j].pair0);
    U.base = a.job[j].w.p[0] + (size_t)U.tile * (uint32_t)(a.K >> 8) * rec_bytes<TYPE>();
}

template <int TYPE>
DEV BlkImg<TYPE> v6_load_img(const UnitV6& U, int nb, int i, int wv, const LaneGeom& G) {
    int b = wv + i * 16;
    b = b < nb ? b : nb - 1;
    return img_load<TYPE>(U.base + (size_t)b * rec_bytes<TYPE>(), G);
}

template <int TYPE, int S, int T, bool GU>
DEV void v6_begin(const MatvecArgs& a, int item0, int n_items, GroupV6<TYPE, S, T, GU>& g, int wv, const LaneGeom& G) {
    g.item0 = item0;
    g.stride = (int)gridDim.x;
    g.first = (int)blockIdx.x;
    const int n_loc = g.first < n_items ? (n_items - g.first + g.stride - 1) / g.stride : 0;
    g.n_units = n_loc * (GU ? 2 : 1);
    if (g.n_units == 0) return;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        v6_unit_of<TYPE, S, T, GU>(a, g, t, g.cur[t]);
        if (g.cur[t].valid) {
#pragma unroll
            for (int i = 0; i < S; ++i) g.R[t][i] = v6_load_img<TYPE>(g.cur[t], a.K >> 8, i, wv, G);
        }
    }
}

// Chain duty of round `rs` (slot rs % NBUF) for the units `U[]` that round had.
template <int TYPE, int MAXK, int T, int TCB, int NBUF, bool GU>
DEV void v6_duty(const MatvecArgs& a, SmemV6<MAXK, TCB, NBUF>& SM, const UnitV6 (&U)[T], int rs, float res_in, int lane, int wv,
                 const LaneGeom& G, int pos) {
    constexpr int MAXNB = MAXK / 256;
    const int slot = rs % NBUF;
    const unsigned all_waves = 16u * (unsigned)(rs / NBUF + 1);
    const int nb = a.K >> 8;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if (GU && (t & 1)) continue;                                  // the gate slot replays the up chain as well
        if (wv != ((rs * T + t) & 15)) continue;
        lds_wait_ge(&SM.arrive[slot], all_waves);
        if (U[t].valid) {
            const int row = U[t].tile * 8 + G.r;
            const bool own = G.g == 0 && row < a.job[U[t].j].w.M;
            if (GU) {
                float gate, up;
                chain_typed2<TYPE, MAXK, MAXNB>(nb, SM.L, SM.CB[slot][t], SM.CB[slot][t + 1 < T ? t + 1 : t], lane, G, gate, up);
                if (own) a.out[row] = f16_bits_to_f32(a.silu_tab[f32_to_f16_bits(gate)]) * up;
            } else {
                const float res = chain_typed<TYPE, MAXK, MAXNB>(nb, SM.L, SM.CB[slot][t], lane, G);
                const int epi = a.job[U[t].j].epi;
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
        lds_signal(&SM.freed[slot], lane, 4u / (GU ? T / 2 : T));   // T in {2, 4}: a round's duty slots add up to 4
    }
}

// All rounds of one group.  `hook` runs once, after this wave's block math of the last round: the kernel uses it to
// issue the next group's first loads.  The duty of the group's last round is flushed before returning.
template <int TYPE, int MAXK, int S, int T, int TCB, int NBUF, bool GU, bool GROUP_B, class Hook>
DEV void v6_rounds(const MatvecArgs& a, GroupV6<TYPE, S, T, GU>& g, SmemV6<MAXK, TCB, NBUF>& SM, int lane, int wv,
                   const LaneGeom& G, int pos, int& rs, Hook hook) {
    constexpr int MAXNB = MAXK / 256;
    static_assert(T == 2 || T == 4, "freed[] accounting assumes 1, 2 or 4 duty slots per round");
    const bool trace = (a.dbg & 32) && blockIdx.x == 0 && lane == 0;
    unsigned long long* tr = (unsigned long long*)a.dbg_sink + 16 * wv + (GROUP_B ? 6 : 0);
    const int n_rounds = (g.n_units + T - 1) / T;
    if (n_rounds == 0) { hook(); return; }
    const int nb = a.K >> 8;
    UnitV6 nxt[T], prev[T];
    float res_prev = 0.0f;
    bool pending = false;
    for (int rd = 0; rd < n_rounds; ++rd) {
        const int slot = rs % NBUF;
        lds_wait_ge(&SM.freed[slot], 4u * (unsigned)(rs / NBUF));     // chains of round rs - NBUF are replayed
        float res_in = 0.0f;                                            // residual operand of my duty for THIS round
        if (!GU) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (wv == ((rs * T + t) & 15) && g.cur[t].valid && (a.job[g.cur[t].j].epi == EPI_ADD || a.job[g.cur[t].j].epi == EPI_ADD2)) {
                    const int row = g.cur[t].tile * 8 + G.r;
                    if (row < a.job[g.cur[t].j].w.M) res_in = a.res[row];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            v6_unit_of<TYPE, S, T, GU>(a, g, (rd + 1) * T + t, nxt[t]);
#pragma unroll
            for (int i = 0; i < S; ++i) {
                int b = wv + i * 16;
                b = b < nb ? b : nb - 1;
                if (g.cur[t].valid) img_to_chain<TYPE, MAXK, MAXNB>(g.R[t][i], b, SM.L, SM.CB[slot][t], lane, G);
                if (nxt[t].valid) g.R[t][i] = v6_load_img<TYPE>(nxt[t], nb, i, wv, G);
            }
        }
        if (rd == n_rounds - 1) hook();
        if (trace && rd == 0) tr[3] = clock64_dev();
        lds_signal(&SM.arrive[slot], lane, 1u);
        if (trace && !GROUP_B && rd < 8) tr[8 + rd] = clock64_dev();
        if (pending) v6_duty<TYPE, MAXK, T, TCB, NBUF, GU>(a, SM, prev, rs - 1, res_prev, lane, wv, G, pos);
        if (trace && rd == 1) tr[5] = clock64_dev();
#pragma unroll
        for (int t = 0; t < T; ++t) { prev[t] = g.cur[t]; g.cur[t] = nxt[t]; }
        res_prev = res_in;
        pending = true;
        ++rs;
    }
    v6_duty<TYPE, MAXK, T, TCB, NBUF, GU>(a, SM, prev, rs - 1, res_prev, lane, wv, G, pos);
}

// TA / TB: weight types of the two job groups (TB == 0: one group).  Dynamic LDS: sizeof(SmemV6<MAXK, T, NBUF>).
template <int MAXK, int S, int T, int NBUF, int TA, int TB, bool GU, bool LN = false>
__global__ void __launch_bounds__(1024) matvec_v6_kernel(const MatvecArgs a) {
    static_assert(NBUF >= 2, "the lagged chain duty needs at least two slots (three to gain anything)");
    CT_DYN_SMEM(smem_raw);
    SmemV6<MAXK, T, NBUF>& SM = *reinterpret_cast<SmemV6<MAXK, T, NBUF>*>(smem_raw);
    const int lane = lane_id();
    const int wv = uniform_int(wave_id());
    const LaneGeom G = lane_geom(lane);
    const bool trace = (a.dbg & 32) && blockIdx.x == 0 && lane == 0;
    unsigned long long* tr = (unsigned long long*)a.dbg_sink + 16 * wv;
    if (trace) tr[0] = clock64_dev();
    if (threadIdx.x < NBUF) { SM.arrive[threadIdx.x] = 0u; SM.freed[threadIdx.x] = 0u; }   // published by the prologue's barrier
    const int pos = a.pos ? *a.pos : 0;
    int rs = 0;
    GroupV6<TA, S, T, GU> ga;
    v6_begin<TA, S, T, GU>(a, 0, a.n_groupA, ga, wv, G);
    if (trace) tr[1] = clock64_dev();
    // LN (falcon's LayerNorm prologue) is a template parameter: merged into one body it cost every instantiation registers
    if constexpr (LN) prologue_q8k_exact16_ln<1024, MAXK>(SM.L, a.x, a.norm_w, a.K, a.pro, a.eps, a.norm_b);
    else prologue_q8k_exact16<1024, MAXK>(SM.L, a.x, a.norm_w, a.K, a.pro, a.eps);
    if (trace) tr[2] = clock64_dev();
    if constexpr (TB != 0) {
        constexpr int T2 = T > 2 ? 2 : T;   // the Q6_K group of a mixed launch is the small one
        GroupV6<TB, S, T2, false> gb;
        v6_rounds<TA, MAXK, S, T, T, NBUF, GU, false>(a, ga, SM, lane, wv, G, pos, rs, [&]() __attribute__((always_inline)) {
            v6_begin<TB, S, T2, false>(a, a.n_groupA, a.n_pairs - a.n_groupA, gb, wv, G);
        });
        v6_rounds<TB, MAXK, S, T2, T, NBUF, false, true>(a, gb, SM, lane, wv, G, pos, rs, []() {});
    } else {
        v6_rounds<TA, MAXK, S, T, T, NBUF, GU, false>(a, ga, SM, lane, wv, G, pos, rs, []() {});
    }
    if (trace) tr[6] = clock64_dev();
}
