// Decode attention, generation 9: one launch per layer on EVERY CU, written for its latency chain.
//
// attn_fused_exact_kernel (kernels_exact.h) runs a token's attention as n_head x head_dim/64 workgroups — 64 of 256 CUs for a 7B —
// and as a serial chain: position (vector load) -> K rows -> scores -> max -> exp table -> sum -> V*P -> leftovers, each link one
// memory latency: 9200 cycles at 140 positions (profiles/r03_v9_inkernel_trace_7b_q4km.txt), bandwidth-bound on a quarter of the
// chip at 2000 (31 us).  Here:
//   * grid = n_head x ng workgroups, ng = head_dim / (channels per workgroup) chosen by the host so that the grid is about one
//     workgroup per CU (7B: 32 heads x 8 groups of 16 channels; 70B: 64 heads x 4 groups of 32): a workgroup owns a group of output
//     channels of one head and recomputes the head's score row — cheap at short contexts, and at long ones the redundancy (ng) is
//     what the host bounds.  The groups of a head sit on one XCD (blockIdx % 8 is the XCD: for speed only), so its K rows come out of
//     that XCD's L2 after the first fetch and HBM sees K and V once.
//   * seven score waves (a quad of lanes per position, 112 positions per pass and slot) + one V*P wave per 16 channels — the V*P
//     waves are the last of the workgroup, which start up to a microsecond after the first (wave launch rate): they are not on the
//     critical path before the probabilities exist.  Nothing waits for
//     the cursor: the query, the K rows of the first 256 positions (clamped to the cache) and the first V chunks are requested at
//     kernel entry, the cursor {step, pos, n_total, batch} arrives as ONE scalar load beside them; the V*P waves' rows landed long
//     before the probabilities exist, the leftover positions' V values are requested as soon as the cursor is known.
// The arithmetic per position / channel and its order are attn_fused_exact_kernel's (reference ggml_vec_dot_f16, ggml.c:2392-2425:
// a quad of lanes owns the four AVX accumulator vectors of a dot product; softmax through the fp16 exp table with an order-free
// double sum, ggml.c:12009-12078; V*P in steps of 32 positions in order, leftovers sequentially in double).
#pragma once
#include "kernels_exact.h"

// ---- tagged granules: the in-launch hand-off between workgroups (MI355X_MICROARCH.md "handoff-1to1" / "allgather"; used by the shared score
// row below and by the fused QKV + attention launch, kernels_qa9.h) ----
// one granule: {payload, tag} in one 8-byte device-scope store / load
#ifdef CT_EMU
static inline void st_granule(uint32_t* p, uint32_t data, uint32_t tag) { p[0] = data; p[1] = tag; }
static inline void ld_granule(const uint32_t* p, uint32_t& data, uint32_t& tag) { data = p[0]; tag = p[1]; }
static inline unsigned long long wall_ticks() { return 0ull; }
#else
DEV void st_granule(uint32_t* p, uint32_t data, uint32_t tag) {
    const unsigned long long v = ((unsigned long long)tag << 32) | data;
    __hip_atomic_store((unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_store_dwordx2 ... sc1
}
// (A plain store instead — the line then stays in the writer's XCD L2, where the other workgroups of a KV head sit when the head counts allow it —
// measured +0.2 % decode, +0.6 % at 2k positions, and is not safe: a reader on another XCD never sees the line; it timed out in
// tests/test_gpu_parity.py::test_context_above_8192.  Same-XCD placement stays a speed bonus, never something correctness rests on.)
// (Polling with a memory-side atomic instead — fetch_or with 0 — was measured: no different in outcome, and the fused launch lost its
// whole gain over two launches: 734 against 732 tok/s.)
DEV void ld_granule(const uint32_t* p, uint32_t& data, uint32_t& tag) {
    const unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    data = (uint32_t)v;
    tag = (uint32_t)(v >> 32);
}
DEV unsigned long long wall_ticks() { return wall_clock64(); }   // 100 MHz
#endif

// PB / VB: K-row slots per quad / V chunk slots per V*P lane — the depth of the request rings.  The host picks (2, 4) for contexts up
// to 1024 (the requests of a deeper ring only delay the short chain) and (4, 16) above.
template <bool B> struct A9Req { static constexpr bool value = B; };

// NWV score waves.  Short contexts: seven, beside up to four V*P waves (<= 768 threads, three waves on a SIMD: 170 registers).  The deep
// rings of the long-context form need more registers than that: it runs four score waves (one per SIMD) beside the V*P waves, at most
// 512 threads, two waves per SIMD.
// SHARE (long contexts): the ng channel-group workgroups of a head share ONE score row instead of each recomputing it (at 2001 positions the
// score phase was 14 600 of 33 000 cycles, ng = 4 .. 8 times over): workgroup `grp` computes the scores of positions [grp * per, (grp + 1) * per),
// per = ceil(n_kv / ng), and publishes them as tagged granules {score bits, tag} in a.xs[head][position] (one 8-byte device-scope store each);
// then every workgroup gathers the head's whole row into LDS, polling until every tag is this launch's (tag = (token epoch, layer): a
// granule of an earlier token step never matches).  The scores are per position and unchanged, so the row — and everything after it —
// is bit for bit the recomputed one.  The workgroups of a head wait for each other: the grid (about one workgroup per CU, at most 512
// threads) must be resident as a whole; a gather that sees nothing new for 20 ms raises a.err (the host reports the eval as failed and
// goes back to the recomputing form).
template <int HD, int PB, int VB, int NWV = 7, int MAXT = 768, bool SHARE = false>
// The four leading arguments repeat what the first dependent requests need (the cursor's address, the workgroup map's inputs): the build preloads
// them into scalar registers (Makefile: PRELOAD), so the cursor load leaves at wave start instead of behind the kernel-argument fetch.
__global__ void __launch_bounds__(MAXT) attn_decode9_kernel(const int* cur0, int ng, int n_head, int n_head_kv, const AttnArgsX a) {
    kernarg_touch<24 + sizeof(AttnArgsX)>();
    constexpr int NT = 64 * NWV, NQ = NT / 4;   // score threads; NQ quads: positions per pass and slot
    constexpr int NC = HD / 32;                     // 16-byte chunks of a K row per quad lane
    constexpr int PSPEC = 2, VSPEC = 4;             // slots requested before the cursor is known (a slot is re-requested for the next pass right after its use)
    CT_DYN_SMEM(smem_raw);   // the score / probability row of this token: n_ctx floats
    float* prob = reinterpret_cast<float*>(smem_raw);
    __shared__ double red[NWV];
    __shared__ float redf[SHARE ? 16 : NWV];
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = uniform_int(wave_id()), j = tid & 3, quad = tid >> 2;
    // workgroup -> (head, channel group): the workgroups that read the same K / V rows (the query heads of one KV head, all their
    // channel groups) take consecutive positions on ONE XCD (blockIdx % 8) where the head counts allow it — for speed only
    int h, grp;
    {
        const int b = (int)blockIdx.x, rep = n_head / n_head_kv;
        if ((n_head_kv & 7) == 0) {
            const int i = b >> 3, per = rep * ng;
            const int hkv = (b & 7) + 8 * (i / per), r = (i / ng) % rep;
            h = hkv * rep + r; grp = i % ng;
        } else { h = b / ng; grp = b - h * ng; }
    }
    const bool trace = a.trace && blockIdx.x == 0 && lane == 0;
    unsigned long long* tr = a.trace + 16 * (wv < 16 ? wv : 15);
    if (trace) tr[0] = clock64_dev();
    const int hk = h / (n_head / n_head_kv);
    const bool pv_wave = wv >= NWV;                 // V*P: 16 channels x 4 lanes per wave, HD / ng / 16 such waves
    // ---- requests that do not depend on the cursor ----
    const uint16_t* kbase = a.kcache + (size_t)hk * a.n_ctx * HD + 8 * j;
    // One register set for both kinds of waves (they are the same kernel: separate arrays would be allocated side by side and spill):
    // score waves: buf = K-row slots [PB][NC], aux = this lane's slices of the query (fp16 pairs, read through v_fma_mix_f32);
    // V*P waves: buf = V chunk slots [VB], aux = the leftover positions' values
    constexpr int NBUF = PB * NC > VB ? PB * NC : VB;
    static_assert(NC <= 4, "register set");
    u32x4 buf[NBUF], aux[4];
    const int d = grp * (HD / ng) + (pv_wave ? wv - NWV : 0) * 16 + (lane >> 2);
    const uint16_t* vrow = a.vcache + ((size_t)hk * HD + d) * a.v_stride;
    if (pv_wave) {
#pragma unroll
        for (int u = 0; u < VSPEC; ++u) {
            const int off = 32 * u + 32 <= a.v_stride ? 32 * u : a.v_stride - 32;
            buf[u] = ld16(vrow + off + 8 * j);
        }
    } else {
        const uint16_t* qrow = a.q_f16 + (size_t)h * HD;
#pragma unroll
        for (int c = 0; c < NC; ++c) aux[c] = ld16(qrow + 32 * c + 8 * j);
#pragma unroll
        for (int u = 0; u < (SHARE ? 0 : PSPEC); ++u) {   // (SHARE: the workgroup's slice of the row starts where the cursor says)
            int p = u * NQ + quad;
            p = p < a.n_ctx ? p : a.n_ctx - 1;
            const uint16_t* krow = kbase + (size_t)p * HD;
#pragma unroll
            for (int c = 0; c < NC; ++c) buf[u * NC + c] = ld16(krow + 32 * c);
        }
    }
    // ---- the cursor: {step, pos, n_past + n, batch} (kernels.h), one scalar load ----
    int cur[4];
    uint32_t tag = 0u;
    if constexpr (SHARE) tag = (((uint32_t)sload_i32x4_and(cur0, cur, (const int*)a.epoch) + 1u) << 8) | (uint32_t)a.layer;   // (one scalar round trip for both)
    else sload_i32x4(cur0, cur);
    const int n_kv = cur[1] + 1;
    // The length of the value dot product is that of the reference batch this token belongs to (attn_fused_exact_kernel).
    int n_tot = cur[2];
    if (cur[3] > 0) {
        const int idx = cur[0], base = cur[1] - cur[0];
        const int end = (idx / cur[3] + 1) * cur[3], n_eval = n_tot - base;
        n_tot = base + (end < n_eval ? end : n_eval);
    }
    const int np = n_tot & ~31;
    const int nl = n_kv - np;   // leftover positions (< 32), wave-uniform
    // Every request below is UNCONDITIONAL (an address clamped into the cache instead of a branch around the load).  With a branch
    // around a load anywhere in a loop hipcc cannot count the loads in flight and waits `vmcnt(0)` before every use of a slot — the
    // rings then hold one request at a time, each step costs a whole memory latency (the kernel had 43 such waits: 370 cycles per
    // 32-position V step, 23 000 of 48 900 cycles at 2000 positions).  The loops are peeled instead: all rounds but the last re-request
    // every slot, the last requests nothing (kernels_v9.h has the same rule for the weight ring).
    const int last_c = np >= 32 ? np - 32 : 0;   // first position of the last 32-position V chunk of the fma part
    const int last_p = n_kv - 1;
    // SHARE: this workgroup's slice [p0, p1) of the score row
    const int per = SHARE ? (n_kv + ng - 1) / ng : n_kv;
    const int p0 = SHARE ? grp * per : 0, p1 = SHARE ? (p0 + per < n_kv ? p0 + per : n_kv) : n_kv;
    if (pv_wave) {   // the rest of the first V chunks, the leftover positions' values (the row is padded: np + 31 stays inside the cache)
#pragma unroll
        for (int u = VSPEC; u < VB; ++u) buf[u] = ld16(vrow + (32 * u < last_c ? 32 * u : last_c) + 8 * j);
#pragma unroll
        for (int c = 0; c < 4; ++c) aux[c] = ld16(vrow + np + 8 * c);
    } else {         // the K rows of the first pass beyond the speculative slots
#pragma unroll
        for (int u = (SHARE ? 0 : PSPEC); u < PB; ++u) {
            const int p = p0 + u * NQ + quad;
            const uint16_t* krow = kbase + (size_t)(p < last_p ? p : last_p) * HD;
#pragma unroll
            for (int c = 0; c < NC; ++c) buf[u * NC + c] = ld16(krow + 32 * c);
        }
    }
    if (trace) { tr[1] = clock64_dev(); tr[7] = (unsigned long long)n_kv; }
    // ---- scores: a pass = PB x 128 positions; a slot's K row of the NEXT pass is requested right after its dot product ----
    float mx = -INFINITY;
    if (!pv_wave) {
        auto slot = [&](int u, int base, auto REQ) __attribute__((always_inline)) {
            const int p = base + u * NQ + quad;
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < NC; ++c) fma8_hh(acc, buf[u * NC + c], aux[c]);   // acc[l] = fma(k[l], q[l], acc[l]): both halves converted by the fma
            if constexpr (decltype(REQ)::value) {   // the slot's K row of the NEXT pass
                const int pn = p + NQ * PB;
                const uint16_t* krow = kbase + (size_t)(pn < last_p ? pn : last_p) * HD;
#pragma unroll
                for (int c = 0; c < NC; ++c) buf[u * NC + c] = ld16(krow + 32 * c);
            }
            const float sc = f16dot_reduce_exact(acc, j) * a.kq_scale;
            if constexpr (SHARE) {
                if (p < p1 && j == 0) st_granule(a.xs + ((size_t)h * a.n_ctx + p) * 2, f32_to_bits(sc), tag);
            } else if (p < n_kv) {
                mx = fmaxf(mx, sc);
                if (j == 0) prob[p] = sc;
            }
        };
        int base = p0;
        for (; base + NQ * PB < p1; base += NQ * PB) {
#pragma unroll
            for (int u = 0; u < PB; ++u) slot(u, base, A9Req<true>{});
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) slot(u, base, A9Req<false>{});
    }
    if (trace) tr[2] = clock64_dev();
    if constexpr (SHARE) {
        // gather the head's row: every thread of the workgroup (score and V*P waves alike) takes positions tid, tid + T, ..; eight granules in
        // flight per thread and round, a round repeats until the wave has seen this launch's tag on all of its granules.  (Gathering by the
        // V*P waves alone, from the start of the launch, so that the row is in LDS when the last slice lands, measured WORSE: one wave needs
        // four dependent rounds of device-scope loads — 24 000 cycles at 2001 positions against 6 800 for this form.)
        const int T = (int)blockDim.x;
        const uint32_t* row = a.xs + (size_t)h * a.n_ctx * 2;
        const unsigned long long t0 = wall_ticks();
        (void)t0;
        bool gave_up = false;
        for (int i0 = 0; i0 < n_kv && !gave_up; i0 += 8 * T) {
            for (;;) {
                uint32_t dat[8], tg[8];
                bool ok = true;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * T + tid;
                    dat[u] = 0u; tg[u] = tag;
                    if (i < n_kv) ld_granule(row + 2 * i, dat[u], tg[u]);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) ok = ok && tg[u] == tag;
                if (__ballot(!ok) == 0ull) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + u * T + tid;
                        if (i < n_kv) { const float sc = bits_to_f32(dat[u]); prob[i] = sc; mx = fmaxf(mx, sc); }
                    }
                    break;
                }
#ifdef CT_EMU
                emu::spin_yield();
#else
                __builtin_amdgcn_s_sleep(1);
                if (wall_ticks() - t0 > 2000000ull) { if (lane == 0) *a.err = 1; gave_up = true; break; }   // 20 ms: a workgroup of the head is not running
#endif
            }
        }
        mx = fmaxf(mx, lane_xor1(mx)); mx = fmaxf(mx, lane_xor2(mx));
    }
    mx = fmaxf(mx, lane_xor4(mx)); mx = fmaxf(mx, lane_xor8(mx)); mx = fmaxf(mx, lane_xor16(mx)); mx = fmaxf(mx, lane_xor32(mx));
    if (lane == 0 && (SHARE || !pv_wave)) redf[wv] = mx;
    __syncthreads();
    mx = redf[0];
    if constexpr (SHARE) {
        const int nwaves = (int)(blockDim.x >> 6);
        for (int w = 1; w < nwaves; ++w) mx = fmaxf(mx, redf[w]);
    } else {
#pragma unroll
        for (int w = 1; w < NWV; ++w) mx = fmaxf(mx, redf[w]);
    }
    if (trace) tr[3] = clock64_dev();
    // ---- softmax: fp16 exp table, order-free double sum (the addends are multiples of 2^-24 in (0, 1]) ----
    double sum = 0.0;
    constexpr int SB = VB >= 16 ? 8 : 4;   // exp-table lookups per thread in flight together (long-context form: one round for 2000+ positions)
    for (int i0 = 0; !pv_wave && i0 < n_kv; i0 += NT * SB) {
        uint16_t e16[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) { const int i = i0 + u * NT + tid; e16[u] = i < n_kv ? a.exp_tab[f32_to_f16_bits(prob[i] - mx)] : (uint16_t)0; }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int i = i0 + u * NT + tid;
            if (i < n_kv) { const float e = f16_bits_to_f32(e16[u]); prob[i] = e; sum += (double)e; }
        }
    }
    if (!pv_wave) {
        sum = wave_sum_fast(sum);
        if (lane == 0) red[wv] = sum;
    }
    __syncthreads();
    double tot = red[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) tot += red[w];
    const float inv = (float)(1.0 / tot);
    if (!pv_wave) {
        for (int i = tid; i < n_kv; i += NT) prob[i] = f16_bits_to_f32(f32_to_f16_bits(prob[i] * inv));
        for (int i = n_kv + tid; i < np; i += NT) prob[i] = 0.0f;  // masked columns of this batch
    }
    __syncthreads();
    if (trace) tr[4] = clock64_dev();
    if (!pv_wave) return;
    // ---- V*P of this wave's 16 channels: a quad per channel; a chunk slot is re-requested right after its fmas ----
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i0 = 0;
    float pn[8];   // the probabilities of the NEXT step, read one step ahead (an LDS round trip per step otherwise: 150 cycles per step at 2000 positions)
    {
        const float* p0 = &prob[8 * j];   // position 0 .. 7 + 8j: inside the row whatever np is
#pragma unroll
        for (int l = 0; l < 8; ++l) pn[l] = p0[l];
    }
    for (; i0 + 32 * VB < np; i0 += 32 * VB) {   // every chunk of this round exists; each slot is re-requested (clamped to the last chunk)
#pragma unroll
        for (int u = 0; u < VB; ++u) {
            const int i = i0 + 32 * u;
            float pc[8];
#pragma unroll
            for (int l = 0; l < 8; ++l) pc[l] = pn[l];
            {
                const float* pr = &prob[i + 32 + 8 * j];   // i + 32 <= i0 + 32 * VB < np
#pragma unroll
                for (int l = 0; l < 8; ++l) pn[l] = pr[l];
            }
            fma8_hf(acc, buf[u], pc);   // acc[l] = fma(v[l], p[l], acc[l])
            const int in = i + 32 * VB;
            buf[u] = ld16(vrow + (in < last_c ? in : last_c) + 8 * j);
        }
    }
#pragma unroll
    for (int u = 0; u < VB; ++u) {   // the last round requests nothing
        const int i = i0 + 32 * u;
        if (i < np) {
            float pc[8];
#pragma unroll
            for (int l = 0; l < 8; ++l) pc[l] = pn[l];
            {
                const int inx = i + 32 < last_c ? i + 32 : last_c;   // the step after the last re-reads the last chunk's probabilities (unused)
                const float* pr = &prob[inx + 8 * j];
#pragma unroll
                for (int l = 0; l < 8; ++l) pn[l] = pr[l];
            }
            fma8_hf(acc, buf[u], pc);
        }
    }
    const float res = f16dot_reduce_exact(acc, j);
    double sumf = (double)res;
    if (trace) tr[5] = clock64_dev();
    // leftover positions np .. n_kv - 1: ggml_vec_dot_f16's scalar tail, sumf += (double)(x[i] * y[i]) in order
    if (nl > 0) sumf = f16_tail32(sumf, aux, prob + np, nl);
    if (j == 0) a.out[(size_t)h * HD + d] = (float)sumf;
    if (trace) tr[6] = clock64_dev();
}

// ---- order-free decode attention (CT_AMD_DECODE_ATTN=fast, opt-in; long-context form only) ----
// The bit-identical kernel above keeps the reference's V*P order: per (head, channel) ONE chain of fma steps over the positions in file order,
// so one wave walks a channel group's whole V slice (62 dependent ring steps at 2001 positions, three memory round trips behind the softmax).
// Here every rounding POINT of the reference stays — the K.Q dot per position in its lane order (scores bit-identical), the fp16 input of the
// exp table, the exact double sum, the fp16 probabilities, f32 fma steps of 32 positions, the scalar double tail — and only the ORDER in which
// the 32-position steps of a channel meet is given up: the workgroup's eight waves are ALL score waves and ALL V*P waves; wave (cg, s) owns
// the chunk pairs s, s + SPL, s + 2 SPL, .. of channel group cg (SPL = 8 / channel groups) and requests its first sixteen chunks as soon as its scores
// are computed — at 2001 positions that is the workgroup's whole V slice, landed before the probabilities exist; the SPL partial sums of a channel meet in
// slice order, in double, and the scalar tail (its own chain from 0.0, run by the last slice's wave beside the others' fma steps) is added last.
// Same grid, same workgroup map, same shared score row (SHARE) and the same residency contract as attn_decode9_kernel<.., SHARE>.
// Measured at 2001 positions (7B / 70B / Falcon-40B widths, us per launch, same box; bit-identical form 15.0 / 14.0 / 14.2): V requested at kernel entry,
// single chunks per wave 14.8 / 13.7 / -; chunk PAIRS (2 m, 2 m + 1: the two halves of a 128-byte line of a V row) 14.3 / 13.2 / 13.6; pairs with the
// non-temporal hint on the K / V requests 16.6 / 17.5 (the two half-line requests of a line no longer meet in L1); pairs requested behind the row gather
// 14.0 / 12.9 / 13.2; pairs requested behind the wave's scores (what is built) 13.5 / 12.1 / 12.7; the same with four K-row slots per quad 13.6 / 12.3 / 12.3; with FOUR score
// waves of four slots (the other four waves wait at a barrier before their V requests) 13.3 / 11.7 / 12.8 — and 0.5 us slower at 1025 positions.
// At 1025 positions the built form and the bit-identical one are level (11.0 / 10.5 / 10.6 against 10.7 / 11.2 / 10.8).  Where the rest of the time is
// (in-kernel stamps, profiles/r06_decode_attn_free.txt): the K / V bytes arrive at ~4 TB/s (7B: 128 KB per CU; 70B widths: 256 KB per CU, seven eighths
// of it out of L2 — every query head of a group fetches the group's rows again), then the row gather (2.5 us), the softmax with its table look-ups (2 us),
// V*P (1 us) and the meeting of the slices (1.2 us).
template <int HD, bool SHARE>
__global__ void __launch_bounds__(512) attn_decode9_free_kernel(const int* cur0, int ng, int n_head, int n_head_kv, const AttnArgsX a) {
    kernarg_touch<24 + sizeof(AttnArgsX)>();
    constexpr int NWV = 8, NT = 64 * NWV, NQ = NT / 4;   // every wave computes scores: 128 positions per slot
    constexpr int NC = HD / 32, PB = 2, VB = 16;
    CT_DYN_SMEM(smem_raw);
    float* prob = reinterpret_cast<float*>(smem_raw);
    __shared__ double red[NWV];
    __shared__ float redf[NWV];
    __shared__ float part[NWV][16];
    __shared__ double tailp[4][16];
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = uniform_int(wave_id()), j = tid & 3, quad = tid >> 2;
    int h, grp;
    {
        const int b = (int)blockIdx.x, rep = n_head / n_head_kv;
        if ((n_head_kv & 7) == 0) {
            const int i = b >> 3, per = rep * ng;
            const int hkv = (b & 7) + 8 * (i / per), r = (i / ng) % rep;
            h = hkv * rep + r; grp = i % ng;
        } else { h = b / ng; grp = b - h * ng; }
    }
    const bool trace = a.trace && blockIdx.x == 0 && lane == 0;   // measurement only (tools/attn_trace_ctx.py): stamps of workgroup 0's waves
    unsigned long long* tr = a.trace + 16 * wv;
    if (trace) tr[0] = clock64_dev();
    const int hk = h / (n_head / n_head_kv);
    const int CG = HD / ng / 16, SPL = NWV / CG;   // channel groups of 16 in this workgroup (1, 2 or 4), position slices per group
    const int cg = wv % CG, s = wv / CG;
    // ---- requests that do not depend on the cursor: the query, this wave's first VB chunks of V, (no SHARE) the first K rows ----
    const uint16_t* kbase = a.kcache + (size_t)hk * a.n_ctx * HD + 8 * j;
    const int d = grp * (HD / ng) + cg * 16 + (lane >> 2);
    const uint16_t* vrow = a.vcache + ((size_t)hk * HD + d) * a.v_stride;
    u32x4 vb[VB], kb[PB * NC], qv[NC], tailv[4];
    auto chunk_of = [&](int k) __attribute__((always_inline)) { return 2 * (s + SPL * (k >> 1)) + (k & 1); };   // this wave's k-th chunk: pairs 2 m, 2 m + 1
    // Request order: the query and the K rows first — they are what the first dependent work waits for; V follows behind the scores (below).
    {
        const uint16_t* qrow = a.q_f16 + (size_t)h * HD;
#pragma unroll
        for (int c = 0; c < NC; ++c) qv[c] = ld16(qrow + 32 * c + 8 * j);
    }
    if constexpr (!SHARE) {
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            int p = u * NQ + quad;
            p = p < a.n_ctx ? p : a.n_ctx - 1;
            const uint16_t* krow = kbase + (size_t)p * HD;
#pragma unroll
            for (int c = 0; c < NC; ++c) kb[u * NC + c] = ld16(krow + 32 * c);
        }
    }
    auto vreq = [&](int u) __attribute__((always_inline)) {
        const int off = 32 * chunk_of(u);
        vb[u] = ld16(vrow + (off + 32 <= a.v_stride ? off : a.v_stride - 32) + 8 * j);
    };
    int cur[4];
    uint32_t tag = 0u;
    if constexpr (SHARE) tag = (((uint32_t)sload_i32x4_and(cur0, cur, (const int*)a.epoch) + 1u) << 8) | (uint32_t)a.layer;
    else sload_i32x4(cur0, cur);
    const int n_kv = cur[1] + 1;
    int n_tot = cur[2];
    if (cur[3] > 0) {   // the reference batch this token belongs to (attn_fused_exact_kernel)
        const int idx = cur[0], base = cur[1] - cur[0];
        const int end = (idx / cur[3] + 1) * cur[3], n_eval = n_tot - base;
        n_tot = base + (end < n_eval ? end : n_eval);
    }
    const int np = n_tot & ~31, nl = n_kv - np, nchunk = np >> 5;
    const int last_c = np >= 32 ? np - 32 : 0, last_p = n_kv - 1;
    const int per = SHARE ? (n_kv + ng - 1) / ng : n_kv;
    const int p0 = SHARE ? grp * per : 0, p1 = SHARE ? (p0 + per < n_kv ? p0 + per : n_kv) : n_kv;
    if constexpr (SHARE) {
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int p = p0 + u * NQ + quad;
            const uint16_t* krow = kbase + (size_t)(p < last_p ? p : last_p) * HD;
#pragma unroll
            for (int c = 0; c < NC; ++c) kb[u * NC + c] = ld16(krow + 32 * c);
        }
    }
    // The exp table's negative half (64 KB; every softmax input is <= 0) is touched once per XCD — workgroups 0 .. 7 sit on the eight XCDs — so that
    // the look-ups behind the row maximum find it in their L2: between two layers 100+ MB of weights and K / V rows have passed through it.
    const uint32_t warm = *(const uint32_t*)(a.exp_tab + 0x8000 + (blockIdx.x < 8u ? 64 * tid : 0));   // (unconditional: no branch around a load)
    if (s == SPL - 1) {   // the tail positions' values (the row is padded: np + 31 stays inside the cache); the last slice's wave runs the tail
#pragma unroll
        for (int c = 0; c < 4; ++c) tailv[c] = ld16(vrow + np + 8 * c);
    }
    if (trace) { tr[1] = clock64_dev(); tr[7] = (unsigned long long)n_kv; }
    // ---- scores (per position: the bit-identical kernel's) ----
    float mx = -INFINITY;
    {
        auto slot = [&](int u, int base, auto REQ) __attribute__((always_inline)) {
            const int p = base + u * NQ + quad;
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < NC; ++c) fma8_hh(acc, kb[u * NC + c], qv[c]);
            if constexpr (decltype(REQ)::value) {
                const int pn = p + NQ * PB;
                const uint16_t* krow = kbase + (size_t)(pn < last_p ? pn : last_p) * HD;
#pragma unroll
                for (int c = 0; c < NC; ++c) kb[u * NC + c] = ld16(krow + 32 * c);
            }
            const float sc = f16dot_reduce_exact(acc, j) * a.kq_scale;
            if constexpr (SHARE) {
                if (p < p1 && j == 0) st_granule(a.xs + ((size_t)h * a.n_ctx + p) * 2, f32_to_bits(sc), tag);
            } else if (p < n_kv) {
                mx = fmaxf(mx, sc);
                if (j == 0) prob[p] = sc;
            }
        };
        int base = p0;
        for (; base + NQ * PB < p1; base += NQ * PB) {
#pragma unroll
            for (int u = 0; u < PB; ++u) slot(u, base, A9Req<true>{});
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) slot(u, base, A9Req<false>{});
    }
    // The V chunks leave HERE, behind this wave's score computation: a CU keeps about 32 KB of requests in flight, and V requested at kernel entry (the
    // first form of this kernel) took that budget from the K rows of the waves that start a little later — their scores, and with them the whole head's
    // row gather, ended 4 000 cycles late.  Behind the scores the V bytes travel beside the gather and the softmax, which wait on latency, not bandwidth.
#pragma unroll
    for (int u = 0; u < VB; ++u) vreq(u);
    if (trace) tr[2] = clock64_dev();
    if constexpr (SHARE) {   // gather the head's row (attn_decode9_kernel<.., SHARE>)
        const uint32_t* row = a.xs + (size_t)h * a.n_ctx * 2;
        const unsigned long long t0 = wall_ticks();
        (void)t0;
        bool gave_up = false;
        for (int i0 = 0; i0 < n_kv && !gave_up; i0 += 8 * NT) {
            for (;;) {
                uint32_t dat[8], tg[8];
                bool ok = true;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * NT + tid;
                    dat[u] = 0u; tg[u] = tag;
                    if (i < n_kv) ld_granule(row + 2 * i, dat[u], tg[u]);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) ok = ok && tg[u] == tag;
                if (__ballot(!ok) == 0ull) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + u * NT + tid;
                        if (i < n_kv) { const float sc = bits_to_f32(dat[u]); prob[i] = sc; mx = fmaxf(mx, sc); }
                    }
                    break;
                }
#ifdef CT_EMU
                emu::spin_yield();
#else
                __builtin_amdgcn_s_sleep(1);
                if (wall_ticks() - t0 > 2000000ull) { if (lane == 0) *a.err = 1; gave_up = true; break; }
#endif
            }
        }
    }
    mx = fmaxf(mx, lane_xor1(mx)); mx = fmaxf(mx, lane_xor2(mx));
    mx = fmaxf(mx, lane_xor4(mx)); mx = fmaxf(mx, lane_xor8(mx)); mx = fmaxf(mx, lane_xor16(mx)); mx = fmaxf(mx, lane_xor32(mx));
    if (lane == 0) redf[wv] = mx;
    __syncthreads();
    mx = redf[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) mx = fmaxf(mx, redf[w]);
    if (trace) tr[3] = clock64_dev();
#ifndef CT_EMU
    asm volatile("" :: "v"(warm));   // (the touch is a real load: its value is "used" here, long after it has landed)
#else
    (void)warm;
#endif
    // ---- softmax: fp16 exp table, exact double sum, fp16 probabilities ----
    double sum = 0.0;
    constexpr int SB = 4;
    for (int i0 = 0; i0 < n_kv; i0 += NT * SB) {
        uint16_t e16[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) { const int i = i0 + u * NT + tid; e16[u] = i < n_kv ? a.exp_tab[f32_to_f16_bits(prob[i] - mx)] : (uint16_t)0; }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int i = i0 + u * NT + tid;
            if (i < n_kv) { const float e = f16_bits_to_f32(e16[u]); prob[i] = e; sum += (double)e; }
        }
    }
    sum = wave_sum_fast(sum);
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    double tot = red[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) tot += red[w];
    const float inv = (float)(1.0 / tot);
    for (int i = tid; i < n_kv; i += NT) prob[i] = f16_bits_to_f32(f32_to_f16_bits(prob[i] * inv));
    for (int i = n_kv + tid; i < np; i += NT) prob[i] = 0.0f;   // masked columns of this batch
    __syncthreads();
    if (trace) tr[4] = clock64_dev();
    // ---- V*P: this wave's chunks s, s + SPL, .. of its sixteen channels (a quad per channel) ----
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int npair = nchunk > 2 * s ? (nchunk - 2 * s + 2 * SPL - 1) / (2 * SPL) : 0;   // pairs of this wave whose first chunk exists
    const int nk = npair > 0 ? 2 * npair - (2 * (s + SPL * (npair - 1)) + 1 >= nchunk ? 1 : 0) : 0;   // its chunks
    auto step = [&](int u, int k) __attribute__((always_inline)) {
        const float* pr = &prob[32 * chunk_of(k) + 8 * j];
        float pc[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) pc[l] = pr[l];
        fma8_hf(acc, vb[u], pc);
    };
    int k0 = 0;
    for (; k0 + VB < nk; k0 += VB) {   // every chunk of this round exists and more follow: each slot is re-requested (clamped to the last chunk)
#pragma unroll
        for (int u = 0; u < VB; ++u) {
            step(u, k0 + u);
            const int in = 32 * chunk_of(k0 + u + VB);
            vb[u] = ld16(vrow + (in < last_c ? in : last_c) + 8 * j);
        }
    }
#pragma unroll
    for (int u = 0; u < VB; ++u)
        if (k0 + u < nk) step(u, k0 + u);
    const float res = f16dot_reduce_exact(acc, j);
    if (trace) tr[5] = clock64_dev();
    if (j == 0) part[wv][lane >> 2] = res;
    if (s == SPL - 1) {   // the scalar tail: its own chain from 0.0 (with no 32-position step at all that IS the reference's sum)
        double tl = 0.0;
        if (nl > 0) tl = f16_tail32(tl, tailv, prob + np, nl);
        if (j == 0) tailp[cg][lane >> 2] = tl;
    }
    __syncthreads();
    if (s != 0) return;
    double sumf = (double)part[cg][lane >> 2];
    for (int t = 1; t < SPL; ++t) sumf += (double)part[cg + CG * t][lane >> 2];
    sumf += tailp[cg][lane >> 2];
    if (j == 0) a.out[(size_t)h * HD + d] = (float)sumf;
    if (trace) tr[6] = clock64_dev();
}
