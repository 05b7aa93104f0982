// Decode, llama graph: QKV mat-vec -> RoPE -> KV append -> attention in ONE launch per layer (reference chain llm_build_llama,
// llama.cpp:2291-2402: the three ggml_mul_mat, ggml_rope, the ggml_cpy into the cache, KQ, scale, mask, soft_max, KQV).
//
// Why: the separate attention launch is a latency chain, not bytes (2.3 MB at position 140 in 5.3-5.9 us: boundary, cursor, query and
// first K rows, scores, two barriers, exp table, V*P); of it the boundary and the entry go away when the workgroups that produce a head's
// q / k / v rows are the ones that need them.  How:
//   * grid = the attention grid of kernels_attn9.h: n_head x ng workgroups of 1024 threads, (head, channel group) -> workgroup by the same
//     map (the workgroups of a KV head take consecutive positions on one XCD: for speed only).  The rep x ng workgroups of KV head g are a
//     GROUP: phase 1 deals the group's rows — the q rows of its rep query heads, its k rows, its v rows — to the group's waves
//     (GroupItems9: 24 units per workgroup on a 7B, what the plain launch gives a workgroup too) and runs generation 9's mat-vec on them
//     (kernels_v9.h:v9_run: same prologue, ring, block math, chain and epilogues — bit for bit).
//   * the hand-off inside the group is a GRANULE EXCHANGE, the placement-independent form of MI355X_MICROARCH.md ("handoff-1to1",
//     "allgather"): an epilogue lane pair stores its two fp16 results {row 2u | row 2u + 1} with a TAG as ONE naturally aligned 8-byte
//     device-scope (sc1, write-through) store into the group's exchange record; no flag, no counter, no fence, no ordering between
//     granules.  ONE wave per workgroup sweeps the granules its workgroup needs (the head's q, the k row, its channels of v: 136 of them on
//     a 7B) with sc1 loads until every tag is this launch's, and parks the payload in LDS.  The tag is (token epoch, layer): the epoch word
//     is advanced by the launch that advances the cursor, so a granule of an earlier token step never matches.  The KV cache is written
//     with plain stores as before (for the tokens to come: a kernel boundary lies between); nothing of THIS launch reads it at the
//     position being written — the score of the new position uses the k row from the exchange, the V*P product the v value from it — so
//     no workgroup ever depends on another workgroup's cache lines, on XCD placement or on L2 state.
//   * while the exchange is under way the waves already hold what does not depend on it: the K rows of the older positions and the V
//     chunks are requested as soon as a wave's last record step is done (the cursor is known since kernel entry).
//   * phase 2 is attn_decode9_kernel's arithmetic (kernels_attn9.h; reference ggml_vec_dot_f16 / soft_max / the V*P dot with its
//     double-precision leftovers) with 15 - (pv waves - 1) score waves instead of seven.
// Co-residency: the sweep waits for granules other workgroups of the group produce, so all of a group's workgroups must be resident.  The
// host launches this kernel only with grid <= CUs of the device at one 1024-thread workgroup per CU (engine.cc:qa_can), and the sweep gives
// up after 20 ms: it raises QaArgs::err, the host reports the eval as failed and the handle goes on with the two-launch form.  Seen on
// MI355X: ANY other resident wave breaks residency for the 127-register instantiations (four of their waves fill a SIMD's register
// file) — e.g. the polling shader the runtime parks on the device for a hipStreamWaitValue32 of another stream (pipeline.cc: why stages
// that share a device hand over by events).
// Test builds (CT_EMU) run workgroups one after the other: the host launches phase 1 and phase 2 as two passes (QaArgs::phase).
#pragma once
#include "kernels_v9.h"
#include "kernels_attn9.h"

// Everything phase 2 needs beyond MatvecArgs (K / V cache, cursor, context and row strides are the mat-vec launch's own fields).  Kept
// small on purpose: with the attention launch's whole argument block beside MatvecArgs the mat-vec phase ran out of scalar registers and
// re-read kernel arguments inside its epilogue — 1.5 us per launch, measured (profiles/r05_qa_fusion.txt).
struct QaArgs {
    uint32_t* xq;            // exchange records: [n_head_kv][(rep + 2) * head_dim / 2] granules of {payload, tag}
    const unsigned* epoch;   // token-step counter (advanced with the cursor)
    int* err;                // pinned host word: raised when a sweep times out
    float* out;              // attention output f32[n_embd]
    const uint16_t* exp_tab; // fp16 exp table (ggml.c:4318-4332)
    unsigned long long* trace;   // measurement only
    float kq_scale;
    int n_head, n_head_kv;
    int layer;               // tag = (epoch + 1) << 8 | layer  (never 0: the records start zeroed)
    int ng;                  // channel groups per head (kernels_attn9.h); a power of two, as is rep = n_head / n_head_kv (engine.cc:qa_can):
    int ng_sh, rep_sh;       // their logarithms — the workgroup map is shifts and masks (with divisions it was 130 scalar instructions in front of
                             // every wave's first weight request)
    int phase;               // 0: the product form; 1 / 2: mat-vec only / attention only (CPU emulation of the kernels: two passes)
};

constexpr int kQaXw = 200;   // payload words a workgroup parks: head_dim / 2 (q) + head_dim / 2 (k) + channels / 2 (v) <= 64 + 64 + 32
struct QaSmem {
    uint32_t xw[kQaXw];      // q pairs | k pairs | v pairs of this workgroup's channels
    double red[16];
    float redf[16];
};

// The units of one wave of the fused launch: at most two (host: engine.cc:qa_can) — local units wl and wl + gw of the group's list, as
// items of the launch's unit list; their granules.  Few scalars on purpose (see QaArgs).
struct QaItems {
    int n, it0, it1;      // count, the two items
    uint32_t* g0;         // granule of the first unit; the second one's is gstep words further
    int gstep;
    uint32_t tag;
    int pos;
    const QaArgs* qa;     // (kernel-argument segment)
    DEV int count() const { return n; }
    DEV int at(int k) const { return k ? it1 : it0; }
    // called by v9_run once the wave's first records are requested: the position and the token epoch through the scalar cache, both
    // loads in flight together (a scalar round trip at the top of the kernel would hold up every wave's first requests: 742 -> 601 tok/s)
    DEV void load_scalars(const MatvecArgs& a) {
#ifdef CT_EMU
        pos = a.pos[0];
        tag = ((uint32_t)(*qa->epoch + 1u) << 8) | (uint32_t)qa->layer;
#else
        int p;
        unsigned e;
        asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(p), "=&s"(e) : "s"(a.pos), "s"(qa->epoch) : "memory");
        pos = p;
        tag = ((e + 1u) << 8) | (uint32_t)qa->layer;
#endif
    }
    DEV int cursor_pos() const { return pos; }
    DEV void publish(int k, uint32_t data) const { st_granule(g0 + (k ? gstep : 0), data, tag); }
};

DEV void patch_f16(u32x4& v, int elem, uint32_t h) {   // fp16 element `elem` (0..7) of a 16-byte operand := h
    const int w = elem >> 1;
    const uint32_t keep = (elem & 1) ? 0x0000FFFFu : 0xFFFF0000u, ins = (elem & 1) ? (h << 16) : h;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k == w) v[k] = (v[k] & keep) | ins;
}

// TA: type of wq / wk; TB: type of wv when it differs (GT_Q6_K), else 0.  NWV score waves, 16 - NWV V*P waves of EIGHT channels: an OCTET of
// lanes per channel — lane (j, hh) of it owns accumulators 4 hh .. 4 hh + 3 of quad-lane j of kernels_attn9.h's V*P (positions 8 j + 4 hh .. + 3
// of every 32-step): the same thirty-two fma chains, the same reduction tree (the j exchanges two lane bits up, t_m = S[m] + S[m + 4] is the
// exchange between the halves), half the instructions per wave and step — the lone V*P wave of a workgroup was 3 800 of the attention
// phase's 8 600 cycles at 200 positions.
#ifndef QA_VB
#define QA_VB 6   // V chunk slots per V*P lane: 4 -> 773.3, 6 -> 776.1, 8 (six registers spilled) -> 765.8 tok/s over 256 steps (positions 144..400), alternating on one box
#endif
// LN (falcon graph, llama.cpp:2652-2700): LayerNorm prologue; with MatvecArgs::rope_neox the q / k granule u of a head carries the NEOX pair
// (row u | row u + HD / 2) of the reordered matrix — the sweep parks its halves at elements u and u + HD / 2, so the K.Q dot reads the vectors in file order.
template <int TA, int TB, int HD, int NWV, int NS, bool LN = false>
__global__ void __launch_bounds__(1024) qkv_attn9_kernel(const float* x0, const float* nw0, int K0, int pro0, const MatvecArgs a, const QaArgs q) {   // (leading scalars: kernels_v9.h:matvec_v9_kernel)
    constexpr int MAXK = 16384;
    constexpr int PB = 2, VB = QA_VB;
    constexpr int NT = 64 * NWV, NQ = NT / 4, NC = HD / 32;
    CT_DYN_SMEM(smem_raw);
    SmemV9<MAXK>& SM = *reinterpret_cast<SmemV9<MAXK>*>(smem_raw);
    QaSmem& QS = *reinterpret_cast<QaSmem*>(smem_raw + ((sizeof(SmemV9<MAXK>) + 15) & ~(size_t)15));
    float* prob = reinterpret_cast<float*>(smem_raw + ((sizeof(SmemV9<MAXK>) + 15) & ~(size_t)15) + ((sizeof(QaSmem) + 15) & ~(size_t)15));
    // What maps this workgroup to (head, channel group, KV-head group) is computed TWICE — for phase 1's unit list and again, from an
    // opaque copy of the block index, for phase 2: nothing of it (nor the trace pointer, the record address, ...) then lives through the
    // streaming loop, whose scalar registers are all taken (a mat-vec phase that ran out of them re-read kernel arguments in its
    // epilogue: + 1.5 us per launch, profiles/r05_qa_fusion.txt).
    struct Map { int h, grp, hk, r, rep, NG, gidx, lu_all; };
    auto map_of = [&](int b) __attribute__((always_inline)) {
        Map m;
        const int ng = q.ng, ngs = q.ng_sh, rs = q.rep_sh;
        m.rep = 1 << rs;
        if ((q.n_head_kv & 7) == 0) {   // kernels_attn9.h's map
            const int i = b >> 3;
            const int hkv0 = (b & 7) + 8 * (i >> (rs + ngs)), r0 = (i >> ngs) & (m.rep - 1);
            m.h = (hkv0 << rs) + r0; m.grp = i & (ng - 1);
        } else { m.h = b >> ngs; m.grp = b & (ng - 1); }
        m.hk = m.h >> rs; m.r = m.h & (m.rep - 1);
        m.NG = m.rep << ngs; m.gidx = (m.r << ngs) + m.grp;   // workgroups of the group, this one's index in it
        m.lu_all = (m.rep + 2) * (HD / 2);                    // granules (= units) of a group
        return m;
    };

    // ---------------------------------------------------------------- phase 1: the group's q / k / v rows (kernels_v9.h)
#ifdef CT_EMU
    const bool run1 = !(q.phase & 2);
#else
    constexpr bool run1 = true;   // (phase 2 alone is the CPU emulation's second pass: the product reads no argument in front of the activation requests)
#endif
    if (run1) {
        const int lane = lane_id(), wv = uniform_int(wave_id());
        Pro9<MAXK, TB == 0, 16> P;
        pro9_load<MAXK, TB == 0, 16>(P, x0, nw0, K0, pro0, wv, lane);   // the activation requests FIRST: nothing in front of them
        kernarg_touch<24 + sizeof(MatvecArgs) + sizeof(QaArgs)>();   // (gpu.h) behind them: one round trip for every argument line
        if (q.trace && blockIdx.x == 0 && lane == 0) q.trace[16 * wv] = clock64_dev();
        if (threadIdx.x == 0) SM.L.cnt = 0u;
        __syncthreads();
        const Map M1 = map_of(uniform_int(opaque_int((int)blockIdx.x)));   // (the map's integer divisions run under the activations' latency)
        const int rep = M1.rep, hk = M1.hk, NG = M1.NG, gidx = M1.gidx, lu_all = M1.lu_all;
        uint32_t* rec = q.xq + (size_t)hk * lu_all * 2;
        auto pro = [&](bool trc, unsigned long long (&ts)[4]) __attribute__((always_inline)) {
            pro9_finish<MAXK, LN, false, TB == 0, TA == GT_Q6_K || TB == GT_Q6_K, 16>(SM.L, P, a.norm_w, a.norm_b, a.K, a.pro, a.eps, nullptr, wv, lane, trc, ts);
        };
        const int uq = rep * (HD / 2), uk = HD / 2;            // q / k units of the group
        const int nq_all = a.job[1].pair0, nqk_all = a.job[2].pair0;
        QaItems it;
        it.qa = &q; it.tag = 0u; it.pos = 0;
        // local unit li of the group's list {q rows | k rows | v rows} -> item of the launch's unit list; granule li of the record
        auto item_of = [&](int li) { return li < uq ? hk * uq + li : (li < uq + uk ? nq_all + hk * uk + (li - uq) : nqk_all + hk * uk + (li - uq - uk)); };
        int wl, gw, lu, off;   // this wave's first local unit, the stride to its second, the list's length, the list's first local unit
        if (TB != 0 && wv >= a.nwA) { wl = (wv - a.nwA) * NG + gidx; gw = (16 - a.nwA) * NG; lu = uk; off = uq + uk; }   // the v rows (second weight type)
        else if (TB != 0) { wl = wv * NG + gidx; gw = a.nwA * NG; lu = uq + uk; off = 0; }
        else { wl = wv * NG + gidx; gw = 16 * NG; lu = lu_all; off = 0; }
        it.n = wl < lu ? (wl + gw < lu ? 2 : 1) : 0;
        it.it0 = item_of(off + (wl < lu ? wl : 0));
        it.it1 = item_of(off + (wl + gw < lu ? wl + gw : 0));
        it.g0 = rec + 2 * (off + wl);
        it.gstep = 2 * gw;
        if constexpr (TB != 0) {
            if (wv < a.nwA) v9_run<TA, MAXK, true, NS, false, true>(a, SM, a.baseA, 0, it, lane, wv, pro);
            else v9_run<TB, MAXK, true, NS, false, true>(a, SM, a.baseB, a.n_groupA, it, lane, wv, pro);
        } else {
            v9_run<TA, MAXK, false, NS, false, true>(a, SM, a.baseA, 0, it, lane, wv, pro);
        }
    }
    if (q.phase & 1) return;
    // the map, the cursor {step, pos, n_past + n, batch} and the tag: computed / read again here rather than carried through the mat-vec phase
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = uniform_int(wave_id());
    const Map M2 = map_of(uniform_int(opaque_int((int)blockIdx.x)));
    const int h = M2.h, grp = M2.grp, hk = M2.hk, r = M2.r, rep = M2.rep, ng = q.ng, lu_all = M2.lu_all;
    const uint32_t* rec = q.xq + (size_t)hk * lu_all * 2;
    const bool trace = q.trace && blockIdx.x == 0 && lane == 0;
    unsigned long long* tr = q.trace + 16 * wv;
    int cur[4];
    const uint32_t tag = (((uint32_t)sload_i32x4_and(a.pos - 1, cur, (const int*)q.epoch) + 1u) << 8) | (uint32_t)q.layer;   // one scalar round trip for both
    if (trace) tr[1] = clock64_dev();   // this wave's rows are done

    // ---------------------------------------------------------------- phase 2: attention of (head h, channel group grp)
    const int pos = cur[1], n_kv = pos + 1;
    int n_tot = cur[2];
    if (cur[3] > 0) {   // the reference batch this token belongs to (kernels_attn9.h)
        const int idx = cur[0], base = cur[1] - cur[0];
        const int end = (idx / cur[3] + 1) * cur[3], n_eval = n_tot - base;
        n_tot = base + (end < n_eval ? end : n_eval);
    }
    const int np = n_tot & ~31, nl = n_kv - np;
    const int last_c = np >= 32 ? np - 32 : 0;
    const int j = tid & 3, quad = tid >> 2;
    const bool pv_wave = wv >= NWV;
    constexpr int NBUF = PB * NC > VB ? PB * NC : VB;
    static_assert(NC <= 4, "register set");
    u32x4 buf[NBUF], aux[4];
    const int chg = HD / ng;                                   // channels of this workgroup
    const int pj = (lane >> 1) & 3, hh = lane & 1;             // V*P waves: octet lane = (slice j of the 32-step, accumulator half)
    const int d = grp * chg + (pv_wave ? wv - NWV : 0) * 8 + (lane >> 3);
    const uint16_t* kbase = a.kcache + (size_t)hk * a.n_ctx * HD + 8 * j;
    const uint16_t* vrow = a.vcache + ((size_t)hk * HD + d) * a.v_stride;
    // what does not depend on the exchange: the K rows of the older positions (row `pos` itself arrives through the exchange: an address
    // clamped to the row before it — row 0 for the very first token, whose copy nobody uses), the V chunks (their element at `pos` is
    // replaced from the exchange below)
    const int old_p = pos > 0 ? pos - 1 : 0;
    u32x2 vb[VB];
    if (pv_wave) {
#pragma unroll
        for (int u = 0; u < VB; ++u) vb[u] = *(const u32x2*)(vrow + (32 * u < last_c ? 32 * u : last_c) + 8 * pj + 4 * hh);
#pragma unroll
        for (int c = 0; c < 4; ++c) aux[c] = ld16(vrow + np + 8 * c);
    } else {
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int p = u * NQ + quad;
            const uint16_t* krow = kbase + (size_t)(p < old_p ? p : old_p) * HD;
#pragma unroll
            for (int c = 0; c < NC; ++c) buf[u * NC + c] = ld16(krow + 32 * c);
        }
    }
    // ---- the sweep: one wave gathers this workgroup's granules into LDS ----
    if (wv == 15) {
        const int nqg = HD / 2, nvg = chg / 2, need = 2 * nqg + nvg;
        int gi[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int i = lane + 64 * t;
            gi[t] = i < nqg ? r * nqg + i : (i < 2 * nqg ? rep * nqg + (i - nqg) : (rep + 1) * nqg + grp * nvg + (i - 2 * nqg));
            if (i >= need) gi[t] = -1;
        }
        const unsigned long long t0 = wall_ticks();
        (void)t0;
        for (;;) {
            uint32_t dat[3], tg[3];
            bool ok = true;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                dat[t] = 0u; tg[t] = tag;
                if (gi[t] >= 0) ld_granule(rec + 2 * gi[t], dat[t], tg[t]);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) ok = ok && tg[t] == tag;
            if (__ballot(!ok) == 0ull) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int i = lane + 64 * t;
                    if (gi[t] < 0) continue;
                    if (LN && a.rope_neox && i < 2 * nqg) {   // q / k granule u of the head: elements u and u + HD / 2 of the vector
                        uint16_t* v16 = reinterpret_cast<uint16_t*>(QS.xw) + (i < nqg ? 0 : HD);
                        const int u = i < nqg ? i : i - nqg;
                        v16[u] = (uint16_t)(dat[t] & 0xFFFFu);
                        v16[u + HD / 2] = (uint16_t)(dat[t] >> 16);
                    } else {
                        QS.xw[i] = dat[t];
                    }
                }
                break;
            }
#ifdef CT_EMU
            emu::spin_yield();
#else
            __builtin_amdgcn_s_sleep(1);
            if (wall_ticks() - t0 > 2000000ull) {   // 20 ms: some workgroup of the group is not running (a shared device)
                if (lane == 0) *q.err = 1;
                break;
            }
#endif
        }
    }
    if (trace) tr[2] = clock64_dev();   // (wave 15: the sweep is complete)
    __syncthreads();
    if (trace) { tr[3] = clock64_dev(); tr[7] = (unsigned long long)n_kv; }
    // ---- scores ----
    const uint16_t* q16 = reinterpret_cast<const uint16_t*>(QS.xw);
    const uint16_t* k16 = q16 + HD;
    float mx = -INFINITY;
    if (!pv_wave) {
#pragma unroll
        for (int c = 0; c < NC; ++c) aux[c] = *(const u32x4*)(q16 + 32 * c + 8 * j);
        auto slot = [&](int u, int base, auto REQ) __attribute__((always_inline)) {
            const int p = base + u * NQ + quad;
            if (p == pos) {   // the new position's k row: from the exchange
#pragma unroll
                for (int c = 0; c < NC; ++c) buf[u * NC + c] = *(const u32x4*)(k16 + 32 * c + 8 * j);
            }
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < NC; ++c) fma8_hh(acc, buf[u * NC + c], aux[c]);
            if constexpr (decltype(REQ)::value) {
                const int pn = p + NQ * PB;
                const uint16_t* krow = kbase + (size_t)(pn < old_p ? pn : old_p) * HD;
#pragma unroll
                for (int c = 0; c < NC; ++c) buf[u * NC + c] = ld16(krow + 32 * c);
            }
            const float sc = f16dot_reduce_exact(acc, j) * q.kq_scale;
            if (p < n_kv) {
                mx = fmaxf(mx, sc);
                if (j == 0) prob[p] = sc;
            }
        };
        int base = 0;
        for (; base + NQ * PB < n_kv; base += NQ * PB) {
#pragma unroll
            for (int u = 0; u < PB; ++u) slot(u, base, A9Req<true>{});
        }
#pragma unroll
        for (int u = 0; u < PB; ++u)
            if (base + u * NQ < n_kv) slot(u, base, A9Req<false>{});   // (the last pass requests nothing: a slot without positions is skipped)
    }
    if (trace) tr[8] = clock64_dev();   // scores
    mx = fmaxf(mx, lane_xor4(mx)); mx = fmaxf(mx, lane_xor8(mx)); mx = fmaxf(mx, lane_xor16(mx)); mx = fmaxf(mx, lane_xor32(mx));
    if (lane == 0 && !pv_wave) QS.redf[wv] = mx;
    __syncthreads();
    if (trace) tr[9] = clock64_dev();   // max known
    mx = QS.redf[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) mx = fmaxf(mx, QS.redf[w]);
    // ---- softmax: fp16 exp table, order-free double sum (kernels_attn9.h) ----
    double sum = 0.0;
    constexpr int SB = 4;
    for (int i0 = 0; !pv_wave && i0 < n_kv; i0 += NT * SB) {
        uint16_t e16[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) { const int i = i0 + u * NT + tid; e16[u] = i < n_kv ? q.exp_tab[f32_to_f16_bits(prob[i] - mx)] : (uint16_t)0; }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int i = i0 + u * NT + tid;
            if (i < n_kv) { const float e = f16_bits_to_f32(e16[u]); prob[i] = e; sum += (double)e; }
        }
    }
    if (!pv_wave) {
        sum = wave_sum_fast(sum);
        if (lane == 0) QS.red[wv] = sum;
    }
    __syncthreads();
    double tot = QS.red[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) tot += QS.red[w];
    const float inv = (float)(1.0 / tot);
    if (!pv_wave) {
        for (int i = tid; i < n_kv; i += NT) prob[i] = f16_bits_to_f32(f32_to_f16_bits(prob[i] * inv));
        for (int i = n_kv + tid; i < np; i += NT) prob[i] = 0.0f;
    }
    __syncthreads();
    if (trace) tr[4] = clock64_dev();
    if (!pv_wave) return;
    // ---- V*P: an octet per channel; the value at `pos` comes from the exchange ----
    const int dl = (wv - NWV) * 8 + (lane >> 3);                        // channel inside this workgroup's group
    const uint32_t vnew = (QS.xw[HD + (dl >> 1)] >> ((dl & 1) * 16)) & 0xFFFFu;
    const bool in_fma = pos < np;   // `pos` lies in the 32-step part (n_kv a whole number of steps, or a token in the middle of a reference batch)
    if (!in_fma) {   // ... or among the leftover positions np .. n_kv - 1
        const int off = pos - np;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c == (off >> 3)) patch_f16(aux[c], off & 7, vnew);
    }
    const int pos_c = pos & ~31;
    const bool my_new = in_fma && pj == ((pos & 31) >> 3) && hh == ((pos & 7) >> 2);   // this lane's four positions of chunk pos_c hold `pos`
    auto patch4 = [&](u32x2& v) __attribute__((always_inline)) {
        const int e = pos & 3;
        const uint32_t keep = (e & 1) ? 0x0000FFFFu : 0xFFFF0000u, ins = (e & 1) ? (vnew << 16) : vnew;
        if (e >> 1) v[1] = (v[1] & keep) | ins; else v[0] = (v[0] & keep) | ins;
    };
    float acc[4] = {0, 0, 0, 0};
    int i0 = 0;
    float pn[4];
    {
        const float* p0 = &prob[8 * pj + 4 * hh];
#pragma unroll
        for (int l = 0; l < 4; ++l) pn[l] = p0[l];
    }
    for (; i0 + 32 * VB < np; i0 += 32 * VB) {
#pragma unroll
        for (int u = 0; u < VB; ++u) {
            const int i = i0 + 32 * u;
            float pc[4];
#pragma unroll
            for (int l = 0; l < 4; ++l) pc[l] = pn[l];
            {
                const float* pr = &prob[i + 32 + 8 * pj + 4 * hh];
#pragma unroll
                for (int l = 0; l < 4; ++l) pn[l] = pr[l];
            }
            if (my_new && i == pos_c) patch4(vb[u]);
            fma4_hf(acc, vb[u][0], vb[u][1], pc);
            const int in = i + 32 * VB;
            vb[u] = *(const u32x2*)(vrow + (in < last_c ? in : last_c) + 8 * pj + 4 * hh);
        }
    }
#pragma unroll
    for (int u = 0; u < VB; ++u) {
        const int i = i0 + 32 * u;
        if (i < np) {
            float pc[4];
#pragma unroll
            for (int l = 0; l < 4; ++l) pc[l] = pn[l];
            {
                const int inx = i + 32 < last_c ? i + 32 : last_c;
                const float* pr = &prob[inx + 8 * pj + 4 * hh];
#pragma unroll
                for (int l = 0; l < 4; ++l) pn[l] = pr[l];
            }
            if (my_new && i == pos_c) patch4(vb[u]);
            fma4_hf(acc, vb[u][0], vb[u][1], pc);
        }
    }
    // the reference's reduce (ggml.c:1964-1982; kernels_exact.h:f16dot_reduce_exact): S[l] = (s_j0[l] + s_j2[l]) + (s_j1[l] + s_j3[l]) — j sits in
    // lane bits 1..2 here —, t_m = S[m] + S[m + 4] — the two halves, lane bit 0 —, res = (t0 + t1) + (t2 + t3)
    float S[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const float x = acc[l] + lane_xor4(acc[l]);
        S[l] = x + lane_xor2(x);
    }
    const float t0 = S[0] + lane_xor1(S[0]), t1 = S[1] + lane_xor1(S[1]), t2 = S[2] + lane_xor1(S[2]), t3 = S[3] + lane_xor1(S[3]);
    const float res = (t0 + t1) + (t2 + t3);
    double sumf = (double)res;
    if (nl > 0) sumf = f16_tail32(sumf, aux, prob + np, nl);
    if ((lane & 7) == 0) q.out[(size_t)h * HD + d] = (float)sumf;
    if (trace) tr[6] = clock64_dev();
}
