// Mat-vec for the 32-element block types (Q8_0, Q4_0 weights x Q8_0 activations), bit-identical to the reference's
// AVX2 build:
//   quantize_row_q8_0 (ggml.c:1208-1300 AVX2 path): per 32: d = amax/127 stored as fp16, id = 127/amax (0 when amax == 0),
//       q = round-half-even(x * id);
//   ggml_vec_dot_q8_0_q8_0 (ggml.c:3321, AVX2) / ggml_vec_dot_q4_0_q8_0 (ggml.c:2428, AVX2): 8 f32 lanes,
//       acc[l] = fma(fp16(x.d) * fp16(y.d), (float)sumi[l], acc[l]) block after block, sumi[l] = the four products of
//       elements 4l..4l+3 (mul_sum_i8_pairs_float), Q4_0 elements = nibble - 8 with the low nibbles first (elements
//       0..15) and the high nibbles after (16..31); result = hsum_float_8(acc).
// The fma chain runs over K/32 blocks per row (128 at K = 4096), so unlike the K-quants there is no short chain to
// replay: a lane owns one (row, AVX lane l) accumulator and walks the row; 8 rows x 8 lanes = one wave per 8-row tile.
//
// Layout LAYOUT_G4 (engine.cc:upload_matrix): per 8-row tile and group of 4 consecutive blocks one record, so that a
// lane's four dwords (its 4 elements of 4 blocks) are one 16-byte load and a wave reads 1 KiB contiguous:
//   Q8_0 (1088 B): qs[r][l][i] 4 B at (r*8 + l)*16 + i*4        | d[r][i] fp16 at 1024 + r*8 + i*2
//   Q4_0 ( 576 B): qs[r][l&3][i] 4 B at (r*4 + (l&3))*16 + i*4  | d[r][i] fp16 at  512 + r*8 + i*2
//       (lane l < 4 uses the low nibbles of its dword = elements 4l.., lane l >= 4 the high nibbles = elements 16+4(l-4)..)
// The activation vector is quantized once per workgroup into LDS in the matching [group][l][i] order.
#pragma once
#include "kernels_kq.h"

template <int MAXK> struct ActLdsQ32 {
    int q8[MAXK / 4];        // [g][l][i]: the 4 int8 of elements 4l..4l+3 of block 4g+i
    float yd[MAXK / 32];     // fp16-rounded block scales, as f32
    double red[16];
};

constexpr int kRecQ8_0 = 1088, kRecQ4_0 = 576;

// (RMSNorm ->) Q8_0 into LDS.  1024 threads: 8 lanes per 32-block (lane l holds elements 4l..4l+3), 128 blocks per pass.
// `after_requests`: called by every thread once the activations (and norm weights) are requested and before anything waits for
// them — the decode kernels put a workgroup barrier and their first weight requests there (the CU's memory pipeline serves its
// requests in order: an activation load queued behind other waves' weight records waits for them, kernels_v9.h).
struct NoHook { DEV void operator()() const {} };
template <int MAXK, class Hook = NoHook>
DEV void prologue_q8_0(ActLdsQ32<MAXK>& L, const float* __restrict__ x, const float* __restrict__ nw, int K, int pro, float eps,
                       const float* __restrict__ nbias = nullptr, Hook after_requests = Hook()) {
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6, l = tid & 7;
    const int nblk = K >> 5;
    constexpr int PASSES = (MAXK / 32 + 127) / 128;
    float4 v[PASSES], wn[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int b = (tid >> 3) + ps * 128;
        v[ps] = float4{0.f, 0.f, 0.f, 0.f};
        wn[ps] = float4{0.f, 0.f, 0.f, 0.f};
        if (b < nblk) {
            v[ps] = *(const float4*)(x + b * 32 + l * 4);
            if (pro != PRO_PLAIN) wn[ps] = *(const float4*)(nw + b * 32 + l * 4);
        }
    }
    after_requests();
    double s = 0.0;
    if (pro == PRO_RMSNORM) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            if ((tid >> 3) + ps * 128 < nblk) {
                s += (double)(v[ps].x * v[ps].x);
                s += (double)(v[ps].y * v[ps].y);
                s += (double)(v[ps].z * v[ps].z);
                s += (double)(v[ps].w * v[ps].w);
            }
        }
    }
    float scale = 1.0f;
    if (pro == PRO_RMSNORM) {   // ggml.c:10700-10716: double sum, f32 mean, 1/sqrtf
        s = wave_sum_fast(s);
        if (lane == 0) L.red[wv] = s;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += L.red[w];
        const float mean = (float)(tot / (double)K);
        scale = 1.0f / sqrtf(mean + eps);
    } else if (pro == PRO_LAYERNORM) {   // ggml.c:10605-10654, see prologue_q8k_exact16
        double s1 = 0.0;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps)
            if ((tid >> 3) + ps * 128 < nblk) { s1 += (double)v[ps].x; s1 += (double)v[ps].y; s1 += (double)v[ps].z; s1 += (double)v[ps].w; }
        s1 = wave_sum_fast(s1);
        if (lane == 0) L.red[wv] = s1;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += L.red[w];
        const float mean = (float)(tot / (double)K);
        __syncthreads();
        double s2 = 0.0;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            if ((tid >> 3) + ps * 128 < nblk) {
                v[ps].x -= mean; v[ps].y -= mean; v[ps].z -= mean; v[ps].w -= mean;
                s2 += (double)(v[ps].x * v[ps].x); s2 += (double)(v[ps].y * v[ps].y);
                s2 += (double)(v[ps].z * v[ps].z); s2 += (double)(v[ps].w * v[ps].w);
            }
        }
        s2 = wave_sum_fast(s2);
        if (lane == 0) L.red[wv] = s2;
        __syncthreads();
        double tot2 = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot2 += L.red[w];
        const float variance = (float)(tot2 / (double)K);
        scale = 1.0f / sqrtf(variance + eps);
    }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int b = (tid >> 3) + ps * 128;
        const bool live = b < nblk;
        float4 t = v[ps];
        if (live && pro != PRO_PLAIN) {
            const float4 w4 = wn[ps];
            t.x = (t.x * scale) * w4.x;
            t.y = (t.y * scale) * w4.y;
            t.z = (t.z * scale) * w4.z;
            t.w = (t.w * scale) * w4.w;
            if (pro == PRO_LAYERNORM) {
                const float4 b4 = *(const float4*)(nbias + b * 32 + l * 4);
                t.x += b4.x; t.y += b4.y; t.z += b4.z; t.w += b4.w;
            }
        }
        float amax = fmaxf(fmaxf(fabsf(t.x), fabsf(t.y)), fmaxf(fabsf(t.z), fabsf(t.w)));
        amax = fmaxf(amax, lane_xor1(amax));
        amax = fmaxf(amax, lane_xor2(amax));
        amax = fmaxf(amax, lane_xor4(amax));
        const float d = amax / 127.f;
        const float id = (amax != 0.0f) ? 127.f / amax : 0.0f;
        const int q0 = (int)__builtin_rintf(t.x * id), q1 = (int)__builtin_rintf(t.y * id);
        const int q2 = (int)__builtin_rintf(t.z * id), q3 = (int)__builtin_rintf(t.w * id);
        if (live) {
            L.q8[((b >> 2) * 8 + l) * 4 + (b & 3)] = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | ((q3 & 0xff) << 24);
            if (l == 0) L.yd[b] = f16_bits_to_f32(f32_to_f16_bits(d));
        }
    }
    __syncthreads();
}

// One 8-row tile: this lane's accumulator (row r = lane >> 3, AVX lane l = lane & 7) over all K/32 blocks.
template <int TYPE, int MAXK>
DEV float q32_tile_dot(const uint8_t* __restrict__ tile, int ng, const ActLdsQ32<MAXK>& L, int lane) {
    constexpr int REC = TYPE == GT_Q8_0 ? kRecQ8_0 : kRecQ4_0;
    constexpr int PF = 4;   // groups in flight per wave
    // Lane position p = lane & 7 of a row carries AVX lane l = bitrev3(p): hsum8_exact_dpp adds the partners at lane
    // distance 1, 2, 4 in that order, which must be the AVX lanes at distance 4, 2, 1 of hsum_float_8's tree.
    const int r = lane >> 3, p3 = lane & 7, l = ((p3 & 1) << 2) | (p3 & 2) | (p3 >> 2);
    const uint32_t qoff = TYPE == GT_Q8_0 ? (uint32_t)(r * 8 + l) * 16u : (uint32_t)(r * 4 + (l & 3)) * 16u;
    const uint32_t doff = (TYPE == GT_Q8_0 ? 1024u : 512u) + (uint32_t)r * 8u;
    const int sh = (TYPE == GT_Q4_0 && l >= 4) ? 4 : 0;
    float acc = 0.0f;
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
            {   // refill this slot for the next pass before consuming it
                const int gn = (g + PF < ng) ? g + PF : ng - 1;
                qv[u] = ld_stream16(tile + (size_t)gn * REC + qoff);
                dv[u] = *(const uint64_t*)(tile + (size_t)gn * REC + doff);
            }
            if (g < ng) {
                const u32x4 y = *(const u32x4*)&L.q8[(g * 8 + l) * 4];
                const float4 yd = *(const float4*)&L.yd[g * 4];
                const int ys[4] = {(int)y[0], (int)y[1], (int)y[2], (int)y[3]};
                const float yds[4] = {yd.x, yd.y, yd.z, yd.w};
                const uint32_t dw[4] = {(uint32_t)(dd & 0xFFFFu), (uint32_t)((dd >> 16) & 0xFFFFu),
                                        (uint32_t)((dd >> 32) & 0xFFFFu), (uint32_t)(dd >> 48)};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int sumi;
                    if constexpr (TYPE == GT_Q8_0) {
                        sumi = sdot4((int)q[i], ys[i], 0);
                    } else {
                        const int nib = (int)((q[i] >> sh) & 0x0F0F0F0Fu);
                        sumi = sdot4(nib, ys[i], 0) - 8 * sdot4(0x01010101, ys[i], 0);
                    }
                    const float d = f16_bits_to_f32((uint16_t)dw[i]) * yds[i];
                    acc = fmaf(d, (float)sumi, acc);
                }
            }
        }
    }
    return hsum8_exact_dpp(acc);
}

// grid = chip CUs, 1024 threads; wave gw = blockIdx*16 + wave takes items gw, gw + 16*gridDim, ...
// (item = 8-row tile of the concatenated jobs; gate/up launches: item = tile t of BOTH matrices).
template <int TYPE, int MAXK, bool GU>
__global__ void __launch_bounds__(1024) matvec_q32_kernel(const MatvecArgs a) {
    __shared__ ActLdsQ32<MAXK> L;
    const int lane = lane_id();
    const int wv = uniform_int(wave_id());
    constexpr int REC = TYPE == GT_Q8_0 ? kRecQ8_0 : kRecQ4_0;
    const int ng = a.K >> 7;
    const int pos = a.pos ? *a.pos : 0;
    prologue_q8_0<MAXK>(L, a.x, a.norm_w, a.K, a.pro, a.eps, a.norm_b);
    const int r = lane >> 3, l = lane & 7;
    const int stride = (int)gridDim.x * 16;
    for (int it = (int)blockIdx.x * 16 + wv; it < a.n_pairs; it += stride) {
        int j = 0;
        if (!GU) {
            if (a.njobs > 1 && it >= a.job[1].pair0) j = 1;
            if (a.njobs > 2 && it >= a.job[2].pair0) j = 2;
        }
        const int tile = it - (GU ? 0 : a.job[j].pair0);
        const DevMat& w = a.job[j].w;
        const float res = q32_tile_dot<TYPE, MAXK>(w.p[0] + (size_t)tile * ng * REC, ng, L, lane);
        const int row = tile * 8 + r;
        const bool own = l == 0 && row < w.M;
        if (GU) {
            const float up = q32_tile_dot<TYPE, MAXK>(a.job[1].w.p[0] + (size_t)tile * ng * REC, ng, L, lane);
            if (own) a.out[row] = f16_bits_to_f32(a.silu_tab[f32_to_f16_bits(res)]) * up;
            continue;
        }
        const int epi = a.job[j].epi;
        if (epi == EPI_ADD) {
            if (own) a.out[row] = res + a.res[row];
        } else if (epi == EPI_STORE) {
            if (own) a.out[row] = res;
        } else if (epi == EPI_V) {
            if (own) a.vcache[(size_t)row * a.v_stride + pos] = f32_to_f16_bits(res);
        } else if (epi == EPI_GELU) {
            if (own) a.out[row] = f16_bits_to_f32(a.gelu_tab[f32_to_f16_bits(res)]);
        } else if (epi == EPI_ADD2) {
            if (own) a.out[row] = (res + a.res[row]) + a.res2[row];
        } else if (epi == EPI_BIAS_STORE) {
            if (own) a.out[row] = a.bias[row] + res;
        } else if (epi == EPI_BIAS_ADD) {
            if (own) a.out[row] = (a.bias[row] + res) + a.res[row];
        } else if (epi == EPI_BIAS_GELU) {
            if (own) a.out[row] = f16_bits_to_f32(a.gelu_tab[f32_to_f16_bits(a.bias[row] + res)]);
        } else {
            const float other = lane_xor8(res);
            const int ip = (row % a.head_dim) >> 1;
            const float cs = a.rope_cs[((size_t)pos * (a.head_dim >> 1) + ip) * 2 + 0];
            const float sn = a.rope_cs[((size_t)pos * (a.head_dim >> 1) + ip) * 2 + 1];
            const float o = (r & 1) ? fmaf(res, cs, other * sn) : fmaf(res, cs, -(other * sn));
            if (own) {
                if (epi == EPI_ROPE_Q) a.q_f16[row] = f32_to_f16_bits(o);
                else a.kcache[kcache_off(pos, row, a.head_dim, a.n_ctx)] = f32_to_f16_bits(o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Systolic form (default): the K range of every tile is split over the NA <= 16 waves of the workgroup, each wave
// turns its block groups into (d_b, (float)sumi_b) register pairs — the order-free part, with all of the tile's loads
// in flight at once — and the row accumulators travel from wave to wave through an LDS mailbox: wave w waits for
// ctr[slot] == gen*NA + w, continues the reference's fma chain over its own blocks, passes the 64 accumulators on.
// Waves work on different tiles at the same time (wave 0 is NA-1 tiles ahead of the last wave), so after the first
// tile's NA hops the workgroup retires one tile per hop time, and the last wave runs hsum + epilogue.
// wave-per-tile (matvec_q32_kernel above) kept a single wave walking 128 dependent block steps behind a 4-deep
// prefetch: 18 us for Wo at 7B shapes, latency-bound (Q8_0 and Q4_0 took the same time).
// ------------------------------------------------------------------------------------------------------------------

constexpr int kQ32Slots = 8;
template <int MAXK> struct SmemQ32S {
    ActLdsQ32<MAXK> L;
    float mail[kQ32Slots][64];
    unsigned ctr[kQ32Slots];
};

template <int TYPE, int MAXK, int MAXG, bool GU>
__global__ void __launch_bounds__(1024) matvec_q32s_kernel(const MatvecArgs a) {
    __shared__ SmemQ32S<MAXK> SM;
    constexpr int REC = TYPE == GT_Q8_0 ? kRecQ8_0 : kRecQ4_0;
    const int lane = lane_id();
    const int wv = uniform_int(wave_id());
    const int ng = a.K >> 7;
    const int NA = ng < 16 ? ng : 16;                 // waves that own block groups
    if (threadIdx.x < kQ32Slots) SM.ctr[threadIdx.x] = 0u;   // published by the prologue's barrier
    const int base = ng / NA, rem = ng % NA;
    const int gcnt = base + (wv < rem ? 1 : 0);
    const int gbeg = wv * base + (wv < rem ? wv : rem);
    const int r = lane >> 3, p3 = lane & 7, l = ((p3 & 1) << 2) | (p3 & 2) | (p3 >> 2);   // AVX lane of this position
    const uint32_t qoff = TYPE == GT_Q8_0 ? (uint32_t)(r * 8 + l) * 16u : (uint32_t)(r * 4 + (l & 3)) * 16u;
    const uint32_t doff = (TYPE == GT_Q8_0 ? 1024u : 512u) + (uint32_t)r * 8u;
    const int sh = (TYPE == GT_Q4_0 && l >= 4) ? 4 : 0;
    // this workgroup's tile sequence: items blockIdx, blockIdx + gridDim, ...; gate/up launches visit (gate t, up t)
    const int stride = (int)gridDim.x, first = (int)blockIdx.x;
    const int n_loc = first < a.n_pairs ? (a.n_pairs - first + stride - 1) / stride : 0;
    const int n_seq = n_loc * (GU ? 2 : 1);
    auto tile_of = [&](int seq, int& j, int& tile) __attribute__((always_inline)) {
        const int it = first + (GU ? (seq >> 1) : seq) * stride;
        j = 0;
        if (GU) {
            j = seq & 1;
        } else {
            if (a.njobs > 1 && it >= a.job[1].pair0) j = 1;
            if (a.njobs > 2 && it >= a.job[2].pair0) j = 2;
        }
        tile = it - (GU ? 0 : a.job[j].pair0);
    };
    u32x4 qv[MAXG];
    uint64_t dv[MAXG];
    auto load_tile = [&](int seq) __attribute__((always_inline)) {
        int j, tile;
        tile_of(seq, j, tile);
        const uint8_t* tp = a.job[j].w.p[0] + ((size_t)tile * ng + gbeg) * REC;
#pragma unroll
        for (int u = 0; u < MAXG; ++u) {
            if (u < gcnt) {
                qv[u] = ld_stream16(tp + (size_t)u * REC + qoff);
                dv[u] = *(const uint64_t*)(tp + (size_t)u * REC + doff);
            }
        }
    };
    // the first tile's records are requested behind every wave's activation requests (barrier) and before the prologue waits
    prologue_q8_0<MAXK>(SM.L, a.x, a.norm_w, a.K, a.pro, a.eps, a.norm_b, [&]() __attribute__((always_inline)) {
        __syncthreads();
        if (wv < NA && n_seq > 0) load_tile(0);
    });
    if (wv >= NA) return;
    bool need_pos = false;
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) need_pos = need_pos || (jj < a.njobs && (a.job[jj].epi == EPI_ROPE_Q || a.job[jj].epi == EPI_ROPE_K || a.job[jj].epi == EPI_V));
    const int pos = (need_pos && a.pos) ? sload_i32(a.pos) : 0;
    float gate_res = 0.0f;
    for (int seq = 0; seq < n_seq; ++seq) {
        float dd[MAXG][4], ss[MAXG][4];
#pragma unroll
        for (int u = 0; u < MAXG; ++u) {
            if (u < gcnt) {
                const int g = gbeg + u;
                const u32x4 y = *(const u32x4*)&SM.L.q8[(g * 8 + l) * 4];
                const float4 yd = *(const float4*)&SM.L.yd[g * 4];
                const float yds[4] = {yd.x, yd.y, yd.z, yd.w};
                const uint32_t dw[4] = {(uint32_t)(dv[u] & 0xFFFFu), (uint32_t)((dv[u] >> 16) & 0xFFFFu),
                                        (uint32_t)((dv[u] >> 32) & 0xFFFFu), (uint32_t)(dv[u] >> 48)};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int sumi;
                    if constexpr (TYPE == GT_Q8_0) {
                        sumi = sdot4((int)qv[u][i], (int)y[i], 0);
                    } else {
                        const int nib = (int)((qv[u][i] >> sh) & 0x0F0F0F0Fu);
                        sumi = sdot4(nib, (int)y[i], 0) - 8 * sdot4(0x01010101, (int)y[i], 0);
                    }
                    dd[u][i] = f16_bits_to_f32((uint16_t)dw[i]) * yds[i];
                    ss[u][i] = (float)sumi;
                }
            }
        }
        if (seq + 1 < n_seq) load_tile(seq + 1);     // registers of this tile are consumed: request the next one now
        const int slot = seq % kQ32Slots;
        const unsigned gen_base = (unsigned)(seq / kQ32Slots) * (unsigned)NA;
        lds_wait_ge(&SM.ctr[slot], gen_base + (unsigned)wv);   // wave 0: the slot's previous tile was retired
        float acc = (wv == 0) ? 0.0f : SM.mail[slot][lane];
#pragma unroll
        for (int u = 0; u < MAXG; ++u) {
            if (u < gcnt) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = fmaf(dd[u][i], ss[u][i], acc);
            }
        }
        if (wv < NA - 1) {
            SM.mail[slot][lane] = acc;
            lds_signal(&SM.ctr[slot], lane, 1u);
            continue;
        }
        lds_signal(&SM.ctr[slot], lane, 1u);          // last wave: the slot is free again
        const float res = hsum8_exact_dpp(acc);
        int j, tile;
        tile_of(seq, j, tile);
        const int row = tile * 8 + r;
        const bool own = p3 == 0 && row < a.job[j].w.M;
        if (GU) {
            if (!(seq & 1)) { gate_res = res; continue; }
            if (own) a.out[row] = f16_bits_to_f32(a.silu_tab[f32_to_f16_bits(gate_res)]) * res;
            continue;
        }
        const int epi = a.job[j].epi;
        if (epi == EPI_ADD) {
            if (own) a.out[row] = res + a.res[row];
        } else if (epi == EPI_STORE) {
            if (own) a.out[row] = res;
        } else if (epi == EPI_V) {
            if (own) a.vcache[(size_t)row * a.v_stride + pos] = f32_to_f16_bits(res);
        } else if (epi == EPI_GELU) {
            if (own) a.out[row] = f16_bits_to_f32(a.gelu_tab[f32_to_f16_bits(res)]);
        } else if (epi == EPI_ADD2) {
            if (own) a.out[row] = (res + a.res[row]) + a.res2[row];
        } else if (epi == EPI_BIAS_STORE) {
            if (own) a.out[row] = a.bias[row] + res;
        } else if (epi == EPI_BIAS_ADD) {
            if (own) a.out[row] = (a.bias[row] + res) + a.res[row];
        } else if (epi == EPI_BIAS_GELU) {
            if (own) a.out[row] = f16_bits_to_f32(a.gelu_tab[f32_to_f16_bits(a.bias[row] + res)]);
        } else {
            const float other = lane_xor8(res);
            const int ip = (row % a.head_dim) >> 1;
            const float cs = a.rope_cs[((size_t)pos * (a.head_dim >> 1) + ip) * 2 + 0];
            const float sn = a.rope_cs[((size_t)pos * (a.head_dim >> 1) + ip) * 2 + 1];
            const float o = (r & 1) ? fmaf(res, cs, other * sn) : fmaf(res, cs, -(other * sn));
            if (own) {
                if (epi == EPI_ROPE_Q) a.q_f16[row] = f32_to_f16_bits(o);
                else a.kcache[kcache_off(pos, row, a.head_dim, a.n_ctx)] = f32_to_f16_bits(o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Wide rows (12288 < K <= 32768: the down projections of MPT-7B / StarCoder-7B and -15B, K = 4 d_model): the systolic
// form with a wave's share of the row (up to 16 block groups) taken in sub-batches of MAXG groups — the (d, sumi) pairs of
// the first sub-batch are ready before the mailbox wait as above, the later ones are formed while the wave holds the
// accumulators, each sub-batch's loads requested as soon as the previous one's registers are consumed.  Same chain,
// same order: block after block of the row.  Single-matrix and multi-job launches; no gate/up form (no such model).
// ------------------------------------------------------------------------------------------------------------------
template <int TYPE, int MAXK, int MAXG>
__global__ void __launch_bounds__(1024) matvec_q32w_kernel(const MatvecArgs a) {
    __shared__ SmemQ32S<MAXK> SM;
    constexpr int REC = TYPE == GT_Q8_0 ? kRecQ8_0 : kRecQ4_0;
    const int lane = lane_id();
    const int wv = uniform_int(wave_id());
    const int ng = a.K >> 7;
    const int NA = ng < 16 ? ng : 16;
    if (threadIdx.x < kQ32Slots) SM.ctr[threadIdx.x] = 0u;
    const int pos = a.pos ? *a.pos : 0;
    prologue_q8_0<MAXK>(SM.L, a.x, a.norm_w, a.K, a.pro, a.eps, a.norm_b);
    if (wv >= NA) return;
    const int base = ng / NA, rem = ng % NA;
    const int gcnt = base + (wv < rem ? 1 : 0);
    const int gbeg = wv * base + (wv < rem ? wv : rem);
    const int nsb = (gcnt + MAXG - 1) / MAXG;
    const int r = lane >> 3, p3 = lane & 7, l = ((p3 & 1) << 2) | (p3 & 2) | (p3 >> 2);
    const uint32_t qoff = TYPE == GT_Q8_0 ? (uint32_t)(r * 8 + l) * 16u : (uint32_t)(r * 4 + (l & 3)) * 16u;
    const uint32_t doff = (TYPE == GT_Q8_0 ? 1024u : 512u) + (uint32_t)r * 8u;
    const int sh = (TYPE == GT_Q4_0 && l >= 4) ? 4 : 0;
    const int stride = (int)gridDim.x, first = (int)blockIdx.x;
    const int n_seq = first < a.n_pairs ? (a.n_pairs - first + stride - 1) / stride : 0;
    auto tile_of = [&](int seq, int& j, int& tile) __attribute__((always_inline)) {
        const int it = first + seq * stride;
        j = 0;
        if (a.njobs > 1 && it >= a.job[1].pair0) j = 1;
        if (a.njobs > 2 && it >= a.job[2].pair0) j = 2;
        tile = it - a.job[j].pair0;
    };
    u32x4 qv[MAXG];
    uint64_t dv[MAXG];
    auto load_sub = [&](int seq, int sb) __attribute__((always_inline)) {
        int j, tile;
        tile_of(seq, j, tile);
        const uint8_t* tp = a.job[j].w.p[0] + ((size_t)tile * ng + gbeg + sb * MAXG) * REC;
#pragma unroll
        for (int u = 0; u < MAXG; ++u) {
            if (sb * MAXG + u < gcnt) {
                qv[u] = ld_stream16(tp + (size_t)u * REC + qoff);
                dv[u] = *(const uint64_t*)(tp + (size_t)u * REC + doff);
            }
        }
    };
    if (n_seq > 0) load_sub(0, 0);
    for (int seq = 0; seq < n_seq; ++seq) {
        const int slot = seq % kQ32Slots;
        const unsigned gen_base = (unsigned)(seq / kQ32Slots) * (unsigned)NA;
        float acc = 0.0f;
        for (int sb = 0; sb < nsb; ++sb) {
            float dd[MAXG][4], ss[MAXG][4];
#pragma unroll
            for (int u = 0; u < MAXG; ++u) {
                if (sb * MAXG + u < gcnt) {
                    const int g = gbeg + sb * MAXG + u;
                    const u32x4 y = *(const u32x4*)&SM.L.q8[(g * 8 + l) * 4];
                    const float4 yd = *(const float4*)&SM.L.yd[g * 4];
                    const float yds[4] = {yd.x, yd.y, yd.z, yd.w};
                    const uint32_t dw[4] = {(uint32_t)(dv[u] & 0xFFFFu), (uint32_t)((dv[u] >> 16) & 0xFFFFu),
                                            (uint32_t)((dv[u] >> 32) & 0xFFFFu), (uint32_t)(dv[u] >> 48)};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int sumi;
                        if constexpr (TYPE == GT_Q8_0) {
                            sumi = sdot4((int)qv[u][i], (int)y[i], 0);
                        } else {
                            const int nib = (int)((qv[u][i] >> sh) & 0x0F0F0F0Fu);
                            sumi = sdot4(nib, (int)y[i], 0) - 8 * sdot4(0x01010101, (int)y[i], 0);
                        }
                        dd[u][i] = f16_bits_to_f32((uint16_t)dw[i]) * yds[i];
                        ss[u][i] = (float)sumi;
                    }
                }
            }
            if (sb + 1 < nsb) load_sub(seq, sb + 1);
            else if (seq + 1 < n_seq) load_sub(seq + 1, 0);
            if (sb == 0) {
                lds_wait_ge(&SM.ctr[slot], gen_base + (unsigned)wv);
                acc = (wv == 0) ? 0.0f : SM.mail[slot][lane];
            }
#pragma unroll
            for (int u = 0; u < MAXG; ++u) {
                if (sb * MAXG + u < gcnt) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc = fmaf(dd[u][i], ss[u][i], acc);
                }
            }
        }
        if (wv < NA - 1) {
            SM.mail[slot][lane] = acc;
            lds_signal(&SM.ctr[slot], lane, 1u);
            continue;
        }
        lds_signal(&SM.ctr[slot], lane, 1u);
        const float res = hsum8_exact_dpp(acc);
        int j, tile;
        tile_of(seq, j, tile);
        const int row = tile * 8 + r;
        const bool own = p3 == 0 && row < a.job[j].w.M;
        const int epi = a.job[j].epi;
        if (epi == EPI_ADD) {
            if (own) a.out[row] = res + a.res[row];
        } else if (epi == EPI_STORE) {
            if (own) a.out[row] = res;
        } else if (epi == EPI_V) {
            if (own) a.vcache[(size_t)row * a.v_stride + pos] = f32_to_f16_bits(res);
        } else if (epi == EPI_GELU) {
            if (own) a.out[row] = f16_bits_to_f32(a.gelu_tab[f32_to_f16_bits(res)]);
        } else if (epi == EPI_ADD2) {
            if (own) a.out[row] = (res + a.res[row]) + a.res2[row];
        } else if (epi == EPI_BIAS_STORE) {
            if (own) a.out[row] = a.bias[row] + res;
        } else if (epi == EPI_BIAS_ADD) {
            if (own) a.out[row] = (a.bias[row] + res) + a.res[row];
        } else if (epi == EPI_BIAS_GELU) {
            if (own) a.out[row] = f16_bits_to_f32(a.gelu_tab[f32_to_f16_bits(a.bias[row] + res)]);
        }   // the rotary epilogues belong to the llama graph, whose K = n_embd rows never reach this kernel
    }
}
