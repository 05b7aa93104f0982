// Bit-exact mat-vec: reproduces the f32 accumulation ORDER of the reference's AVX2 kernels, not just their
// quantization points, so logits come out bit-identical to the reference CPU build and greedy decoding can never
// drift.  (Without this, a 1e-7 summation-order difference occasionally flips one int8 rounding in the next Q8_K
// activation quantization — ~10 flips per 7B token — and the logits wander at the 1e-4 level.)
//
// What the reference build computes per weight row (k_quants.c:2651-2720 Q4_K, :3183-3262 Q5_K, :3800-3872 Q6_K; all
// use 8 f32 lanes `acc` updated ONCE PER 256-BLOCK, in block order):
//     sumi[l] = sum over the block's 32-element sub-vectors s of  scale_s(l) * dot4(w_s[4l..4l+3], q8_s[4l..4l+3])   (int32)
//     acc[l]  = fma(y.d * fp16(x.d), (float)sumi[l], acc[l])                          l = 0..7
//   Q4_K min term: prod[t] = m[2t]*q8s[2t] + m[2t+1]*q8s[2t+1] (q8s[j] = bsums[2j]+bsums[2j+1]);
//                  acc_m[t] = fma(-y.d*fp16(x.dmin), (float)prod[t], acc_m[t])        t = 0..3
//   Q5_K min term: summs = fma(dmin, (float)(prod[0]+..+prod[3]), summs)              (scalar, gcc contracts it)
//   result = hsum_float_8(acc) [+ (acc_m0+acc_m2)+(acc_m1+acc_m3) | + summs],
//   hsum_float_8(x) = ((x0+x4)+(x2+x6)) + ((x1+x5)+(x3+x7))                           (k_quants.c:90-97)
//
// Mapping onto a wavefront: 8 lanes per weight row (8 rows per wave).  Lane g of a row owns the 16-byte unit g of each
// 128-byte nibble block, i.e. bytes 4k..4k+3 (k = 0..3) of AVX half h of two sub-vectors; the four lanes with the same
// h together cover all sub-vectors.  A 3-shuffle transpose-reduce over those four lanes leaves lane g holding the
// block's complete integer sumi[l(g)], l(g) = 4*(g&1) + 2*((g>>1)&1) + (g>>2), and the lane then performs exactly
// the reference's per-block fma on its own private accumulator.  Integer sums are exact in any order; every f32
// operation is the reference's, in the reference's order.
#pragma once
#include "kernels.h"

template <int MAXK> struct ActLdsX {
    int q8[MAXK / 4];        // int8 quants, 4 per word
    float yd[MAXK / 256];    // Q8_K block scale d
    int bsums[MAXK / 16];    // sums of 16 quants
    int sb[MAXK / 32];       // sums of 32 quants (= q8s[j] of the reference)
    double red[16];
};

// Prologue: (RMSNorm ->) Q8_K into LDS, values held in registers between the two phases (one global read of x).
// Reference k_quants.c:1191-1226 with `iscale*x[j] + 12582912.f` fused into one fma as the reference build does.
template <int NT, int MAXK>
DEV void prologue_q8k_exact(ActLdsX<MAXK>& L, const float* __restrict__ x, const float* __restrict__ nw, int K, int pro,
                            float eps, const float* __restrict__ nbias = nullptr) {
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    constexpr int NW = NT / 64;
    constexpr int MAXB = (MAXK / 256 + NW - 1) / NW;  // blocks per wave
    const int nblk = K >> 8;
    float4 v[MAXB];
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int b = wv + i * NW;
        if (b < nblk) {
            v[i] = *(const float4*)(x + b * 256 + lane * 4);
            if (pro == PRO_RMSNORM) {
                s += (double)(v[i].x * v[i].x);
                s += (double)(v[i].y * v[i].y);
                s += (double)(v[i].z * v[i].z);
                s += (double)(v[i].w * v[i].w);
            }
        }
    }
    float scale = 1.0f;
    if (pro == PRO_RMSNORM) {
        s = wave_sum(s);
        if (lane == 0) L.red[wv] = s;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < NW; ++w) tot += L.red[w];
        const float mean = (float)(tot / (double)K);
        scale = 1.0f / sqrtf(mean + eps);
    } else if (pro == PRO_LAYERNORM) {   // ggml.c:10605-10654 (see prologue_q8k_exact16)
        double s1 = 0.0;
#pragma unroll
        for (int i = 0; i < MAXB; ++i)
            if (wv + i * NW < nblk) { s1 += (double)v[i].x; s1 += (double)v[i].y; s1 += (double)v[i].z; s1 += (double)v[i].w; }
        s1 = wave_sum(s1);
        if (lane == 0) L.red[wv] = s1;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < NW; ++w) tot += L.red[w];
        const float mean = (float)(tot / (double)K);
        __syncthreads();
        double s2 = 0.0;
#pragma unroll
        for (int i = 0; i < MAXB; ++i) {
            if (wv + i * NW < nblk) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                s2 += (double)(v[i].x * v[i].x); s2 += (double)(v[i].y * v[i].y);
                s2 += (double)(v[i].z * v[i].z); s2 += (double)(v[i].w * v[i].w);
            }
        }
        s2 = wave_sum(s2);
        if (lane == 0) L.red[wv] = s2;
        __syncthreads();
        double tot2 = 0.0;
        for (int w = 0; w < NW; ++w) tot2 += L.red[w];
        const float variance = (float)(tot2 / (double)K);
        scale = 1.0f / sqrtf(variance + eps);
    }
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int b = wv + i * NW;
        if (b < nblk) {  // wave-uniform
            float4 t = v[i];
            if (pro != PRO_PLAIN) {
                const float4 w4 = *(const float4*)(nw + b * 256 + lane * 4);
                t.x = (t.x * scale) * w4.x;
                t.y = (t.y * scale) * w4.y;
                t.z = (t.z * scale) * w4.z;
                t.w = (t.w * scale) * w4.w;
                if (pro == PRO_LAYERNORM) {
                    const float4 b4 = *(const float4*)(nbias + b * 256 + lane * 4);
                    t.x += b4.x; t.y += b4.y; t.z += b4.z; t.w += b4.w;
                }
            }
            const float a0 = fabsf(t.x), a1 = fabsf(t.y), a2 = fabsf(t.z), a3 = fabsf(t.w);
            const float am = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
            const float amax = wave_max(am);
            const unsigned long long hit = __ballot(am == amax);
            const int first = __ffsll(hit) - 1;
            const float mine = (a0 == amax) ? t.x : (a1 == amax) ? t.y : (a2 == amax) ? t.z : t.w;
            const float maxv = __shfl(mine, first);
            int packed = 0, s4 = 0;
            float d = 0.0f;
            if (amax != 0.0f) {
                const float iscale = -128.f / maxv;
                int q0 = ((int)f32_to_bits(fmaf(iscale, t.x, 12582912.f)) & 0x007fffff) - 0x00400000;
                int q1 = ((int)f32_to_bits(fmaf(iscale, t.y, 12582912.f)) & 0x007fffff) - 0x00400000;
                int q2 = ((int)f32_to_bits(fmaf(iscale, t.z, 12582912.f)) & 0x007fffff) - 0x00400000;
                int q3 = ((int)f32_to_bits(fmaf(iscale, t.w, 12582912.f)) & 0x007fffff) - 0x00400000;
                q0 = q0 > 127 ? 127 : q0;
                q1 = q1 > 127 ? 127 : q1;
                q2 = q2 > 127 ? 127 : q2;
                q3 = q3 > 127 ? 127 : q3;
                packed = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | ((q3 & 0xff) << 24);
                s4 = q0 + q1 + q2 + q3;
                d = 1.0f / iscale;
            }
            L.q8[b * 64 + lane] = packed;
            s4 += __shfl_xor(s4, 1);
            s4 += __shfl_xor(s4, 2);
            if ((lane & 3) == 0) L.bsums[b * 16 + (lane >> 2)] = s4;
            s4 += __shfl_xor(s4, 4);
            if ((lane & 7) == 0) L.sb[b * 8 + (lane >> 3)] = s4;
            if (lane == 0) L.yd[b] = d;
        }
    }
    __syncthreads();
}

// Transpose-reduce over the four lanes {g, g^2, g^4, g^6} (c = g>>1): in: 4 ints per lane, out: lane c holds the
// 4-lane total of element kk(c) = 2*(c&1) + (c>>1).
DEV int quad_transpose_reduce(int p0, int p1, int p2, int p3, int c) {
    const bool odd = (c & 1) != 0;
    const int send0 = odd ? p0 : p2, send1 = odd ? p1 : p3;
    const int keep0 = odd ? p2 : p0, keep1 = odd ? p3 : p1;
    const int q0 = keep0 + __shfl_xor(send0, 2);
    const int q1 = keep1 + __shfl_xor(send1, 2);
    const bool up = (c & 2) != 0;
    const int send = up ? q0 : q1;
    const int keep = up ? q1 : q0;
    return keep + __shfl_xor(send, 4);
}

// hsum_float_8 of the 8 per-lane accumulators of one row, in the reference's association order (k_quants.c:90-97).
// Lane g holds x[l(g)], l(g) = 4*(g&1) + 2*((g>>1)&1) + (g>>2).
DEV float hsum8_exact(float acc) {
    const float t = acc + __shfl_xor(acc, 1);  // x[k] + x[k+4]
    const float u = t + __shfl_xor(t, 2);      // (r0+r2) or (r1+r3)
    return u + __shfl_xor(u, 4);               // (r0+r2) + (r1+r3)
}

struct TileResult { float v; };

// One 8-row tile (this lane: row r = lane>>3, unit g = lane&7) against the LDS-resident activation vector.
// Returns the finished dot product of row r in every lane of the row's group.
template <int MAXK, int UB>
DEV float tile_dot_exact(const DevMat& w, int tile, const ActLdsX<MAXK>& L, int lane) {
    const int nb = w.nb;
    const int r = lane >> 3, g = lane & 7, c = g >> 1, h = g & 1;
    const int rec = tile8_record_bytes(w.type);
    const uint8_t* base = w.p[0] + (size_t)tile * nb * rec;
    float acc = 0.0f, accm = 0.0f;
    if (w.type == GT_Q4_K || w.type == GT_Q5_K) {
        const bool q5 = w.type == GT_Q5_K;
        const int qh_off = 128, qs_off = q5 ? 128 + 256 : 128;
        for (int b0 = 0; b0 < nb; b0 += UB) {
            u32x4 qs[UB], hd[UB], qh[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int bb = (b0 + u < nb) ? b0 + u : nb - 1;
                const uint8_t* rp = base + (size_t)bb * rec;
                hd[u] = ld_stream16(rp + r * 16);
                qs[u] = ld_stream16(rp + qs_off + r * 128 + g * 16);
                if (q5) qh[u] = ld_stream16(rp + qh_off + r * 32 + h * 16);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int b = b0 + u;
                if (b < nb) {  // wave-uniform
                    const int* alo = &L.q8[b * 64 + 16 * c + 4 * h];
                    const int* ahi = alo + 8;
                    int sc_lo, sc_hi, m_lo, m_hi;
                    scale_min_pair(hd[u][1], hd[u][2], hd[u][3], c, sc_lo, sc_hi, m_lo, m_hi);
                    int part[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        uint32_t lo = qs[u][k] & 0x0F0F0F0Fu;
                        uint32_t hi = (qs[u][k] >> 4) & 0x0F0F0F0Fu;
                        if (q5) {
                            lo |= ((qh[u][k] >> (2 * c)) & 0x01010101u) << 4;
                            hi |= ((qh[u][k] >> (2 * c + 1)) & 0x01010101u) << 4;
                        }
                        part[k] = sc_lo * sdot4((int)lo, alo[k], 0) + sc_hi * sdot4((int)hi, ahi[k], 0);
                    }
                    const int sumi = quad_transpose_reduce(part[0], part[1], part[2], part[3], c);
                    const float yd = L.yd[b];
                    const float d = yd * f16_bits_to_f32((uint16_t)(hd[u][0] & 0xFFFF));
                    const float dmin = -yd * f16_bits_to_f32((uint16_t)(hd[u][0] >> 16));
                    acc = fmaf(d, (float)sumi, acc);
                    // min term: this lane's chunk c owns sub-blocks 2c, 2c+1 => prod[c]; only the h == 0 lanes carry it
                    int prod = (h == 0) ? m_lo * L.sb[b * 8 + 2 * c] + m_hi * L.sb[b * 8 + 2 * c + 1] : 0;
                    if (q5) {  // scalar summs: all four prods are added as integers first
                        prod += __shfl_xor(prod, 2);
                        prod += __shfl_xor(prod, 4);
                    }
                    accm = fmaf(dmin, (float)prod, accm);
                }
            }
        }
        float tot = hsum8_exact(acc);
        if (!q5) {
            const float wsum = accm + __shfl_xor(accm, 4);   // (m0+m2) | (m1+m3)   [t = c, lanes with h == 0]
            accm = wsum + __shfl_xor(wsum, 2);               // (m0+m2) + (m1+m3)
        }
        accm = __shfl(accm, lane & ~7);                      // lane g = 0 of the row (h == 0, c == 0)
        tot = tot + accm;
        return tot;
    }
    // ---- GT_Q6_K ----
    {
        const int n = g >> 2, gg = g & 3, hh = gg & 1, kq = gg >> 1;
        const int s_lo = 2 * kq, s_hi = 4 + 2 * kq;
        for (int b0 = 0; b0 < nb; b0 += UB) {
            u32x4 ql[UB], qh[UB], sc[UB];
            uint16_t dd[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int bb = (b0 + u < nb) ? b0 + u : nb - 1;
                const uint8_t* rp = base + (size_t)bb * rec;
                dd[u] = *(const uint16_t*)(rp + r * 2);
                sc[u] = ld_stream16(rp + 16 + r * 16);
                qh[u] = ld_stream16(rp + 144 + r * 64 + n * 32 + hh * 16);
                ql[u] = ld_stream16(rp + 656 + r * 128 + g * 16);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int b = b0 + u;
                if (b < nb) {
                    const int* alo = &L.q8[b * 64 + 32 * n + 4 * gg];
                    const int* ahi = alo + 16;
                    const uint32_t w_lo = n ? sc[u][2] : sc[u][0];
                    const uint32_t w_hi = n ? sc[u][3] : sc[u][1];
                    const int sc_lo = (int)(int8_t)((w_lo >> (8 * gg)) & 0xFF);
                    const int sc_hi = (int)(int8_t)((w_hi >> (8 * gg)) & 0xFF);
                    int part[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t lo = (ql[u][k] & 0x0F0F0F0Fu) | (((qh[u][k] >> s_lo) & 0x03030303u) << 4);
                        const uint32_t hi = ((ql[u][k] >> 4) & 0x0F0F0F0Fu) | (((qh[u][k] >> s_hi) & 0x03030303u) << 4);
                        const int dl = sdot4((int)lo, alo[k], 0) - 32 * sdot4(0x01010101, alo[k], 0);
                        const int dh = sdot4((int)hi, ahi[k], 0) - 32 * sdot4(0x01010101, ahi[k], 0);
                        part[k] = sc_lo * dl + sc_hi * dh;
                    }
                    const int sumi = quad_transpose_reduce(part[0], part[1], part[2], part[3], c);
                    const float d = L.yd[b] * f16_bits_to_f32(dd[u]);
                    acc = fmaf(d, (float)sumi, acc);
                }
            }
        }
        return hsum8_exact(acc);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Fused launch: prologue -> one 8-row tile per wave step -> epilogue.  Work item = tile index over the concatenated
// jobs (gate/up mode: item t = gate tile t followed by up tile t).
// ------------------------------------------------------------------------------------------------------------------
template <int NT, int MAXK, int UB>
__global__ void __launch_bounds__(NT) matvec_exact_kernel(const MatvecArgs a) {
    __shared__ ActLdsX<MAXK> L;
    const int lane = lane_id();
    prologue_q8k_exact<NT, MAXK>(L, a.x, a.norm_w, a.K, a.pro, a.eps, a.norm_b);
    constexpr int NW = NT / 64;
    const int gw = (int)blockIdx.x * NW + wave_id();
    const int W = (int)gridDim.x * NW;
    const int i_begin = (int)(((long long)a.n_pairs * gw) / W);  // n_pairs == number of work items (tiles) here
    const int i_end = (int)(((long long)a.n_pairs * (gw + 1)) / W);
    const int pos = a.pos ? *a.pos : 0;
    const int r = lane >> 3, g = lane & 7;
    for (int it = i_begin; it < i_end; ++it) {
        int j = 0;
        if (!a.gateup) {
            if (a.njobs > 1 && it >= a.job[1].pair0) j = 1;
            if (a.njobs > 2 && it >= a.job[2].pair0) j = 2;
        }
        const MatJob& jb = a.job[j];
        const int tile = it - jb.pair0;
        const int row = tile * 8 + r;
        float res = tile_dot_exact<MAXK, UB>(jb.w, tile, L, lane);
        const int epi = a.gateup ? EPI_SILU_MUL : jb.epi;
        if (epi == EPI_SILU_MUL) {
            const float up = tile_dot_exact<MAXK, UB>(a.job[1].w, tile, L, lane);
            if (g == 0 && row < jb.w.M) a.out[row] = f16_bits_to_f32(a.silu_tab[f32_to_f16_bits(res)]) * up;
        } else if (epi == EPI_STORE) {
            if (g == 0 && row < jb.w.M) a.out[row] = res;
        } else if (epi == EPI_ADD) {
            if (g == 0 && row < jb.w.M) a.out[row] = res + a.res[row];
        } else if (epi == EPI_V) {
            if (g == 0 && row < jb.w.M) a.vcache[(size_t)row * a.v_stride + pos] = f32_to_f16_bits(res);
        } else if (epi == EPI_GELU) {
            if (g == 0 && row < jb.w.M) a.out[row] = f16_bits_to_f32(a.gelu_tab[f32_to_f16_bits(res)]);
        } else if (epi == EPI_ADD2) {
            if (g == 0 && row < jb.w.M) a.out[row] = (res + a.res[row]) + a.res2[row];
        } else {  // RoPE on the interleaved pair (row&~1, row|1): partner row lives in the neighbouring lane group
            const float other = __shfl_xor(res, 8);
            const int ip = (row % a.head_dim) >> 1;
            const float cs = a.rope_cs[((size_t)pos * (a.head_dim >> 1) + ip) * 2 + 0];
            const float sn = a.rope_cs[((size_t)pos * (a.head_dim >> 1) + ip) * 2 + 1];
            // reference build: out0 = fma(x0, cos, -(x1*sin)), out1 = fma(x1, cos, x0*sin)  (how gcc contracts
            // ggml.c:12536-12537; established against the reference's own rope op, see oracle/mirror.c mir_rope)
            const float o = (r & 1) ? fmaf(res, cs, other * sn) : fmaf(res, cs, -(other * sn));
            if (g == 0 && row < jb.w.M) {
                if (epi == EPI_ROPE_Q) a.q_f16[row] = f32_to_f16_bits(o);
                else a.kcache[kcache_off(pos, row, a.head_dim, a.n_ctx)] = f32_to_f16_bits(o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Bit-exact decode attention.  The reference computes both attention mat-muls with ggml_vec_dot_f16 (ggml.c:2392-2425,
// AVX: 4 accumulators x 8 f32 lanes, 32 elements per step, fma; reduce macro ggml.c:1964-1982):
//     s_j[l] accumulates elements e = 32*step + 8*j + l, in step order          (j = 0..3, l = 0..7)
//     S[l]   = (s_0[l] + s_2[l]) + (s_1[l] + s_3[l])
//     t0[m]  = S[m] + S[m+4]  (m = 0..3);   res = (t0[0] + t0[1]) + (t0[2] + t0[3])
//     leftovers (n % 32): sumf = (double)res; sumf += (double)(x[i]*y[i]) sequentially;  result = (float)sumf
// A quad of lanes (j = lane&3) owns the four accumulator vectors: lane j reads the 16-byte chunks j, j+4, j+8, ... of
// the row (8 halves = its 8 l-lanes for one step) and runs 8 independent fma chains; a 6-shuffle transpose-reduce
// reproduces the reduction tree.
// ------------------------------------------------------------------------------------------------------------------
DEV void unpack8_f16(const u32x4 v, float* f) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[2 * k] = f16_bits_to_f32((uint16_t)(v[k] & 0xFFFF));
        f[2 * k + 1] = f16_bits_to_f32((uint16_t)(v[k] >> 16));
    }
}

// in: acc[8] = s_j[0..7] of quad lane j; out (all 4 lanes): res of the reference's reduce.
DEV float f16dot_reduce_exact(const float* acc, int j) {
    // step A (partner j^2): keep l in {0,1,4,5} (j<2) or {2,3,6,7} (j>=2)
    const bool hiA = (j & 2) != 0;
    float kA[4], sA[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int l_lo = (t & 1) + 4 * (t >> 1);        // 0,1,4,5
        const int l_hi = l_lo + 2;                      // 2,3,6,7
        kA[t] = hiA ? acc[l_hi] : acc[l_lo];
        sA[t] = hiA ? acc[l_lo] : acc[l_hi];
    }
    float a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = kA[t] + __shfl_xor(sA[t], 2);   // s_j[l] + s_{j^2}[l], l = base(t) (+2 if j>=2)
    // a[] now holds l = {0,1,4,5} (+2 when j>=2).  step B (partner j^1): even j keeps t = {0,2} (l', l'+4 with l' even
    // offset 0), odd j keeps t = {1,3}.
    const bool odd = (j & 1) != 0;
    const float k0 = odd ? a[1] : a[0], k1 = odd ? a[3] : a[2];
    const float s0 = odd ? a[0] : a[1], s1 = odd ? a[2] : a[3];
    const float S_lo = k0 + __shfl_xor(s0, 1);   // S[l'],   l' = j
    const float S_hi = k1 + __shfl_xor(s1, 1);   // S[l'+4]
    const float t0 = S_lo + S_hi;                // t0[j]
    const float t1 = t0 + __shfl_xor(t0, 1);     // t0[0]+t0[1]  |  t0[2]+t0[3]
    return t1 + __shfl_xor(t1, 2);
}

struct AttnArgsX {
    const uint16_t* q_f16;
    const uint16_t* kcache;  // layer base [n_head_kv][n_ctx][head_dim] (kcache_off)
    const uint16_t* vcache;  // layer base [n_embd_gqa][v_stride]
    float* scores;           // [n_head][n_ctx]
    float* out;
    const int* pos;
    const int* n_total;      // device scalar: n_past + N of the batch this token belongs to (see attn_softmax_pv_exact_kernel)
    const uint16_t* exp_tab;
    int n_head, n_head_kv, head_dim, n_embd_gqa, n_ctx, v_stride;
    float kq_scale;
    unsigned long long* trace;   // measurement only: s_memtime stamps of workgroup (0,0)
};

// scores[h][p] = vec_dot_f16(K[p], Q[h]) * kq_scale.   grid (n_head, ceil(n_ctx/64)), 256 threads: quad per position.
__global__ void __launch_bounds__(256) attn_scores_exact_kernel(const AttnArgsX a) {
    const int h = (int)blockIdx.x;
    const int n_kv = *a.pos + 1;
    const int c0 = (int)blockIdx.y * 64;
    if (c0 >= n_kv) return;
    const int tid = (int)threadIdx.x, j = tid & 3;
    const int p = c0 + (tid >> 2);
    const bool ok = p < n_kv;
    const int pp = ok ? p : c0;
    const int hd = a.head_dim;
    const int hk = h / (a.n_head / a.n_head_kv);
    const uint16_t* krow = a.kcache + ((size_t)hk * a.n_ctx + pp) * hd;
    const uint16_t* qrow = a.q_f16 + (size_t)h * hd;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e0 = 0; e0 < hd; e0 += 32) {
        float kf[8], qf[8];
        unpack8_f16(ld16(krow + e0 + 8 * j), kf);
        unpack8_f16(ld16(qrow + e0 + 8 * j), qf);
#pragma unroll
        for (int l = 0; l < 8; ++l) acc[l] = fmaf(kf[l], qf[l], acc[l]);
    }
    const float res = f16dot_reduce_exact(acc, j);
    if (ok && j == 0) a.scores[(size_t)h * a.n_ctx + p] = res * a.kq_scale;
}

// softmax (reference ggml.c:12047-12069) + out[h][d] = vec_dot_f16(V[d], P).   grid (n_head, head_dim/64), 256 threads.
// Batch structure matters for bit-identity: the reference evaluates a chunk of N tokens as one graph, so every query row
// of the chunk runs vec_dot_f16 over ALL n_total = n_past + N columns (masked columns hold P = 0): the split between the
// 32-wide fma part and the scalar double-precision leftovers is taken at n_total & ~31, not at this token's own length.
// fma(v, 0, acc) == acc, so only the split point has to be reproduced.
__global__ void __launch_bounds__(256) attn_softmax_pv_exact_kernel(const AttnArgsX a) {
    __shared__ float prob[kMaxCtx];
    __shared__ double red[4];
    __shared__ float redf[4];
    const int h = (int)blockIdx.x;
    const int n_kv = *a.pos + 1;
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = wave_id();
    const float* s = a.scores + (size_t)h * a.n_ctx;
    float mx = -INFINITY;
    for (int i = tid; i < n_kv; i += 256) mx = fmaxf(mx, s[i]);
    mx = wave_max(mx);
    if (lane == 0) redf[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    double sum = 0.0;
    for (int i = tid; i < n_kv; i += 256) {
        const float e = f16_bits_to_f32(a.exp_tab[f32_to_f16_bits(s[i] - mx)]);
        prob[i] = e;
        sum += (double)e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    const double tot = ((red[0] + red[1]) + red[2]) + red[3];
    const float inv = (float)(1.0 / tot);
    for (int i = tid; i < n_kv; i += 256) prob[i] = f16_bits_to_f32(f32_to_f16_bits(prob[i] * inv));
    const int n_tot = *a.n_total;
    const int np = n_tot & ~31;
    for (int i = n_kv + tid; i < np; i += 256) prob[i] = 0.0f;  // masked columns of this batch
    __syncthreads();
    const int j = tid & 3;
    const int d = (int)blockIdx.y * 64 + (tid >> 2);
    const int hk = h / (a.n_head / a.n_head_kv);
    const uint16_t* vrow = a.vcache + ((size_t)hk * a.head_dim + d) * a.v_stride;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < np; i += 32) {
        float vf[8];
        unpack8_f16(ld16(vrow + i + 8 * j), vf);
        const float* pr = &prob[i + 8 * j];
#pragma unroll
        for (int l = 0; l < 8; ++l) acc[l] = fmaf(vf[l], pr[l], acc[l]);
    }
    const float res = f16dot_reduce_exact(acc, j);
    double sumf = (double)res;
    for (int i = np; i < n_kv; ++i) sumf += (double)(f16_bits_to_f32(vrow[i]) * prob[i]);
    if (j == 0) a.out[(size_t)h * a.head_dim + d] = (float)sumf;
}

// Fused form of the two kernels above (one launch per layer instead of two): grid (n_head, head_dim/64), 1024 threads.
// Every workgroup recomputes the (cheap) score row of its head into LDS — 256 positions per pass, a quad per position —
// then runs the softmax and its 64 channels of V*P exactly as attn_softmax_pv_exact_kernel does.  The double-precision
// exp sum is order-free here: the addends are fp16 values in (0, 1] (multiples of 2^-24), so any summation order of up
// to 8192 of them is exact in binary64.
template <int NT, int HD>
__global__ void __launch_bounds__(NT) attn_fused_exact_kernel(const AttnArgsX a) {
    constexpr int NWV = NT / 64, NQ = NT / 4;   // NQ quads: positions per pass
    constexpr int NC = HD / 32;                 // 16-byte chunks of a K row per quad lane
    constexpr int PB = 4;                       // positions per quad whose K rows are in flight together
    constexpr int VB = 8;                       // V chunks (32 positions each) in flight together
    __shared__ float prob[kMaxCtx];
    __shared__ double red[NWV];
    __shared__ float redf[NWV];
    const int h = (int)blockIdx.x;
    const bool trace = a.trace && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0;
    unsigned long long* tr = a.trace + 16 * (threadIdx.x >> 6);
    if (trace) tr[0] = clock64_dev();
    const int n_kv = *a.pos + 1;
    const int n_tot = *a.n_total;
    const int np = n_tot & ~31;
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = wave_id(), j = tid & 3;
    const int hk = h / (a.n_head / a.n_head_kv);
    if (trace) { tr[1] = clock64_dev(); tr[7] = (unsigned long long)n_kv; }
    const uint16_t* qrow = a.q_f16 + (size_t)h * HD;
    const uint16_t* kbase = a.kcache + (size_t)hk * a.n_ctx * HD + 8 * j;
    const bool pv_thread = tid < 256;           // 64 channels x 4 lanes run the V*P part
    const int d = (int)blockIdx.y * 64 + ((tid & 255) >> 2);
    const uint16_t* vrow = a.vcache + ((size_t)hk * HD + d) * a.v_stride;
    // The V rows do not depend on the probabilities: their first VB chunks are requested now, so that their latency
    // overlaps the score and softmax phases instead of following them.
    u32x4 vv[VB];
    u32x4 qv[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) qv[c] = ld16(qrow + 32 * c + 8 * j);
    float mx = -INFINITY;
    for (int base = 0; base < n_kv; base += NQ * PB) {
        u32x4 kv[PB][NC];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int p = base + u * NQ + (tid >> 2);
            const uint16_t* krow = kbase + (size_t)p * HD;
#pragma unroll
            for (int c = 0; c < NC; ++c)   // no clamping: hundreds of idle quads re-reading one row serialise in the L1
                kv[u][c] = (p < n_kv) ? ld16(krow + 32 * c) : u32x4{0u, 0u, 0u, 0u};
        }
        if (base == 0) {   // after the K requests (the critical path), before anything waits on them
#pragma unroll
            for (int u = 0; u < VB; ++u) vv[u] = (pv_thread && 32 * u < np) ? ld16(vrow + 32 * u + 8 * j) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int p = base + u * NQ + (tid >> 2);
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float kf[8], qf[8];
                unpack8_f16(kv[u][c], kf);
                unpack8_f16(qv[c], qf);
#pragma unroll
                for (int l = 0; l < 8; ++l) acc[l] = fmaf(kf[l], qf[l], acc[l]);
            }
            const float sc = f16dot_reduce_exact(acc, j) * a.kq_scale;
            if (p < n_kv) {
                mx = fmaxf(mx, sc);
                if (j == 0) prob[p] = sc;
            }
        }
    }
    if (trace) tr[2] = clock64_dev();
    mx = wave_max(mx);
    if (lane == 0) redf[wv] = mx;
    __syncthreads();
    mx = redf[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) mx = fmaxf(mx, redf[w]);
    if (trace) tr[3] = clock64_dev();
    double sum = 0.0;
    for (int i = tid; i < n_kv; i += NT) {
        const float e = f16_bits_to_f32(a.exp_tab[f32_to_f16_bits(prob[i] - mx)]);
        prob[i] = e;
        sum += (double)e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    double tot = red[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) tot += red[w];
    const float inv = (float)(1.0 / tot);
    for (int i = tid; i < n_kv; i += NT) prob[i] = f16_bits_to_f32(f32_to_f16_bits(prob[i] * inv));
    for (int i = n_kv + tid; i < np; i += NT) prob[i] = 0.0f;  // masked columns of this batch
    __syncthreads();
    if (trace) tr[4] = clock64_dev();
    if (!pv_thread) return;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i0 = 0; i0 < np; i0 += 32 * VB) {
        if (i0 > 0) {
#pragma unroll
            for (int u = 0; u < VB; ++u)
                if (i0 + 32 * u < np) vv[u] = ld16(vrow + i0 + 32 * u + 8 * j);
        }
#pragma unroll
        for (int u = 0; u < VB; ++u) {
            if (i0 + 32 * u < np) {
                float vf[8];
                unpack8_f16(vv[u], vf);
                const float* pr = &prob[i0 + 32 * u + 8 * j];
#pragma unroll
                for (int l = 0; l < 8; ++l) acc[l] = fmaf(vf[l], pr[l], acc[l]);
            }
        }
    }
    const float res = f16dot_reduce_exact(acc, j);
    double sumf = (double)res;
    if (trace) tr[5] = clock64_dev();
    for (int i = np; i < n_kv; ++i) sumf += (double)(f16_bits_to_f32(vrow[i]) * prob[i]);
    if (j == 0) a.out[(size_t)h * HD + d] = (float)sumf;
    if (trace) tr[6] = clock64_dev();
}

// ------------------------------------------------------------------------------------------------------------------
// Design C: the same bit-exact arithmetic with K split across the waves of a workgroup.
// The integer part of a block (loads, nibble unpack, dot4, transpose-reduce) is order-free, so the NW waves of a
// workgroup take the blocks b = wave, wave+NW, ... of ONE 8-row tile concurrently and leave per block
//     S[b][lane] = (float)sumi[l(g)]   D[b][row] = y.d*fp16(x.d)   DM[b][row] = -y.d*fp16(x.dmin)   PM[b][row][t]
// in LDS; after one barrier a single wave replays the reference's sequential f32 fma chain over b = 0..nb-1 from LDS
// (nb short dependent steps) and runs the epilogue.  This multiplies the loads in flight per tile by NW and cuts the
// serial depth per tile from nb block-steps to nb/NW, which is what the small 4096-row matrices (Wo, W_down: 512
// tiles) need to keep every CU streaming.
// ------------------------------------------------------------------------------------------------------------------
template <int MAXNB> struct ChainBuf {
    float S[MAXNB][64];
    float D[MAXNB][8];
    float DM[MAXNB][8];
    float PM[MAXNB][32];
};

// Integer work of one block for this lane's row; results go to the chain buffer.
template <int MAXK, int MAXNB>
DEV void block_to_chain(int type, const uint8_t* rp, int b, const ActLdsX<MAXK>& L, ChainBuf<MAXNB>& C, int lane,
                        const u32x4 v0, const u32x4 v1, const u32x4 v2, const uint16_t dd) {
    const int r = lane >> 3, g = lane & 7, c = g >> 1, h = g & 1;
    (void)rp;
    if (type == GT_Q4_K || type == GT_Q5_K) {
        const bool q5 = type == GT_Q5_K;
        // v0 = hdr, v1 = qs, v2 = qh (Q5_K)
        const int* alo = &L.q8[b * 64 + 16 * c + 4 * h];
        const int* ahi = alo + 8;
        int sc_lo, sc_hi, m_lo, m_hi;
        scale_min_pair(v0[1], v0[2], v0[3], c, sc_lo, sc_hi, m_lo, m_hi);
        int part[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t lo = v1[k] & 0x0F0F0F0Fu;
            uint32_t hi = (v1[k] >> 4) & 0x0F0F0F0Fu;
            if (q5) {
                lo |= ((v2[k] >> (2 * c)) & 0x01010101u) << 4;
                hi |= ((v2[k] >> (2 * c + 1)) & 0x01010101u) << 4;
            }
            part[k] = sc_lo * sdot4((int)lo, alo[k], 0) + sc_hi * sdot4((int)hi, ahi[k], 0);
        }
        const int sumi = quad_transpose_reduce(part[0], part[1], part[2], part[3], c);
        C.S[b][lane] = (float)sumi;
        int prod = (h == 0) ? m_lo * L.sb[b * 8 + 2 * c] + m_hi * L.sb[b * 8 + 2 * c + 1] : 0;
        if (q5) {
            prod += __shfl_xor(prod, 2);
            prod += __shfl_xor(prod, 4);
        }
        if (h == 0) C.PM[b][r * 4 + c] = (float)prod;
        if (g == 0) {
            const float yd = L.yd[b];
            C.D[b][r] = yd * f16_bits_to_f32((uint16_t)(v0[0] & 0xFFFF));
            C.DM[b][r] = -yd * f16_bits_to_f32((uint16_t)(v0[0] >> 16));
        }
    } else {  // GT_Q6_K: v0 = sc, v1 = ql, v2 = qh, dd = d
        const int n = g >> 2, gg = g & 3, kq = gg >> 1;
        const int s_lo = 2 * kq, s_hi = 4 + 2 * kq;
        const int* alo = &L.q8[b * 64 + 32 * n + 4 * gg];
        const int* ahi = alo + 16;
        const uint32_t w_lo = n ? v0[2] : v0[0];
        const uint32_t w_hi = n ? v0[3] : v0[1];
        const int sc_lo = (int)(int8_t)((w_lo >> (8 * gg)) & 0xFF);
        const int sc_hi = (int)(int8_t)((w_hi >> (8 * gg)) & 0xFF);
        int part[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = (v1[k] & 0x0F0F0F0Fu) | (((v2[k] >> s_lo) & 0x03030303u) << 4);
            const uint32_t hi = ((v1[k] >> 4) & 0x0F0F0F0Fu) | (((v2[k] >> s_hi) & 0x03030303u) << 4);
            const int dl = sdot4((int)lo, alo[k], 0) - 32 * sdot4(0x01010101, alo[k], 0);
            const int dh = sdot4((int)hi, ahi[k], 0) - 32 * sdot4(0x01010101, ahi[k], 0);
            part[k] = sc_lo * dl + sc_hi * dh;
        }
        const int sumi = quad_transpose_reduce(part[0], part[1], part[2], part[3], c);
        C.S[b][lane] = (float)sumi;
        if (g == 0) C.D[b][r] = L.yd[b] * f16_bits_to_f32(dd);
    }
}

// Prologue, 16 lanes per 256-block: lane `sub` owns 16 consecutive elements, so the per-block reductions are 4 DPP
// steps inside a row of 16 lanes and all 16 (32) blocks of a round proceed at once.  Same arithmetic as
// prologue_q8k_exact (reference k_quants.c:1191-1226 with the fused fma; RMSNorm ggml.c:10700-10716).
// Two copies on purpose: the LayerNorm form (falcon) needs more live registers; keeping it out of the RMSNorm / plain
// function lets the register allocator treat the two call sites separately (with one merged body every mat-vec
// instantiation started to spill and the Q6_K K=11008 kernel went from 16 to 27 us).
template <int NT, int MAXK>
DEV void prologue_q8k_exact16(ActLdsX<MAXK>& L, const float* __restrict__ x, const float* __restrict__ nw, int K, int pro,
                              float eps) {
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, sub = tid & 15, grp = tid >> 4;
    constexpr int NW = NT / 64, NG = NT / 16;
    constexpr int ROUNDS = (MAXK / 256 + NG - 1) / NG;
    const int nblk = K >> 8;
    // A wave whose four 16-lane rows are all past the last block has nothing to quantize: it must SKIP the arithmetic
    // (wave-uniform branches), not run it predicated off — the prologue is VALU-issue bound (about 250 wave
    // instructions), and for K = 4096 only 4 of the 16 waves (one per SIMD) are live.
    const bool wave_live = uniform_int(wv * 4) < nblk;
    float4 v[ROUNDS][4];
    double s = 0.0;
    if (wave_live) {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int b = grp + rd * NG;
            if (b < nblk) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[rd][k] = *(const float4*)(x + b * 256 + sub * 16 + k * 4);
                    if (pro == PRO_RMSNORM) {
                        s += (double)(v[rd][k].x * v[rd][k].x);
                        s += (double)(v[rd][k].y * v[rd][k].y);
                        s += (double)(v[rd][k].z * v[rd][k].z);
                        s += (double)(v[rd][k].w * v[rd][k].w);
                    }
                }
            }
        }
    }
    float scale = 1.0f;
    if (pro == PRO_RMSNORM) {
        if (wave_live) {
            s = wave_sum_fast(s);
            if (lane == 0) L.red[wv] = s;
        } else if (lane == 0) {
            L.red[wv] = 0.0;
        }
        __syncthreads();
        if (wave_live) {
            double tot = 0.0;
            for (int w = 0; w < NW; ++w) tot += L.red[w];
            const float mean = (float)(tot / (double)K);
            scale = 1.0f / sqrtf(mean + eps);
        }
    }
    if (wave_live) {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int b = grp + rd * NG;
            const bool live = b < nblk;            // uniform within a 16-lane row, may differ between rows of a wave
            float t[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float4 q = live ? v[rd][k] : float4{0.f, 0.f, 0.f, 0.f};
                if (live && pro == PRO_RMSNORM) {
                    const float4 w4 = *(const float4*)(nw + b * 256 + sub * 16 + k * 4);
                    q.x = (q.x * scale) * w4.x;
                    q.y = (q.y * scale) * w4.y;
                    q.z = (q.z * scale) * w4.z;
                    q.w = (q.w * scale) * w4.w;
                }
                t[4 * k] = q.x; t[4 * k + 1] = q.y; t[4 * k + 2] = q.z; t[4 * k + 3] = q.w;
            }
            float am = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) am = fmaxf(am, fabsf(t[e]));
            float amax = am;
            amax = fmaxf(amax, lane_xor1(amax));
            amax = fmaxf(amax, lane_xor2(amax));
            amax = fmaxf(amax, lane_xor4(amax));
            amax = fmaxf(amax, lane_xor8(amax));
            // first element (lowest index) attaining amax keeps its sign
            const unsigned long long hit = __ballot(am == amax);
            const unsigned row_bits = (unsigned)((hit >> (lane & 48)) & 0xFFFFu);
            const int first = (lane & 48) + (__ffsll((unsigned long long)row_bits) - 1);
            float mine = 0.0f;
#pragma unroll
            for (int e = 15; e >= 0; --e) mine = (fabsf(t[e]) == amax) ? t[e] : mine;
            const float maxv = __shfl(mine, first);
            int packed[4] = {0, 0, 0, 0}, s16 = 0;
            float d = 0.0f;
            if (amax != 0.0f) {
                const float iscale = -128.f / maxv;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int q = ((int)f32_to_bits(fmaf(iscale, t[e], 12582912.f)) & 0x007fffff) - 0x00400000;
                    q = q > 127 ? 127 : q;
                    packed[e >> 2] |= (q & 0xff) << (8 * (e & 3));
                    s16 += q;
                }
                d = 1.0f / iscale;
            }
            const int s32 = s16 + lane_xor1(s16);
            if (live) {
#pragma unroll
                for (int k = 0; k < 4; ++k) L.q8[b * 64 + sub * 4 + k] = packed[k];
                L.bsums[b * 16 + sub] = s16;
                if ((sub & 1) == 0) L.sb[b * 8 + (sub >> 1)] = s32;
                if (sub == 0) L.yd[b] = d;
            }
        }
    }
    __syncthreads();
}

template <int NT, int MAXK>
DEV void prologue_q8k_exact16_ln(ActLdsX<MAXK>& L, const float* __restrict__ x, const float* __restrict__ nw, int K, int pro,
                              float eps, const float* __restrict__ nbias = nullptr) {
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, sub = tid & 15, grp = tid >> 4;
    constexpr int NW = NT / 64, NG = NT / 16;
    constexpr int ROUNDS = (MAXK / 256 + NG - 1) / NG;
    const int nblk = K >> 8;
    // A wave whose four 16-lane rows are all past the last block has nothing to quantize: it must SKIP the arithmetic
    // (wave-uniform branches), not run it predicated off — the prologue is VALU-issue bound (about 250 wave
    // instructions), and for K = 4096 only 4 of the 16 waves (one per SIMD) are live.
    const bool wave_live = uniform_int(wv * 4) < nblk;
    float4 v[ROUNDS][4];
    double s = 0.0;
    if (wave_live) {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int b = grp + rd * NG;
            if (b < nblk) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[rd][k] = *(const float4*)(x + b * 256 + sub * 16 + k * 4);
                    if (pro == PRO_RMSNORM) {
                        s += (double)(v[rd][k].x * v[rd][k].x);
                        s += (double)(v[rd][k].y * v[rd][k].y);
                        s += (double)(v[rd][k].z * v[rd][k].z);
                        s += (double)(v[rd][k].w * v[rd][k].w);
                    }
                }
            }
        }
    }
    float scale = 1.0f;
    if (pro == PRO_RMSNORM) {
        if (wave_live) {
            s = wave_sum_fast(s);
            if (lane == 0) L.red[wv] = s;
        } else if (lane == 0) {
            L.red[wv] = 0.0;
        }
        __syncthreads();
        if (wave_live) {
            double tot = 0.0;
            for (int w = 0; w < NW; ++w) tot += L.red[w];
            const float mean = (float)(tot / (double)K);
            scale = 1.0f / sqrtf(mean + eps);
        }
    } else if (pro == PRO_LAYERNORM) {
        // ggml_compute_forward_norm_f32 (ggml.c:10605-10654): double sum -> f32 mean; v = x - mean; double sum of v*v ->
        // f32 variance; scale = 1/sqrtf(variance + eps).  v replaces x in the registers.
        double s1 = 0.0;
        if (wave_live) {
#pragma unroll
            for (int rd = 0; rd < ROUNDS; ++rd) {
                if (grp + rd * NG < nblk) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        s1 += (double)v[rd][k].x; s1 += (double)v[rd][k].y; s1 += (double)v[rd][k].z; s1 += (double)v[rd][k].w;
                    }
                }
            }
            s1 = wave_sum_fast(s1);
        }
        if (lane == 0) L.red[wv] = wave_live ? s1 : 0.0;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < NW; ++w) tot += L.red[w];
        const float mean = (float)(tot / (double)K);
        __syncthreads();   // L.red is reused for the second moment
        double s2 = 0.0;
        if (wave_live) {
#pragma unroll
            for (int rd = 0; rd < ROUNDS; ++rd) {
                if (grp + rd * NG < nblk) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[rd][k].x -= mean; v[rd][k].y -= mean; v[rd][k].z -= mean; v[rd][k].w -= mean;
                        s2 += (double)(v[rd][k].x * v[rd][k].x); s2 += (double)(v[rd][k].y * v[rd][k].y);
                        s2 += (double)(v[rd][k].z * v[rd][k].z); s2 += (double)(v[rd][k].w * v[rd][k].w);
                    }
                }
            }
            s2 = wave_sum_fast(s2);
        }
        if (lane == 0) L.red[wv] = wave_live ? s2 : 0.0;
        __syncthreads();
        double tot2 = 0.0;
        for (int w = 0; w < NW; ++w) tot2 += L.red[w];
        const float variance = (float)(tot2 / (double)K);
        scale = 1.0f / sqrtf(variance + eps);
    }
    if (wave_live) {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int b = grp + rd * NG;
            const bool live = b < nblk;            // uniform within a 16-lane row, may differ between rows of a wave
            float t[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float4 q = live ? v[rd][k] : float4{0.f, 0.f, 0.f, 0.f};
                if (live && pro != PRO_PLAIN) {
                    const float4 w4 = *(const float4*)(nw + b * 256 + sub * 16 + k * 4);
                    q.x = (q.x * scale) * w4.x;
                    q.y = (q.y * scale) * w4.y;
                    q.z = (q.z * scale) * w4.z;
                    q.w = (q.w * scale) * w4.w;
                    if (pro == PRO_LAYERNORM) {
                        const float4 b4 = *(const float4*)(nbias + b * 256 + sub * 16 + k * 4);
                        q.x += b4.x; q.y += b4.y; q.z += b4.z; q.w += b4.w;
                    }
                }
                t[4 * k] = q.x; t[4 * k + 1] = q.y; t[4 * k + 2] = q.z; t[4 * k + 3] = q.w;
            }
            float am = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) am = fmaxf(am, fabsf(t[e]));
            float amax = am;
            amax = fmaxf(amax, lane_xor1(amax));
            amax = fmaxf(amax, lane_xor2(amax));
            amax = fmaxf(amax, lane_xor4(amax));
            amax = fmaxf(amax, lane_xor8(amax));
            // first element (lowest index) attaining amax keeps its sign
            const unsigned long long hit = __ballot(am == amax);
            const unsigned row_bits = (unsigned)((hit >> (lane & 48)) & 0xFFFFu);
            const int first = (lane & 48) + (__ffsll((unsigned long long)row_bits) - 1);
            float mine = 0.0f;
#pragma unroll
            for (int e = 15; e >= 0; --e) mine = (fabsf(t[e]) == amax) ? t[e] : mine;
            const float maxv = __shfl(mine, first);
            int packed[4] = {0, 0, 0, 0}, s16 = 0;
            float d = 0.0f;
            if (amax != 0.0f) {
                const float iscale = -128.f / maxv;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int q = ((int)f32_to_bits(fmaf(iscale, t[e], 12582912.f)) & 0x007fffff) - 0x00400000;
                    q = q > 127 ? 127 : q;
                    packed[e >> 2] |= (q & 0xff) << (8 * (e & 3));
                    s16 += q;
                }
                d = 1.0f / iscale;
            }
            const int s32 = s16 + lane_xor1(s16);
            if (live) {
#pragma unroll
                for (int k = 0; k < 4; ++k) L.q8[b * 64 + sub * 4 + k] = packed[k];
                L.bsums[b * 16 + sub] = s16;
                if ((sub & 1) == 0) L.sb[b * 8 + (sub >> 1)] = s32;
                if (sub == 0) L.yd[b] = d;
            }
        }
    }
    __syncthreads();
}

// DPP form of quad_transpose_reduce (lanes g, g^2, g^4, g^6 of a row).
DEV int quad_transpose_reduce_dpp(int p0, int p1, int p2, int p3, int c) {
    const bool odd = (c & 1) != 0;
    const int send0 = odd ? p0 : p2, send1 = odd ? p1 : p3;
    const int keep0 = odd ? p2 : p0, keep1 = odd ? p3 : p1;
    const int q0 = keep0 + lane_xor2(send0);
    const int q1 = keep1 + lane_xor2(send1);
    const bool up = (c & 2) != 0;
    const int send = up ? q0 : q1;
    const int keep = up ? q1 : q0;
    return keep + lane_xor4(send);
}
DEV float hsum8_exact_dpp(float acc) {
    const float t = acc + lane_xor1(acc);
    const float u = t + lane_xor2(t);
    return u + lane_xor4(u);
}

// Register image of one chunk (UB blocks of this wave) of one tile.
// One block image (this lane's pieces of one K-block of an 8-row tile) in named registers — no arrays, so the
// compiler keeps the 4-deep load pipeline in VGPRs (an array-of-vectors image was demoted to scratch by hipcc).
struct BlockRegs {
    u32x4 v0, v1, v2;
    uint32_t dd;
};

// Per-lane constants of the tile geometry (computed once per kernel).
struct LaneGeom {
    int r, g, c, h;           // row in tile, unit in block, chunk, AVX half
    uint32_t off_hdr, off_qs, off_qh5;     // Q4_K / Q5_K byte offsets inside a record
    uint32_t off6_d, off6_sc, off6_qh, off6_ql;  // Q6_K byte offsets inside a record
    int a45, a6;              // LDS word offsets of this lane's activation bytes inside a block (Q4/5_K, Q6_K)
    int sh16;                 // 16*(c&1): which half of the packed scale words
    int s_lo6, s_hi6, sc_sh6; // Q6_K shifts
};
DEV LaneGeom lane_geom(int lane) {
    LaneGeom G;
    G.r = lane >> 3; G.g = lane & 7; G.c = G.g >> 1; G.h = G.g & 1;
    G.off_hdr = G.r * 16;
    G.off_qs = G.r * 128 + G.g * 16;
    G.off_qh5 = 128 + G.r * 32 + G.h * 16;
    const int n = G.g >> 2, gg = G.g & 3, kq = gg >> 1;
    G.off6_d = G.r * 2;
    G.off6_sc = 16 + G.r * 16;
    G.off6_qh = 144 + G.r * 64 + n * 32 + G.h * 16;
    G.off6_ql = 656 + G.r * 128 + G.g * 16;
    G.a45 = 16 * G.c + 4 * G.h;
    G.a6 = 32 * n + 4 * gg;
    G.sh16 = 16 * (G.c & 1);
    G.s_lo6 = 2 * kq; G.s_hi6 = 4 + 2 * kq; G.sc_sh6 = 8 * gg;
    return G;
}

// rec_base: wave-uniform pointer to the (tile, block) record.
DEV BlockRegs block_load2(int type, const uint8_t* rec_base, const LaneGeom& G) {
    BlockRegs R;
    if (type == GT_Q6_K) {
        R.dd = *(const uint16_t*)(rec_base + G.off6_d);
        R.v0 = ld_stream16(rec_base + G.off6_sc);
        R.v2 = ld_stream16(rec_base + G.off6_qh);
        R.v1 = ld_stream16(rec_base + G.off6_ql);
    } else if (type == GT_Q5_K) {
        R.dd = 0;
        R.v0 = ld_stream16(rec_base + G.off_hdr);
        R.v1 = ld_stream16(rec_base + 384 + G.off_qs);
        R.v2 = ld_stream16(rec_base + G.off_qh5);
    } else {
        R.dd = 0;
        R.v0 = ld_stream16(rec_base + G.off_hdr);
        R.v1 = ld_stream16(rec_base + G.off_qs + 128);
        R.v2 = R.v0;
    }
    return R;
}

// Integer work of one block (DPP transposes, 24-bit multiplies), results into the chain buffer.
template <int MAXK, int MAXNB>
DEV void block_to_chain3(int type, int b, const ActLdsX<MAXK>& L, ChainBuf<MAXNB>& C, int lane, const LaneGeom& G,
                         const BlockRegs& R) {
    const int c = G.c;
    if (type == GT_Q4_K || type == GT_Q5_K) {
        const bool q5 = type == GT_Q5_K;
        const int* alo = &L.q8[b * 64 + G.a45];
        const int* ahi = alo + 8;
        // 6-bit scales/mins of sub-blocks 2c, 2c+1 as byte pairs (reference get_scale_min_k4, k_quants.c:306-314)
        const uint32_t A = R.v0[1] >> G.sh16, B = R.v0[2] >> G.sh16, C3 = R.v0[3] >> G.sh16;
        const uint32_t scL = A & 0x3F3Fu, mL = B & 0x3F3Fu;
        const uint32_t scH = (C3 & 0x0F0Fu) | ((A >> 2) & 0x3030u);
        const uint32_t mH = ((C3 >> 4) & 0x0F0Fu) | ((B >> 2) & 0x3030u);
        const uint32_t scp = c < 2 ? scL : scH, mp = c < 2 ? mL : mH;
        const int sc_lo = (int)(scp & 0xFF), sc_hi = (int)(scp >> 8), m_lo = (int)(mp & 0xFF), m_hi = (int)(mp >> 8);
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
        if (G.h != 0) prod = 0;
        if (q5) {
            prod += lane_xor2(prod);
            prod += lane_xor4(prod);
        }
        if (G.h == 0) C.PM[b][G.r * 4 + c] = (float)prod;
        if (G.g == 0) {
            const float yd = L.yd[b];
            C.D[b][G.r] = yd * f16_bits_to_f32((uint16_t)(R.v0[0] & 0xFFFF));
            C.DM[b][G.r] = -yd * f16_bits_to_f32((uint16_t)(R.v0[0] >> 16));
        }
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
            // (q - 32) . a  ==  q . a - 32 * sum(a): fold the -32 into the dot4 chain with a constant operand
            const int dl = sdot4((int)lo, alo[k], sdot4((int)0xE0E0E0E0u, alo[k], 0));
            const int dh = sdot4((int)hi, ahi[k], sdot4((int)0xE0E0E0E0u, ahi[k], 0));
            part[k] = mul24(sc_lo, dl) + mul24(sc_hi, dh);
        }
        const int sumi = quad_transpose_reduce_dpp(part[0], part[1], part[2], part[3], c);
        C.S[b][lane] = (float)sumi;
        if (G.g == 0) C.D[b][G.r] = L.yd[b] * f16_bits_to_f32((uint16_t)R.dd);
    }
}

// One wave: the reference's sequential accumulation over all blocks (operands preloaded 16 blocks at a time so the
// dependent part is a bare fma chain), then its reduction tree.
template <int MAXNB>
DEV float chain_reduce2(int type, int nb, const ChainBuf<MAXNB>& C, int lane) {
    const int r = lane >> 3, g = lane & 7, c = g >> 1, h = g & 1;
    float acc = 0.0f, accm = 0.0f;
    const bool mins = type != GT_Q6_K;
    for (int b0 = 0; b0 < nb; b0 += 16) {
        float dv[16], sv[16], mv[16], pv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int b = (b0 + u < nb) ? b0 + u : nb - 1;
            dv[u] = C.D[b][r];
            sv[u] = C.S[b][lane];
            mv[u] = mins ? C.DM[b][r] : 0.0f;
            pv[u] = (mins && h == 0) ? C.PM[b][r * 4 + c] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (b0 + u < nb) {
                acc = fmaf(dv[u], sv[u], acc);
                accm = fmaf(mv[u], pv[u], accm);
            }
        }
    }
    float tot = hsum8_exact_dpp(acc);
    if (!mins) return tot;
    if (type == GT_Q4_K) {
        const float wsum = accm + lane_xor4(accm);
        accm = wsum + lane_xor2(wsum);
    }
    accm = __shfl(accm, lane & ~7);
    return tot + accm;
}

// Work cursor of a persistent workgroup: walks items -> units (matrix, tile) -> block steps with adds and compares
// only (no integer divisions; everything here is wave-uniform and lives in SGPRs).
struct StepCursor {
    int item, part, chunk;     // current item, unit inside the item (gate/up), block step inside the unit
    int j, tile, unit_seq;     // job index, tile index inside the job's matrix, running unit number (chain buffer parity)
    int type, nb, M;
    const uint8_t* tile_base;  // first record of the tile
    uint32_t rec;
    bool valid;
};
DEV void cursor_set_unit(StepCursor& c, const MatvecArgs& a) {
    int j = 0;
    if (!a.gateup) {
        if (a.njobs > 1 && c.item >= a.job[1].pair0) j = 1;
        if (a.njobs > 2 && c.item >= a.job[2].pair0) j = 2;
    }
    c.j = j;
    const DevMat& w = a.gateup ? a.job[c.part].w : a.job[j].w;
    c.tile = c.item - (a.gateup ? 0 : a.job[j].pair0);
    c.type = w.type; c.nb = w.nb; c.M = w.M;
    c.rec = (uint32_t)tile8_record_bytes(w.type);
    c.tile_base = w.p[0] + (size_t)c.tile * w.nb * c.rec;
}
DEV void cursor_init(StepCursor& c, const MatvecArgs& a, int first_item) {
    c.item = first_item; c.part = 0; c.chunk = 0; c.unit_seq = 0;
    c.valid = first_item < a.n_pairs;
    if (c.valid) cursor_set_unit(c, a);
}
DEV void cursor_next(StepCursor& c, const MatvecArgs& a, int cpu, int upi, int stride) {
    if (!c.valid) return;
    if (++c.chunk < cpu) return;
    c.chunk = 0;
    ++c.unit_seq;
    if (++c.part >= upi) {
        c.part = 0;
        c.item += stride;
        if (c.item >= a.n_pairs) { c.valid = false; return; }
    }
    cursor_set_unit(c, a);
}

// Fused launch, design C: persistent workgroups; the loads of the next four block steps are in flight while the current
// one is unpacked; a unit (one 8-row tile of one matrix) ends with a barrier, after which a rotating wave replays the
// f32 chain from LDS and runs the epilogue while the others continue with the next unit.
template <int NT, int MAXK, int UB>
__global__ void __launch_bounds__(NT) matvec_exact2_kernel(const MatvecArgs a) {
    constexpr int NW = NT / 64;
    constexpr int MAXNB = MAXK / 256;
    __shared__ ActLdsX<MAXK> L;
    __shared__ ChainBuf<MAXNB> CB[2];
    const int lane = lane_id();
    const int wv = uniform_int(wave_id());
    const LaneGeom G = lane_geom(lane);
    const int nb0 = a.job[0].w.nb;                      // all jobs of a launch share K
    const int cpu = (nb0 + NW - 1) / NW;                // block steps per unit for every wave (tail blocks masked)
    const int upi = a.gateup ? 2 : 1;
    const int stride = (int)gridDim.x;

    StepCursor lc, cc;                                  // load cursor (4 steps ahead) and compute cursor
    cursor_init(lc, a, (int)blockIdx.x);
    cursor_init(cc, a, (int)blockIdx.x);
    auto load_step = [&]() __attribute__((always_inline)) -> BlockRegs {
        BlockRegs R;
        if (lc.valid) {
            int b = wv + lc.chunk * NW;
            b = b < lc.nb ? b : lc.nb - 1;
            R = block_load2(lc.type, lc.tile_base + (size_t)b * lc.rec, G);
        } else {
            R.v0 = R.v1 = R.v2 = u32x4{0, 0, 0, 0};
            R.dd = 0;
        }
        cursor_next(lc, a, cpu, upi, stride);
        return R;
    };

    BlockRegs R0 = load_step(), R1 = load_step(), R2 = load_step(), R3 = load_step();
    prologue_q8k_exact16<NT, MAXK>(L, a.x, a.norm_w, a.K, a.pro, a.eps);
    const int pos = a.pos ? *a.pos : 0;
    float res_gate = 0.0f;

    auto finish_unit = [&]() __attribute__((always_inline)) {  // every wave, after the last block step of a unit
        __syncthreads();
        const int item_seq = a.gateup ? (cc.unit_seq >> 1) : cc.unit_seq;
        const int cw = item_seq & (NW - 1);
        if (wv != cw) return;
        const float res = chain_reduce2<MAXNB>(cc.type, cc.nb, CB[cc.unit_seq & 1], lane);
        const int row = cc.tile * 8 + G.r;
        const bool own = G.g == 0 && row < cc.M;
        if (a.gateup) {
            if (cc.part == 0) { res_gate = res; return; }
            if (own) a.out[row] = f16_bits_to_f32(a.silu_tab[f32_to_f16_bits(res_gate)]) * res;
            return;
        }
        const int epi = a.job[cc.j].epi;
        if (epi == EPI_STORE) {
            if (own) a.out[row] = res;
        } else if (epi == EPI_ADD) {
            if (own) a.out[row] = res + a.res[row];
        } else if (epi == EPI_V) {
            if (own) a.vcache[(size_t)row * a.v_stride + pos] = f32_to_f16_bits(res);
        } else if (epi == EPI_GELU) {
            if (own) a.out[row] = f16_bits_to_f32(a.gelu_tab[f32_to_f16_bits(res)]);
        } else if (epi == EPI_ADD2) {
            if (own) a.out[row] = (res + a.res[row]) + a.res2[row];
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
    };
    auto compute_step = [&](const BlockRegs& R) __attribute__((always_inline)) {
        if (!cc.valid) return;
        const int b = wv + cc.chunk * NW;
        if (b < cc.nb) block_to_chain3<MAXK, MAXNB>(cc.type, b, L, CB[cc.unit_seq & 1], lane, G, R);
        if (cc.chunk == cpu - 1) finish_unit();
        cursor_next(cc, a, cpu, upi, stride);
    };
    while (cc.valid) {
        compute_step(R0); R0 = load_step();
        compute_step(R1); R1 = load_step();
        compute_step(R2); R2 = load_step();
        compute_step(R3); R3 = load_step();
    }
}
