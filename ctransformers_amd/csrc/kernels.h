// Hand-written CDNA4 (gfx950, wave64) kernels for the decode hot path that replaces ggml_compute_forward's ops
// (reference models/ggml/ggml.c:11031-11245 mul_mat, :10674 rms_norm, :12430 rope, :12009 soft_max, :8289 dup/cpy,
// k_quants.c:1191 quantize_row_q8_K, :2550/:3081/:3650 vec_dot_q{4,5,6}_K_q8_K).
//
// Numerics contract: the kernels reproduce the reference's *quantization points* —
// activations re-quantized to Q8_K (K-quant weights) or Q8_0 (Q4_0/Q8_0 weights) before every weight mat-vec with
// the reference's exact rounding, integer block dot products, fp16 K/V/Q/P, fp16-table exp/SiLU, double-precision
// norm / softmax sums.  Only the order of f32 partial sums inside one dot product differs (~1e-7 relative).
//
// Mat-vec design (HBM-bound, integer work — no MFMA): one wavefront streams whole weight rows with 16-byte
// non-temporal loads, lane l owning the 16-byte units {l, l+64, ...} of the row; the quantized activation vector is
// built once per workgroup in LDS by a fused prologue (RMSNorm -> Q8_K), each lane keeps the activation bytes that
// pair with its units in VGPRs, so the inner loop is: 2 loads -> nibble unpack -> v_dot4_i32_i8 -> 6-bit scale
// unpack -> f32 FMA; a 6-step xor-butterfly reduces the 64 lane partials; RoPE / KV-cache store / residual add /
// SiLU*up are epilogues of the same launch.
#pragma once
#include "gpu.h"
#include "quant.h"

// ------------------------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------------------------
DEV float bits_to_f32(uint32_t u) {
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
DEV uint32_t f32_to_bits(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
}
// reference k_quants.c:52-57 (round-half-even via the 1.5*2^23 magic constant)
DEV int nearest_int_magic(float fval) {
    float val = fval + 12582912.f;
    int i = (int)f32_to_bits(val);
    return (i & 0x007fffff) - 0x00400000;
}

enum { PRO_PLAIN = 0, PRO_RMSNORM = 1, PRO_LAYERNORM = 2 };   // LAYERNORM: ggml_norm, *w, +b (falcon, llama.cpp:2611-2614)
enum { EPI_STORE = 0, EPI_ADD = 1, EPI_ROPE_Q = 2, EPI_ROPE_K = 3, EPI_V = 4, EPI_SILU_MUL = 5,
       EPI_GELU = 6,     // out = gelu_table[fp16(y)]                       (falcon ffn_up, ggml.c:3568-3575)
       EPI_ADD2 = 7,     // out = (y + res) + res2                          (falcon: ffn + attn_out + inpL, llama.cpp:2763-2764)
       EPI_BIAS_STORE = 8,   // out = bias + y                              (gpt2 c_attn, gpt2.cc:470-473)
       EPI_BIAS_ADD = 9,     // out = (bias + y) + res                      (gpt2 c_proj / mlp proj + residual, gpt2.cc:590-600, :640-646)
       EPI_BIAS_GELU = 10 }; // out = gelu_table[fp16(bias + y)]            (gpt2 mlp fc, gpt2.cc:625-631)

struct MatJob {
    DevMat w;
    int epi;
    int pair0;  // first global pair index of this job
};

// K cache layout: [kv_head][n_ctx][head_dim] fp16 — position-major inside a head, so the decode attention workgroup of a
// head streams ONE contiguous region (the reference keeps [pos][n_embd_gqa], llama.cpp:2342-2347, which at decode time
// is a 256-byte read every 8 KB per head).  The layout is internal; the numerics do not depend on it.
DEV size_t kcache_off(int pos, int row, int head_dim, int n_ctx) {
    const int hk = row / head_dim;
    return ((size_t)hk * n_ctx + pos) * head_dim + (size_t)(row - hk * head_dim);
}

struct MatvecArgs {
    MatJob job[3];
    int njobs;
    int n_pairs;        // total pairs over all jobs
    int n_groupA;       // generation-5 kernels: items of the first type-homogeneous job group (rest = group B)
    int gateup;         // 1: job[0]=gate, job[1]=up, pair t = (gate row t, up row t), epilogue SiLU(gate)*up
    int K;              // input length
    int pro;            // PRO_*
    const float* x;     // input activation f32[K]
    const float* norm_w;  // norm weight f32[K] (PRO_RMSNORM / PRO_LAYERNORM)
    const float* norm_b;  // norm bias f32[K] (PRO_LAYERNORM)
    float eps;
    // epilogue operands
    float* out;             // EPI_STORE / EPI_ADD / EPI_SILU_MUL destination
    const float* res;       // EPI_ADD / EPI_ADD2 residual
    const float* res2;      // EPI_ADD2 second residual
    const float* bias;      // EPI_BIAS_* row bias f32[M]
    const uint16_t* gelu_tab;  // 65536-entry fp16->fp16 GELU table (EPI_GELU)
    uint16_t* q_f16;        // EPI_ROPE_Q destination (fp16 query, n_head*head_dim)
    uint16_t* kcache;       // this layer's K cache  [n_head_kv][n_ctx][head_dim] fp16 (kcache_off)
    uint16_t* vcache;       // this layer's V cache  [n_embd_gqa][n_ctx] fp16 (transposed, as the reference keeps it)
    const float* rope_cs;   // [n_ctx][head_dim/2][2] cos,sin (host-built with the reference's iterative theta)
    const int* pos;         // device scalar: position of this token
    int n_ctx, head_dim, n_embd_gqa, v_stride;
    const uint16_t* silu_tab;  // 65536-entry fp16->fp16 table (reference ggml.c:4328-4332)
    float* dbg_sink;           // measurement only: always-valid scratch the ablation paths may write to
    int dbg;                   // measurement only (CT_AMD_DBG): 1 skip prologue, 2 skip block math, 4 skip chain+epilogue, 8 skip weight loads, 16 return at once
};

// Per-lane copy of the activation bytes that pair with this lane's weight units.
template <int I> struct LaneActs {
    int lo[I][4];
    int hi[I][4];
    float yd[I];
    int bs_lo[I];
    int bs_hi[I];
};

// Workgroup-shared quantized activation vector (Q8_K form: reference k_quants.h:118-126).
template <int MAXK> struct ActLds {
    int q8[MAXK / 4];          // int8 quants, 4 per word
    float yd[MAXK / 256];      // block scale d
    int bsums[MAXK / 16];      // sums of 16 quants (int16 in the reference; exact in int)
    double red[16];
    float bcast[4];
};

// ------------------------------------------------------------------------------------------------------------------
// Prologue: (RMSNorm ->) Q8_K quantization of the activation vector into LDS, once per workgroup.
//   RMSNorm: reference ggml.c:10700-10716 (double sum of f32 squares, scale = 1/sqrtf(mean+eps)), then the broadcast
//   multiply by the norm weight (llama.cpp:2286).  Q8_K: reference k_quants.c:1191-1226.
// ------------------------------------------------------------------------------------------------------------------
template <int NT, int MAXK>
DEV void prologue_q8k(ActLds<MAXK>& L, const float* __restrict__ x, const float* __restrict__ nw, int K, int pro, float eps) {
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    constexpr int NW = NT / 64;
    float scale = 1.0f;
    if (pro == PRO_RMSNORM) {
        double s = 0.0;
        for (int i = tid * 4; i < K; i += NT * 4) {
            const float4 v = *(const float4*)(x + i);
            s += (double)(v.x * v.x);
            s += (double)(v.y * v.y);
            s += (double)(v.z * v.z);
            s += (double)(v.w * v.w);
        }
        s = wave_sum(s);
        if (lane == 0) L.red[wv] = s;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < NW; ++w) tot += L.red[w];
        const float mean = (float)(tot / (double)K);
        scale = 1.0f / sqrtf(mean + eps);
    }
    const int nblk = K >> 8;
    for (int b = wv; b < nblk; b += NW) {
        const int e = b * 256 + lane * 4;
        float4 v = *(const float4*)(x + e);
        if (pro == PRO_RMSNORM) {
            const float4 w4 = *(const float4*)(nw + e);
            v.x = (v.x * scale) * w4.x;
            v.y = (v.y * scale) * w4.y;
            v.z = (v.z * scale) * w4.z;
            v.w = (v.w * scale) * w4.w;
        }
        const float a0 = fabsf(v.x), a1 = fabsf(v.y), a2 = fabsf(v.z), a3 = fabsf(v.w);
        const float am = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
        const float amax = wave_max(am);
        // first element (lowest index) attaining amax keeps its sign: `if (ax > amax) { amax = ax; max = x[j]; }`
        const unsigned long long hit = __ballot(am == amax);
        const int first = __ffsll(hit) - 1;
        const float mine = (a0 == amax) ? v.x : (a1 == amax) ? v.y : (a2 == amax) ? v.z : v.w;
        const float maxv = __shfl(mine, first);
        int packed = 0, s4 = 0;
        float d = 0.0f;
        if (amax != 0.0f) {
            const float iscale = -128.f / maxv;
            int q0 = nearest_int_magic(iscale * v.x); q0 = q0 > 127 ? 127 : q0;
            int q1 = nearest_int_magic(iscale * v.y); q1 = q1 > 127 ? 127 : q1;
            int q2 = nearest_int_magic(iscale * v.z); q2 = q2 > 127 ? 127 : q2;
            int q3 = nearest_int_magic(iscale * v.w); q3 = q3 > 127 ? 127 : q3;
            packed = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | ((q3 & 0xff) << 24);
            s4 = q0 + q1 + q2 + q3;
            d = 1.0f / iscale;
        }
        L.q8[b * 64 + lane] = packed;
        s4 += __shfl_xor(s4, 1);
        s4 += __shfl_xor(s4, 2);
        if ((lane & 3) == 0) L.bsums[b * 16 + (lane >> 2)] = s4;
        if (lane == 0) L.yd[b] = d;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------------------------
// Lane <-> unit geometry.  Unit u = 16 bytes of the row's nibble plane; u = lane + 64*i; blk = u>>3; g = u&7 is a
// per-lane constant because 64*i is a multiple of 8.
//   Q4_K/Q5_K: 32-byte chunk c = g>>1 holds sub-blocks 2c (low nibbles) and 2c+1 (high nibbles); half h = g&1 selects
//              elements [16h,16h+16) of both sub-blocks.                       (reference k_quants.c:784-821)
//   Q6_K:      half n = g>>2, gg = g&3: low nibbles -> elements 128n+16gg.., high nibbles -> +64; the 2 high bits come
//              from qh[(gg&1)*16..] at shift 2*(gg>>1) (+4 for the high-nibble elements).   (k_quants.c:1123-1170)
// ------------------------------------------------------------------------------------------------------------------
template <int I, int MAXK>
DEV void load_acts(LaneActs<I>& A, const ActLds<MAXK>& L, int type, int nb, int lane) {
    const int g = lane & 7;
    const int U = nb * 8;
#pragma unroll
    for (int i = 0; i < I; ++i) {
        const int u = lane + 64 * i;
        const bool valid = u < U;
        const int blk = valid ? (u >> 3) : 0;
        int e_lo, e_hi, b_lo, b_hi;
        if (type == GT_Q6_K) {
            const int n = g >> 2, gg = g & 3;
            e_lo = blk * 256 + 128 * n + 16 * gg;
            e_hi = e_lo + 64;
            b_lo = blk * 16 + 8 * n + gg;
            b_hi = b_lo + 4;
        } else {
            const int c = g >> 1, h = g & 1;
            e_lo = blk * 256 + 64 * c + 16 * h;
            e_hi = e_lo + 32;
            b_lo = blk * 16 + 4 * c + h;
            b_hi = b_lo + 2;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            A.lo[i][k] = valid ? L.q8[(e_lo >> 2) + k] : 0;
            A.hi[i][k] = valid ? L.q8[(e_hi >> 2) + k] : 0;
        }
        A.yd[i] = valid ? L.yd[blk] : 0.0f;
        A.bs_lo[i] = valid ? L.bsums[b_lo] : 0;
        A.bs_hi[i] = valid ? L.bsums[b_hi] : 0;
    }
}

// 6-bit scale/min unpack for this lane's two sub-blocks (reference get_scale_min_k4, k_quants.c:306-314).
DEV void scale_min_pair(uint32_t w1, uint32_t w2, uint32_t w3, int c, int& sc_lo, int& sc_hi, int& m_lo, int& m_hi) {
    const int sh = 16 * (c & 1);
    const uint32_t t1 = w1 >> sh, t2 = w2 >> sh, t3 = w3 >> sh;
    if (c < 2) {
        sc_lo = t1 & 63;
        sc_hi = (t1 >> 8) & 63;
        m_lo = t2 & 63;
        m_hi = (t2 >> 8) & 63;
    } else {
        sc_lo = (t3 & 0xF) | (((t1 >> 6) & 3) << 4);
        sc_hi = ((t3 >> 8) & 0xF) | (((t1 >> 14) & 3) << 4);
        m_lo = ((t3 >> 4) & 0xF) | (((t2 >> 6) & 3) << 4);
        m_hi = ((t3 >> 12) & 0xF) | (((t2 >> 14) & 3) << 4);
    }
}

// One weight row x the lane's activation units; returns this lane's f32 partial (reference per-block algebra:
// k_quants.c:2651-2720 Q4_K, :3081+ Q5_K, :3800+ Q6_K — d = y.d * fp16(x.d), acc = fma(d, (float)isum, acc),
// min term acc_m = fma(-y.d * fp16(x.dmin), (float)sum(m_j * bsums_j), acc_m)).
template <int I>
DEV float row_partial(const DevMat& w, int row, const LaneActs<I>& A, int lane) {
    const int nb = w.nb;
    const int U = nb * 8;
    const int g = lane & 7;
    float acc = 0.0f;
    if (w.type == GT_Q4_K || w.type == GT_Q5_K) {
        const uint8_t* qs_row = w.p[0] + (size_t)row * nb * 128;
        const uint8_t* hdr_row = w.p[1] + (size_t)row * nb * 16;
        const uint8_t* qh_row = (w.type == GT_Q5_K) ? w.p[2] + (size_t)row * nb * 32 : nullptr;
        const int c = g >> 1, h = g & 1;
        u32x4 qs[I], hd[I], qh[I];
#pragma unroll
        for (int i = 0; i < I; ++i) {
            const int u = lane + 64 * i;
            const int uu = u < U ? u : 0;
            qs[i] = ld_stream16(qs_row + (size_t)uu * 16);
            hd[i] = ld_stream16(hdr_row + (size_t)(uu >> 3) * 16);
            if (w.type == GT_Q5_K) qh[i] = ld_stream16(qh_row + (size_t)(uu >> 3) * 32 + h * 16);
        }
#pragma unroll
        for (int i = 0; i < I; ++i) {
            int dlo = 0, dhi = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t lo = qs[i][k] & 0x0F0F0F0Fu;
                uint32_t hi = (qs[i][k] >> 4) & 0x0F0F0F0Fu;
                if (w.type == GT_Q5_K) {
                    lo |= ((qh[i][k] >> (2 * c)) & 0x01010101u) << 4;
                    hi |= ((qh[i][k] >> (2 * c + 1)) & 0x01010101u) << 4;
                }
                dlo = sdot4((int)lo, A.lo[i][k], dlo);
                dhi = sdot4((int)hi, A.hi[i][k], dhi);
            }
            int sc_lo, sc_hi, m_lo, m_hi;
            scale_min_pair(hd[i][1], hd[i][2], hd[i][3], c, sc_lo, sc_hi, m_lo, m_hi);
            const int isum = sc_lo * dlo + sc_hi * dhi;
            const int msum = m_lo * A.bs_lo[i] + m_hi * A.bs_hi[i];
            const float d = A.yd[i] * f16_bits_to_f32((uint16_t)(hd[i][0] & 0xFFFF));
            const float dmin = -A.yd[i] * f16_bits_to_f32((uint16_t)(hd[i][0] >> 16));
            acc = fmaf(d, (float)isum, acc);
            acc = fmaf(dmin, (float)msum, acc);
        }
    } else {  // GT_Q6_K
        const uint8_t* ql_row = w.p[0] + (size_t)row * nb * 128;
        const uint8_t* sc_row = w.p[1] + (size_t)row * nb * 16;
        const uint8_t* qh_row = w.p[2] + (size_t)row * nb * 64;
        const uint16_t* d_row = (const uint16_t*)w.p[3] + (size_t)row * nb;
        const int n = g >> 2, gg = g & 3;
        const int s_lo = 2 * (gg >> 1), s_hi = 4 + 2 * (gg >> 1);
        u32x4 ql[I], qh[I], sc[I];
        uint16_t dd[I];
#pragma unroll
        for (int i = 0; i < I; ++i) {
            const int u = lane + 64 * i;
            const int uu = u < U ? u : 0;
            const int blk = uu >> 3;
            ql[i] = ld_stream16(ql_row + (size_t)uu * 16);
            qh[i] = ld_stream16(qh_row + (size_t)blk * 64 + n * 32 + (gg & 1) * 16);
            sc[i] = ld_stream16(sc_row + (size_t)blk * 16);
            dd[i] = d_row[blk];
        }
#pragma unroll
        for (int i = 0; i < I; ++i) {
            int dlo = 0, dhi = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t lo = (ql[i][k] & 0x0F0F0F0Fu) | (((qh[i][k] >> s_lo) & 0x03030303u) << 4);
                const uint32_t hi = ((ql[i][k] >> 4) & 0x0F0F0F0Fu) | (((qh[i][k] >> s_hi) & 0x03030303u) << 4);
                dlo = sdot4((int)lo, A.lo[i][k], dlo);
                dhi = sdot4((int)hi, A.hi[i][k], dhi);
            }
            const uint32_t w_lo = n ? sc[i][2] : sc[i][0];
            const uint32_t w_hi = n ? sc[i][3] : sc[i][1];
            const int sc_lo = (int)(int8_t)((w_lo >> (8 * gg)) & 0xFF);
            const int sc_hi = (int)(int8_t)((w_hi >> (8 * gg)) & 0xFF);
            const int isum = sc_lo * (dlo - 32 * A.bs_lo[i]) + sc_hi * (dhi - 32 * A.bs_hi[i]);
            const float d = A.yd[i] * f16_bits_to_f32(dd[i]);
            acc = fmaf(d, (float)isum, acc);
        }
    }
    return acc;
}

// ------------------------------------------------------------------------------------------------------------------
// The fused mat-vec launch: prologue (norm+quantize) -> wave-per-row-pair streaming dot -> epilogue.
// ------------------------------------------------------------------------------------------------------------------
template <int NT, int I, int MAXK>
__global__ void __launch_bounds__(NT) matvec_kq_kernel(const MatvecArgs a) {
    __shared__ ActLds<MAXK> L;
    const int lane = lane_id();
    prologue_q8k<NT, MAXK>(L, a.x, a.norm_w, a.K, a.pro, a.eps);

    constexpr int NW = NT / 64;
    const int gw = (int)blockIdx.x * NW + wave_id();
    const int W = (int)gridDim.x * NW;
    // contiguous, balanced split of the pairs over all waves of the grid
    const int p_begin = (int)(((long long)a.n_pairs * gw) / W);
    const int p_end = (int)(((long long)a.n_pairs * (gw + 1)) / W);

    LaneActs<I> A;
    int cur_type = -1;
    const int pos = a.pos ? *a.pos : 0;

    for (int p = p_begin; p < p_end; ++p) {
        int j = 0;
        if (!a.gateup) {
            if (a.njobs > 1 && p >= a.job[1].pair0) j = 1;
            if (a.njobs > 2 && p >= a.job[2].pair0) j = 2;
        }
        const MatJob& jb = a.job[j];
        const int t = p - jb.pair0;
        const int rowA = a.gateup ? t : 2 * t;
        const int rowB = a.gateup ? t : 2 * t + 1;
        const DevMat& wA = jb.w;
        const DevMat& wB = a.gateup ? a.job[1].w : jb.w;
        const bool hasB = a.gateup || rowB < wA.M;
        if (wA.type != cur_type) {
            load_acts<I, MAXK>(A, L, wA.type, wA.nb, lane);
            cur_type = wA.type;
        }
        float rA = row_partial<I>(wA, rowA, A, lane);
        float rB = row_partial<I>(wB, hasB ? rowB : rowA, A, lane);
        rA = wave_sum(rA);
        rB = wave_sum(rB);
        if (lane == 0) {
            const int epi = a.gateup ? EPI_SILU_MUL : jb.epi;
            if (epi == EPI_STORE) {
                a.out[rowA] = rA;
                if (hasB) a.out[rowB] = rB;
            } else if (epi == EPI_ADD) {
                a.out[rowA] = rA + a.res[rowA];
                if (hasB) a.out[rowB] = rB + a.res[rowB];
            } else if (epi == EPI_SILU_MUL) {
                // reference ggml.c:3625-3632: y = fp16->f32(table_silu[fp16(x)]), then silu * up (llama.cpp:2448)
                const float s = f16_bits_to_f32(a.silu_tab[f32_to_f16_bits(rA)]);
                a.out[rowA] = s * rB;
            } else if (epi == EPI_V) {
                // V stored transposed, fp16 RNE (reference llama.cpp:2319-2329, ggml.c:8407)
                a.vcache[(size_t)rowA * a.v_stride + pos] = f32_to_f16_bits(rA);
                if (hasB) a.vcache[(size_t)rowB * a.v_stride + pos] = f32_to_f16_bits(rB);
            } else {  // EPI_ROPE_Q / EPI_ROPE_K: interleaved pairs (2i,2i+1), reference ggml.c:12522-12539
                const int ip = (rowA % a.head_dim) >> 1;
                const float cs = a.rope_cs[((size_t)pos * (a.head_dim >> 1) + ip) * 2 + 0];
                const float sn = a.rope_cs[((size_t)pos * (a.head_dim >> 1) + ip) * 2 + 1];
                const float o0 = fmaf(rA, cs, -(rB * sn));  // as the reference build contracts it (oracle/mirror.c mir_rope)
                const float o1 = fmaf(rB, cs, rA * sn);
                if (epi == EPI_ROPE_Q) {
                    a.q_f16[rowA] = f32_to_f16_bits(o0);
                    a.q_f16[rowB] = f32_to_f16_bits(o1);
                } else {
                    a.kcache[kcache_off(pos, rowA, a.head_dim, a.n_ctx)] = f32_to_f16_bits(o0);
                    a.kcache[kcache_off(pos, rowB, a.head_dim, a.n_ctx)] = f32_to_f16_bits(o1);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Token embedding: dequantize one row of token_embd (file layout) -> f32   (reference ggml.c:11615-11642 get_rows_q)
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) embed_row_kernel(const uint8_t* __restrict__ raw, int type, int K,
                                                        const int* __restrict__ tokens, const int* __restrict__ state,
                                                        float* __restrict__ out, const float* __restrict__ wpe = nullptr) {
    const int token = tokens[state[0]];
    const uint8_t* row = raw + (size_t)token * ggml_row_bytes(type, K);
    for (int e = (int)(blockIdx.x * blockDim.x + threadIdx.x); e < K; e += (int)(gridDim.x * blockDim.x)) {
        float y;
        if (type == GT_F32) {
            y = ((const float*)row)[e];
        } else if (type == GT_F16) {
            y = f16_bits_to_f32(((const uint16_t*)row)[e]);
        } else if (type == GT_Q8_0) {
            const uint8_t* b = row + (size_t)(e >> 5) * 34;
            const float d = f16_bits_to_f32((uint16_t)(b[0] | (b[1] << 8)));
            y = (float)(int8_t)b[2 + (e & 31)] * d;
        } else if (type == GT_Q4_0) {
            const uint8_t* b = row + (size_t)(e >> 5) * 18;
            const float d = f16_bits_to_f32((uint16_t)(b[0] | (b[1] << 8)));
            const int j = e & 31;
            const int q = (j < 16) ? (b[2 + j] & 0xF) : (b[2 + j - 16] >> 4);
            y = (float)(q - 8) * d;
        } else if (type == GT_Q4_K || type == GT_Q5_K) {
            const int bb = (type == GT_Q4_K) ? 144 : 176;
            const uint8_t* b = row + (size_t)(e >> 8) * bb;
            const float d = f16_bits_to_f32((uint16_t)(b[0] | (b[1] << 8)));
            const float dmin = f16_bits_to_f32((uint16_t)(b[2] | (b[3] << 8)));
            const int el = e & 255, j = el >> 5, l = el & 31, c = j >> 1;
            const uint8_t* s = b + 4;
            int sc, m;
            if (j < 4) { sc = s[j] & 63; m = s[j + 4] & 63; }
            else { sc = (s[j + 4] & 0xF) | ((s[j - 4] >> 6) << 4); m = (s[j + 4] >> 4) | ((s[j] >> 6) << 4); }
            int q;
            if (type == GT_Q4_K) {
                const uint8_t v = b[16 + 32 * c + l];
                q = (j & 1) ? (v >> 4) : (v & 0xF);
            } else {
                const uint8_t v = b[48 + 32 * c + l];
                const uint8_t hb = b[16 + l];
                q = ((j & 1) ? (v >> 4) : (v & 0xF)) + (((hb >> j) & 1) ? 16 : 0);
            }
            const float d1 = d * (float)sc, m1 = dmin * (float)m;
            y = d1 * (float)q - m1;
        } else {  // GT_Q6_K
            const uint8_t* b = row + (size_t)(e >> 8) * 210;
            const int el = e & 255, n = el >> 7, r = el & 127, grp = r >> 5, l = r & 31;
            const uint8_t* ql = b + 64 * n;
            const uint8_t* qh = b + 128 + 32 * n;
            const int8_t* sc = (const int8_t*)(b + 192 + 8 * n);
            const float d = f16_bits_to_f32((uint16_t)(b[208] | (b[209] << 8)));
            const uint8_t lo = (grp & 1) ? ql[l + 32] : ql[l];
            const int nib = (grp & 2) ? (lo >> 4) : (lo & 0xF);
            const int q = (int)(int8_t)(nib | (((qh[l] >> (2 * grp)) & 3) << 4)) - 32;
            const int is = l >> 4;
            y = d * (float)sc[is + 2 * grp] * (float)q;
        }
        if (wpe) y = y + wpe[(size_t)state[1] * K + e];   // gpt2: wte[token] + wpe[pos] (gpt2.cc:441-444)
        out[e] = y;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Decode attention, phase 1: scores[h][i] = (k_i . q_h) * scale for i in [0, n_kv)   (K row-major per position)
//   fp16 q and k, f32 products/accumulation (reference ggml.c:2392 vec_dot_f16), then ggml_scale (llama.cpp:2356).
//   grid (n_head, n_chunks); each wave covers 64/LPP positions per step with LPP = head_dim/8 lanes per position
//   (16 contiguous bytes per lane -> a position's whole head row is one coalesced segment).
// ------------------------------------------------------------------------------------------------------------------
struct AttnArgs {
    const uint16_t* q_f16;   // [n_head*head_dim]
    const uint16_t* kcache;  // layer base [n_head_kv][n_ctx][head_dim] (kcache_off)
    const uint16_t* vcache;  // layer base [n_embd_gqa][n_ctx]
    float* scores;           // [n_head][n_ctx]
    float* out;              // [n_head*head_dim]
    const int* pos;
    const uint16_t* exp_tab;  // fp16 -> fp16 exp table (reference ggml.c:4332)
    int n_head, n_head_kv, head_dim, n_embd_gqa, n_ctx, v_stride;
    float kq_scale;
    int chunk;               // positions per workgroup in phase 1
};

template <int NT>
__global__ void __launch_bounds__(NT) attn_scores_kernel(const AttnArgs a) {
    const int h = (int)blockIdx.x;
    const int n_kv = *a.pos + 1;
    const int c0 = (int)blockIdx.y * a.chunk;
    if (c0 >= n_kv) return;  // whole workgroup exits together
    const int c1 = (c0 + a.chunk < n_kv) ? c0 + a.chunk : n_kv;
    const int lane = lane_id(), wv = wave_id();
    constexpr int NW = NT / 64;
    const int hd = a.head_dim;
    const int lpp = hd >> 3;          // lanes per position (8 halves = 16 bytes each)
    const int ppw = 64 / lpp;         // positions per wave step
    const int sub = lane % lpp;       // which 16-byte piece of the head row
    const int pl = lane / lpp;        // position slot inside the wave step
    const int hk = h / (a.n_head / a.n_head_kv);
    const u32x4 qv = ld16(a.q_f16 + (size_t)h * hd + sub * 8);
    float qf[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        qf[2 * k] = f16_bits_to_f32((uint16_t)(qv[k] & 0xFFFF));
        qf[2 * k + 1] = f16_bits_to_f32((uint16_t)(qv[k] >> 16));
    }
    for (int base = c0 + wv * ppw; base < c1; base += NW * ppw) {
        const int p = base + pl;
        const bool ok = p < c1;
        const int pp = ok ? p : c0;
        const u32x4 kv = ld16(a.kcache + ((size_t)hk * a.n_ctx + pp) * hd + sub * 8);
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s = fmaf(f16_bits_to_f32((uint16_t)(kv[k] & 0xFFFF)), qf[2 * k], s);
            s = fmaf(f16_bits_to_f32((uint16_t)(kv[k] >> 16)), qf[2 * k + 1], s);
        }
        for (int m = 1; m < lpp; m <<= 1) s += __shfl_xor(s, m);
        if (ok && sub == 0) a.scores[(size_t)h * a.n_ctx + p] = s * a.kq_scale;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Decode attention, phase 2: softmax over [0,n_kv) with the reference's fp16-table exp and double sum
// (ggml.c:12047-12069), probabilities rounded to fp16 (INIT of the V*P mat-mul), out[d] = sum_i p_i * v[d][i] in f32.
//   grid (n_head, head_dim/DCH); V is channel-major so lanes stride over positions.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kMaxCtx = 8192;
template <int NT, int DCH>
__global__ void __launch_bounds__(NT) attn_softmax_pv_kernel(const AttnArgs a) {
    __shared__ float prob[kMaxCtx];
    __shared__ double red[NT / 64];
    __shared__ float redf[NT / 64];
    const int h = (int)blockIdx.x;
    const int d0 = (int)blockIdx.y * DCH;
    const int n_kv = *a.pos + 1;
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = wave_id();
    constexpr int NW = NT / 64;
    const float* s = a.scores + (size_t)h * a.n_ctx;
    float mx = -INFINITY;
    for (int i = tid; i < n_kv; i += NT) mx = fmaxf(mx, s[i]);
    mx = wave_max(mx);
    if (lane == 0) redf[wv] = mx;
    __syncthreads();
    mx = redf[0];
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, redf[w]);
    double sum = 0.0;
    for (int i = tid; i < n_kv; i += NT) {
        const float e = f16_bits_to_f32(a.exp_tab[f32_to_f16_bits(s[i] - mx)]);
        prob[i] = e;
        sum += (double)e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < NW; ++w) tot += red[w];
    const float inv = (float)(1.0 / tot);
    for (int i = tid; i < n_kv; i += NT) prob[i] = f16_bits_to_f32(f32_to_f16_bits(prob[i] * inv));
    __syncthreads();
    const int hk = h / (a.n_head / a.n_head_kv);
    for (int dd = wv; dd < DCH; dd += NW) {
        const int d = d0 + dd;
        const uint16_t* vrow = a.vcache + ((size_t)hk * a.head_dim + d) * a.v_stride;
        float acc = 0.0f;
        for (int i = lane; i < n_kv; i += 64) acc = fmaf(f16_bits_to_f32(vrow[i]), prob[i], acc);
        acc = wave_sum(acc);
        if (lane == 0) a.out[(size_t)h * a.head_dim + d] = acc;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Final-norm output as f32 (the "embeddings" the ABI exposes: reference llama.cpp:2963-2968 copies result_norm).
// ------------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(NT) rmsnorm_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ out, int K, float eps) {
    __shared__ double red[NT / 64];
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = wave_id();
    double s = 0.0;
    for (int i = tid; i < K; i += NT) s += (double)(x[i] * x[i]);
    s = wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    double tot = 0.0;
    for (int k = 0; k < NT / 64; ++k) tot += red[k];
    const float mean = (float)(tot / (double)K);
    const float scale = 1.0f / sqrtf(mean + eps);
    for (int i = tid; i < K; i += NT) out[i] = (x[i] * scale) * w[i];
}

// Device-side cursor {step, pos}: lets N token steps (eager launches or hipGraph replays) be queued back to back with
// no host round trip — every kernel reads the position from memory, this one advances it.
// state = {step, pos, n_total}; n_total (= n_past + N of the chunk) stays fixed for the whole chunk.
// Falcon: the fused QKV mat-mul leaves f32 rows [Q heads | K heads | V heads]; rotate Q and K in NEOX mode (pairs
// (i, i + head_dim/2), reference ggml.c:12543-12561; the reference build evaluates out[i] = fma(x0, cos, -(x1*sin)),
// out[i + n/2] = fma(x0, sin, x1*cos) — oracle/mirror.c:mir_rope_neox) and store fp16 Q / K cache / V cache.
// grid = n_head + 2*n_head_kv (one head each), head_dim/2 threads.
__global__ void falcon_rope_store_kernel(const float* __restrict__ qkv, uint16_t* __restrict__ q_f16, uint16_t* __restrict__ kcache,
                                         uint16_t* __restrict__ vcache, const float* __restrict__ rope_cs, const int* __restrict__ pos_p,
                                         int n_head, int n_head_kv, int head_dim, int n_ctx, int v_stride) {
    const int hh = (int)blockIdx.x, i = (int)threadIdx.x, half = head_dim >> 1, pos = *pos_p;
    if (i >= half) return;
    const float* src = qkv + (size_t)hh * head_dim;
    if (hh >= n_head + n_head_kv) {   // V head: plain f32 -> f16 into the transposed cache
        const int row0 = (hh - n_head - n_head_kv) * head_dim;
        vcache[(size_t)(row0 + i) * v_stride + pos] = f32_to_f16_bits(src[i]);
        vcache[(size_t)(row0 + i + half) * v_stride + pos] = f32_to_f16_bits(src[i + half]);
        return;
    }
    const float cs = rope_cs[((size_t)pos * half + i) * 2 + 0], sn = rope_cs[((size_t)pos * half + i) * 2 + 1];
    const float x0 = src[i], x1 = src[i + half];
    const float o0 = fmaf(x0, cs, -(x1 * sn)), o1 = fmaf(x0, sn, x1 * cs);
    if (hh < n_head) {
        q_f16[(size_t)hh * head_dim + i] = f32_to_f16_bits(o0);
        q_f16[(size_t)hh * head_dim + i + half] = f32_to_f16_bits(o1);
    } else {
        const int row0 = (hh - n_head) * head_dim;
        kcache[kcache_off(pos, row0 + i, head_dim, n_ctx)] = f32_to_f16_bits(o0);
        kcache[kcache_off(pos, row0 + i + half, head_dim, n_ctx)] = f32_to_f16_bits(o1);
    }
}

// LayerNorm of the final hidden state (falcon embeddings output): same arithmetic as the PRO_LAYERNORM prologue.
template <int NT>
__global__ void __launch_bounds__(NT) layernorm_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ y, int n, float eps) {
    __shared__ double red[NT / 64];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double s = 0.0;
    for (int i = tid; i < n; i += NT) s += (double)x[i];
    s = wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    double tot = 0.0;
    for (int k = 0; k < NT / 64; ++k) tot += red[k];
    const float mean = (float)(tot / (double)n);
    __syncthreads();
    double s2 = 0.0;
    for (int i = tid; i < n; i += NT) { const float v = x[i] - mean; s2 += (double)(v * v); }
    s2 = wave_sum(s2);
    if (lane == 0) red[wv] = s2;
    __syncthreads();
    double tot2 = 0.0;
    for (int k = 0; k < NT / 64; ++k) tot2 += red[k];
    const float variance = (float)(tot2 / (double)n);
    const float scale = 1.0f / sqrtf(variance + eps);
    for (int i = tid; i < n; i += NT) y[i] = (((x[i] - mean) * scale) * w[i]) + b[i];
}

// GPT-2 attention over the F32 KV cache (reference gpt2.cc:476-585).  Both mat-muls are ggml_vec_dot_f32 (ggml.c:2355-2389,
// AVX2: 4 accumulators x 8 f32 lanes, 32 elements per step, fma; the f16 dot's reduction tree; leftovers fmaf in float) —
// restated per thread with the 32 accumulators in registers (this model family is the plumbing config, not a speed
// target).  grid = n_head, 256 threads.  The workgroup first appends its head's slice of the new K / V rows
// (qkv = [q | k | v] f32 rows from the c_attn launch) to the cache, which only this workgroup reads.
DEV float dot_f32_reduce(const float (&s)[4][8]) {
    float S[8], t0[4];
#pragma unroll
    for (int l = 0; l < 8; ++l) S[l] = (s[0][l] + s[2][l]) + (s[1][l] + s[3][l]);
#pragma unroll
    for (int m = 0; m < 4; ++m) t0[m] = S[m] + S[m + 4];
    return (t0[0] + t0[1]) + (t0[2] + t0[3]);
}
__global__ void __launch_bounds__(256) attn_f32_exact_kernel(const float* __restrict__ qkv, float* __restrict__ kmem,
                                                             float* __restrict__ vmem, float* __restrict__ out,
                                                             const uint16_t* __restrict__ exp_tab, const int* __restrict__ pos_p,
                                                             const int* __restrict__ n_total_p, int n_embd, int head_dim,
                                                             float kq_scale) {
    __shared__ float prob[kMaxCtx];
    __shared__ double red[4];
    __shared__ float redf[4];
    const int h = (int)blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int pos = *pos_p, n_kv = pos + 1, n_tot = *n_total_p, E = n_embd, hd = head_dim;
    for (int i = tid; i < hd; i += 256) {
        kmem[(size_t)pos * E + h * hd + i] = qkv[E + h * hd + i];
        vmem[(size_t)pos * E + h * hd + i] = qkv[2 * E + h * hd + i];
    }
    __syncthreads();
    const float* q = qkv + h * hd;
    float mx = -INFINITY;
    for (int p = tid; p < n_kv; p += 256) {
        const float* k = kmem + (size_t)p * E + h * hd;
        float s[4][8] = {};
        const int np = hd & ~31;
        for (int i = 0; i < np; i += 32)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int l = 0; l < 8; ++l) s[j][l] = fmaf(k[i + 8 * j + l], q[i + 8 * j + l], s[j][l]);
        float sumf = dot_f32_reduce(s);
        for (int i = np; i < hd; ++i) sumf = fmaf(k[i], q[i], sumf);
        const float sc = sumf * kq_scale;
        prob[p] = sc;
        mx = fmaxf(mx, sc);
    }
    mx = wave_max(mx);
    if (lane == 0) redf[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    double sum = 0.0;
    for (int p = tid; p < n_kv; p += 256) {
        const float e = f16_bits_to_f32(exp_tab[f32_to_f16_bits(prob[p] - mx)]);
        prob[p] = e;
        sum += (double)e;   // exact in any order: fp16 values in (0, 1]
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    const double tot = ((red[0] + red[1]) + red[2]) + red[3];
    const float inv = (float)(1.0 / tot);
    for (int p = tid; p < n_kv; p += 256) prob[p] = prob[p] * inv;   // stays f32: V is F32, so no fp16 conversion of P
    for (int p = n_kv + tid; p < n_tot; p += 256) prob[p] = 0.0f;     // masked columns of this batch
    __syncthreads();
    for (int d = tid; d < hd; d += 256) {
        const float* v = vmem + h * hd + d;
        float s[4][8] = {};
        const int np = n_tot & ~31;
        for (int i = 0; i < np; i += 32)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int l = 0; l < 8; ++l) s[j][l] = fmaf(v[(size_t)(i + 8 * j + l) * E], prob[i + 8 * j + l], s[j][l]);
        float sumf = dot_f32_reduce(s);
        for (int i = np; i < n_tot; ++i) sumf = fmaf(v[(size_t)i * E], prob[i], sumf);
        out[h * hd + d] = sumf;
    }
}

// Pipeline-stage hand-off: row `step` of the [n_ctx][E] stage buffer <-> the working residual stream.
// to_rows == 0: dst[i] = src[step*E + i] (stage input); to_rows == 1: dst[step*E + i] = src[i] (stage output).
__global__ void stage_row_kernel(const float* src, float* dst, int E, const int* state, int to_rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    const size_t off = (size_t)state[0] * E;
    if (to_rows) dst[off + i] = src[i];
    else dst[i] = src[off + i];
}

__global__ void advance_state_kernel(int* state) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        state[0] += 1;
        state[1] += 1;
    }
}
