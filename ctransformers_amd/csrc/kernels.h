// Hand-written CDNA4 (gfx950, wave64) kernels of the decode hot path that replaces ggml_compute_forward's ops
// (reference models/ggml/ggml.c) for the llama / falcon / gpt2 graphs.  This header: shared argument structures and the
// small kernels (embedding row, norms, falcon RoPE + KV store, gpt2 F32 attention, pipeline hand-off, cursor, greedy pick).  The
// decode mat-vec lives in kernels_v9.h (K-quants), the decode attention in kernels_attn9.h, and kernels_q32.h (Q8_0 / Q4_0), the prompt-chunk
// kernels in kernels_pg.h / kernels_pf.h, the fp16 attention kernels and the Q8_K prologues in kernels_exact.h.
//
// Numerics contract: not only the reference's quantization points but the f32
// accumulation ORDER of its AVX2 build is reproduced, so logits are bit-identical to the reference CPU build.
#pragma once
#include <math.h>

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
    int n_groupA;       // items of the first type-homogeneous job group (rest = group B)
    int nwA;            // generation 7, two-type launches: waves 0..nwA-1 of every workgroup walk group A, the rest group B
    const uint8_t* baseA;   // generation 7: first LAYOUT_R2C4 record of type group A / B (the group's jobs are contiguous)
    const uint8_t* baseB;
    int* bump;          // cursor {step, pos, ..}: workgroup 0 advances it by one token at the end of the launch (the last mat-vec of a token step
                        // that does not read it: one launch fewer per token than a separate advance_state_kernel)
    float* emb_out;     // generation 7: block 0 also stores the normalised activation vector here (final-norm output of the ABI)
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
    float* f16_tmp;            // F16 weight matrices (kernels_f16.h): raw results of the launch's rows, between the dot kernel and the epilogue kernel
    // Greedy pick, first half (kernels_v9.h:v9_pick_store; the EMB instantiations): every wave of the head launch leaves the first maximum
    // of ITS logits rows as a 64-bit key in pick_ws[workgroup][wave]; pick_cont_kernel finishes.  Null: no pick.
    unsigned* pick_ws;
    int rope_neox;             // EPI_ROPE_Q / EPI_ROPE_K on a matrix whose rows were permuted at load so that the NEOX pair (i, i + head_dim / 2) of a head sits in rows
                               // (2i, 2i + 1) (falcon, engine_load.h:falcon_permute_rows_kernel): the pair rotates by the NEOX formulas and is stored at its ORIGINAL rows
    int grid_out;              // host side only: workgroups of the launch these arguments went out with (set by the launch helper)
    float* dbg_sink;           // measurement only: always-valid scratch the ablation paths may write to
    int dbg;                   // measurement only (CT_AMD_DBG / ctamd_trace_site): bit 32 = write in-kernel s_memtime stamps to dbg_sink
};

// ------------------------------------------------------------------------------------------------------------------
// Token embedding: dequantize one row of token_embd (file layout) -> f32   (reference ggml.c:11615-11642 get_rows_q)
// ------------------------------------------------------------------------------------------------------------------
// element e of a token_embd row in file layout (every weight type the loaders accept for the table)
DEV float embed_value(const uint8_t* __restrict__ row, int type, int e) {
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
    } else if (type == GT_Q4_1 || type == GT_Q5_0 || type == GT_Q5_1) {   // ggml.c:1538-1610; q * d is exact (5 x 11 bits): x0*d + m rounds once, fused or not
        const int bbk = type == GT_Q4_1 ? 20 : (type == GT_Q5_0 ? 22 : 24), hdr = type == GT_Q5_0 ? 2 : 4;
        const uint8_t* b = row + (size_t)(e >> 5) * bbk;
        const float d = f16_bits_to_f32((uint16_t)(b[0] | (b[1] << 8)));
        const float m = type == GT_Q5_0 ? 0.0f : f16_bits_to_f32((uint16_t)(b[2] | (b[3] << 8)));
        const int j = e & 31;
        const uint8_t* qs = b + hdr + (type == GT_Q4_1 ? 0 : 4);
        int q = (j < 16) ? (qs[j] & 0xF) : (qs[j - 16] >> 4);
        if (type != GT_Q4_1) q |= ((b[hdr + (j >> 3)] >> (j & 7)) & 1) << 4;
        y = type == GT_Q5_0 ? (float)(q - 16) * d : fmaf((float)q, d, m);
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
    return y;
}

__global__ void __launch_bounds__(256) embed_row_kernel(const uint8_t* __restrict__ raw, int type, int K,
                                                        const int* __restrict__ tokens, const int* __restrict__ state,
                                                        float* __restrict__ out, const float* __restrict__ wpe = nullptr) {
    const int token = tokens[state[0] + (int)blockIdx.y];   // blockIdx.y: token inside a prompt chunk (kernels_pf.h), else 0
    const uint8_t* row = raw + (size_t)token * ggml_row_bytes(type, K);
    out += (size_t)blockIdx.y * K;
    for (int e = (int)(blockIdx.x * blockDim.x + threadIdx.x); e < K; e += (int)(gridDim.x * blockDim.x)) {
        float y = embed_value(row, type, e);
        if (wpe) y = y + wpe[(size_t)(state[1] + (int)blockIdx.y) * K + e];   // gpt2: wte[token] + wpe[pos] (gpt2.cc:441-444)
        out[e] = y;
    }
}

constexpr int kMaxCtx = 8192;        // the GPT-2 attention kernel keeps one score / probability row of this length in static LDS
constexpr int kMaxCtxFused = 32768;  // llama / falcon (attn_fused_exact_kernel): the row lives in dynamic LDS, 4 bytes per position

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
// state (the cursor) = {step, pos, n_total, batch}: n_total = n_past + n of the eval and batch = the reference batch size inside it
// (0: one batch) stay fixed for the whole eval; step and pos advance token by token (or chunk by chunk).
// Falcon: the fused QKV mat-mul leaves f32 rows [Q heads | K heads | V heads]; rotate Q and K in NEOX mode (pairs
// (i, i + head_dim/2), reference ggml.c:12543-12561; the reference build evaluates out[i] = fma(x0, cos, -(x1*sin)),
// out[i + n/2] = fma(x0, sin, x1*cos) — oracle/mirror.c:mir_rope_neox) and store fp16 Q / K cache / V cache.
// grid = n_head + 2*n_head_kv (one head each), head_dim/2 threads.
__global__ void falcon_rope_store_kernel(const float* __restrict__ qkv, uint16_t* __restrict__ q_f16, uint16_t* __restrict__ kcache,
                                         uint16_t* __restrict__ vcache, const float* __restrict__ rope_cs, const int* __restrict__ pos_p,
                                         int n_head, int n_head_kv, int head_dim, int n_ctx, int v_stride, int perm) {
    // perm: the Q and K rows arrive in the load-time order of MatvecArgs::rope_neox (pair (i, i + half) in rows (2i, 2i + 1)); V rows are never permuted
    // blockIdx.y: token inside a prompt chunk (rows of n_head + 2 n_head_kv heads in qkv, n_head heads in q_f16), else 0
    const int hh = (int)blockIdx.x, i = (int)threadIdx.x, half = head_dim >> 1, tok = (int)blockIdx.y, pos = *pos_p + tok;
    if (i >= half) return;
    const float* src = qkv + ((size_t)tok * (n_head + 2 * n_head_kv) + hh) * head_dim;
    q_f16 += (size_t)tok * n_head * head_dim;
    if (hh >= n_head + n_head_kv) {   // V head: plain f32 -> f16 into the transposed cache
        const int row0 = (hh - n_head - n_head_kv) * head_dim;
        vcache[(size_t)(row0 + i) * v_stride + pos] = f32_to_f16_bits(src[i]);
        vcache[(size_t)(row0 + i + half) * v_stride + pos] = f32_to_f16_bits(src[i + half]);
        return;
    }
    const float cs = rope_cs[((size_t)pos * half + i) * 2 + 0], sn = rope_cs[((size_t)pos * half + i) * 2 + 1];
    const float x0 = src[perm ? 2 * i : i], x1 = src[perm ? 2 * i + 1 : i + half];
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

// MPT (mpt_eval, models/llms/mpt.cc:404-434): the fused QKV mat-mul leaves f32 rows [Q | K | V] of n_embd each; ggml_clamp
// (MAX(MIN(x, max), min), ggml.c clamp_f32) when clip_qkv > 0, then fp16 Q / K cache / V cache (ggml_cpy f32 -> f16) — no
// rotation: positions enter through the ALiBi term of the attention kernel.  grid = (3 n_head, tokens), head_dim threads.
__global__ void mpt_store_kernel(const float* __restrict__ qkv, uint16_t* __restrict__ q_f16, uint16_t* __restrict__ kcache,
                                 uint16_t* __restrict__ vcache, const int* __restrict__ pos_p, int n_head, int head_dim, int n_ctx,
                                 int v_stride, float clip) {
    const int hh = (int)blockIdx.x, i = (int)threadIdx.x, tok = (int)blockIdx.y, pos = *pos_p + tok, E = n_head * head_dim;
    if (i >= head_dim) return;
    float v = qkv[(size_t)tok * 3 * E + (size_t)hh * head_dim + i];
    if (clip > 0.0f) {
        v = v < clip ? v : clip;
        v = v > -clip ? v : -clip;
    }
    const uint16_t hv = f32_to_f16_bits(v);
    if (hh < n_head) q_f16[(size_t)tok * E + (size_t)hh * head_dim + i] = hv;
    else if (hh < 2 * n_head) kcache[kcache_off(pos, (hh - n_head) * head_dim + i, head_dim, n_ctx)] = hv;
    else vcache[(size_t)((hh - 2 * n_head) * head_dim + i) * v_stride + pos] = hv;
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
// Prompt chunks: blockIdx.y = token inside the chunk (qkv / out rows of 3E / E floats per token); the K / V rows of ALL the
// chunk's tokens are then appended beforehand by gpt2_kv_append_kernel (append = 0 here), because a token attends to the
// earlier tokens of its chunk whose workgroups may not have run yet.
__global__ void gpt2_kv_append_kernel(const float* __restrict__ qkv, float* __restrict__ kmem, float* __restrict__ vmem,
                                      const int* __restrict__ pos_p, int n_embd) {
    const int tok = (int)blockIdx.x, pos = *pos_p + tok, E = n_embd;
    const float* row = qkv + (size_t)tok * 3 * E;
    for (int i = (int)threadIdx.x; i < E; i += (int)blockDim.x) {
        kmem[(size_t)pos * E + i] = row[E + i];
        vmem[(size_t)pos * E + i] = row[2 * E + i];
    }
}
// The scalar tail of ggml_vec_dot_f32 (`for (i = np; i < n; ++i) sumf += x[i]*y[i]`, ggml.c:2355-2389) as the reference BUILD runs it:
// gcc -O3 vectorises the in-order reduction — groups of 8 products by vmulps (not fused) added in order, one group of 4 the same way if
// at least 4 are left, then at most 3 fused scalar steps (disassembly of oracle/_ref; every length 3..31 checked against the exported
// function).  Round 3: the pure-fma tail this replaces differed by an ulp in ~40 % of the dots with 4 or more leftovers, which the
// Q8_0 quantization of the attention output absorbed on every model of the suite — and did not at GPT-2 XL's width.
DEV float dot_f32_tail(const float* __restrict__ x, size_t sx, const float* __restrict__ y, int i, int n, float sumf) {
    for (; n - i >= 8; i += 8) {
        float p[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) p[k] = x[(size_t)(i + k) * sx] * y[i + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) sumf = sumf + p[k];
    }
    if (n - i >= 4) {
        float p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = x[(size_t)(i + k) * sx] * y[i + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) sumf = sumf + p[k];
        i += 4;
    }
    for (; i < n; ++i) sumf = fmaf(x[(size_t)i * sx], y[i], sumf);
    return sumf;
}

__global__ void __launch_bounds__(256) attn_f32_exact_kernel(const float* __restrict__ qkv, float* __restrict__ kmem,
                                                             float* __restrict__ vmem, float* __restrict__ out,
                                                             const uint16_t* __restrict__ exp_tab, const int* __restrict__ pos_p,
                                                             const int* __restrict__ n_total_p, int n_embd, int head_dim,
                                                             float kq_scale, int append = 1) {
    __shared__ float prob[kMaxCtx];
    __shared__ double red[4];
    __shared__ float redf[4];
    const int h = (int)blockIdx.x, tok = (int)blockIdx.y, tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int pos = *pos_p + tok, n_kv = pos + 1, E = n_embd, hd = head_dim;
    int n_tot = *n_total_p;
    {   // the reference batch this token belongs to inside the eval: cursor = {step, pos, n_past + n, batch} (kernels_exact.h)
        const int bs = n_total_p[1];
        if (bs > 0) {
            const int idx = pos_p[-1] + tok, base = *pos_p - pos_p[-1];
            const int end = (idx / bs + 1) * bs, n_eval = n_tot - base;
            n_tot = base + (end < n_eval ? end : n_eval);
        }
    }
    qkv += (size_t)tok * 3 * E;
    out += (size_t)tok * E;
    if (append) {
        for (int i = tid; i < hd; i += 256) {
            kmem[(size_t)pos * E + h * hd + i] = qkv[E + h * hd + i];
            vmem[(size_t)pos * E + h * hd + i] = qkv[2 * E + h * hd + i];
        }
        __syncthreads();
    }
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
        const float sumf = dot_f32_tail(k, 1, q, np, hd, dot_f32_reduce(s));
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
        out[h * hd + d] = dot_f32_tail(v, (size_t)E, prob, np, n_tot, dot_f32_reduce(s));
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

// Greedy pick on the GPU: index of the FIRST maximum of the logits — what the reference's sampler chain returns for top_k = 1 without a
// repetition penalty (llama.cpp llama_sample_top_k: std::partial_sort with `a.logit > b.logit` keeps the first of equal maxima; one
// candidate left, so top_p / temperature / the seeded draw cannot change it).  One workgroup; 4 bytes go to the host instead of the logits.
__global__ void __launch_bounds__(1024) argmax_first_kernel(const float* __restrict__ v, int n, int* __restrict__ out) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const int tid = (int)threadIdx.x;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    // eight loads in flight per thread (a load-compare-branch loop runs at one memory latency per element: 12 us for 32000 logits);
    // a thread's indices ascend, so the strict comparison keeps its first maximum
    for (int i0 = tid; i0 < n; i0 += 8 * 1024) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * 1024; x[u] = i < n ? v[i] : -INFINITY; }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (x[u] > best) { best = x[u]; idx = i0 + u * 1024; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ob = __shfl_xor(best, m);
        const int oi = __shfl_xor(idx, m);
        if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
    }
    if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        *out = idx == 0x7fffffff ? 0 : idx;   // all -inf / NaN: the reference's heap keeps element 0
    }
}

// first maximum: the larger value, on equal values the lower row (argmax_first_kernel's rule; NaN and -inf never win)
DEV void pick_merge(float& bv, int& bi, float ov, int oi) {
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
}
DEV void pick_wave_reduce(float& bv, int& bi) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(bv, m);
        const int oi = __shfl_xor(bi, m);
        pick_merge(bv, bi, ov, oi);
    }
}
// order-preserving 64-bit key of (value, row): a larger key = a larger value, or the same value in a lower row
DEV unsigned long long pick_key(float v, int row) {
    if (!(v > -INFINITY)) return 0ull;                       // -inf / NaN never win (argmax_first_kernel: `x > best` from -inf)
    uint32_t b = v == 0.0f ? 0u : f32_to_bits(v);            // +0 and -0 compare equal: one key
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)b << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)row);
}

// The greedy pick, second half, and the continuation of a greedy chain: ONE workgroup behind the head launch.  Reads the n keys the head
// launch's waves left (v9_pick_store: 16 per workgroup) — 32 KB instead of the n_vocab logits argmax_first_kernel walks, and every load
// of a thread in flight at once — and writes the token to
// device memory AND straight into the pinned host word the caller's sample() reads (no copy node); then, where the token step advanced
// the cursor in its last ffn_down launch (cont_state != null), prepares the NEXT token step of a greedy chain on the device: token id,
// cursor {0, pos + 1, pos + 2, 0}, and the residual-stream row = the picked token's embedding row (embed_row_kernel's arithmetic) — so
// that step needs no host copy and no embedding launch (engine.h: continuation graphs).
__global__ void __launch_bounds__(1024) pick_cont_kernel(const unsigned long long* __restrict__ keys, int n, int* __restrict__ out, int* __restrict__ host,
                                                         int* __restrict__ cont_state, float* __restrict__ cont_x,
                                                         const uint8_t* __restrict__ embd, int type, int K) {
    __shared__ unsigned long long part[16];
    __shared__ int s_tok;
    CT_DYN_SMEM(smem_raw);   // the picked token's embedding row in file layout (<= 4 K bytes)
    const int tid = (int)threadIdx.x;
    unsigned long long best = 0ull;
    for (int i0 = tid; i0 < n; i0 += 4 * 1024) {
        unsigned long long k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + 1024 * u; k[u] = keys[i < n ? i : i0]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) best = k[u] > best ? k[u] : best;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const unsigned long long o = __shfl_xor(best, m); best = o > best ? o : best; }
    if ((tid & 63) == 0) part[tid >> 6] = best;
    __syncthreads();
    if (tid < 64) {   // the sixteen per-wave maxima: one more reduce inside wave 0
        best = part[tid & 15];
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) { const unsigned long long o = __shfl_xor(best, m); best = o > best ? o : best; }
        if (tid == 0) {
            const int tok = best ? (int)(0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFull)) : 0;   // all -inf / NaN: the reference's heap keeps element 0
            s_tok = tok;
            *out = tok;
            if (host) *host = tok;
            if (cont_state) {
                cont_state[0] = 0;
                cont_state[2] = cont_state[1] + 1;
                cont_state[3] = 0;
                cont_state[4] = tok;
            }
        }
    }
    if (!cont_x) return;   // (kernel argument: uniform)
    __syncthreads();
    // The row travels to LDS in 4-byte pieces, all in flight at once (2.3 KB for a Q4_K row of 4096: dequantizing straight from global
    // memory is some sixty byte-wide loads per thread through ONE CU's memory pipe: 5 us, measured), then embed_row_kernel's arithmetic
    // reads the LDS copy.
    const size_t rb = ggml_row_bytes(type, K);
    const uint8_t* __restrict__ row = embd + (size_t)s_tok * rb;
    uint8_t* lrow = smem_raw;
    const int words = (int)(rb >> 2);
    if ((((uintptr_t)row) & 3u) == 0) {
        for (int w = tid; w < words; w += 1024) ((uint32_t*)lrow)[w] = ((const uint32_t*)row)[w];
        for (int b = 4 * words + tid; b < (int)rb; b += 1024) lrow[b] = row[b];
    } else {
        for (int b = tid; b < (int)rb; b += 1024) lrow[b] = row[b];
    }
    __syncthreads();
    for (int e = tid; e < K; e += 1024) cont_x[e] = embed_value(lrow, type, e);
}

// measurement only (CT_AMD_STAMPS=1): the 100 MHz wall clock at this point of the stream, appended to buf[1 + buf[0]++]
__global__ void stamp_kernel(unsigned long long* buf, unsigned long long tag) {
#ifndef CT_EMU
    const unsigned long long i = buf[0];
    if (i < 40000) { buf[1 + i] = (wall_clock64() << 4) | tag; buf[0] = i + 1; }
#endif
}

__global__ void advance_state_kernel(int* state, int n_ctx) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        state[0] += 1;
        state[1] += 1;
        state[4 + n_ctx] += 1;   // the token epoch behind the token ids (kernels_qa9.h: tags of the in-launch exchange)
    }
}
