// F16 weight matrices (GGUF files of ftype F16 — llama, falcon; legacy gpt2 / starcoder / mpt files of ftype 1, what the reference's convert
// scripts write; the F16 output.weight the reference's quantizer falls back to for rows that are not whole 256-blocks, llama.cpp:4866-4869): token steps only, bit-identical to the reference CPU build.
//
// Reference: ggml_compute_forward_mul_mat with an F16 src0 (ggml.c:11031-11245): vec_dot_type is F16, so the activation row is
// converted with ggml_fp32_to_fp16_row (round to nearest even, F16C) and every output is ggml_vec_dot_f16 (ggml.c:2392-2425) of the
// weight row and that fp16 vector — the dot product of the attention kernels (kernels_exact.h header): 4 accumulator vectors x 8
// lanes, one fma per element in 32-element steps, the AVX reduce tree.  Rows here are whole 32-element steps (checked at load): no
// scalar tail.
//   matvec_f16_kernel    prologue per workgroup: (RMSNorm * w | LayerNorm * w + b ->) fp16 activation vector in LDS; a quad of lanes per output row
//                        (lane j = accumulator vector j: the 16-byte chunks j, j + 4, ... of the row), eight requests in flight per
//                        lane, every request unconditional (a clamped row instead of a branch: kernels_attn9.h); raw f32 results
//   f16_epilogue_kernel  the decode kernels' epilogues (kernels_v9.h) on those results: store / + residual / RoPE -> fp16 Q /
//                        RoPE -> K cache / V cache / SiLU(gate) * up / GELU / + two residuals / row bias forms
// Two launches per site instead of one fused kernel: F16 files are not on any BASELINE config; what matters here is that they load
// and give the reference's bits (13.5 GB per 7B token: bandwidth-bound at a few hundred tokens/s either way).
#pragma once
#include "kernels_exact.h"

// F32W: the matrix is F32 (vec_dot_type F32: the activation row stays f32, ggml_vec_dot_f32 ggml.c:2355-2389 — the same 32-element steps,
// four accumulator vectors and reduce tree on f32 operands); otherwise F16.
template <int NT, bool F32W>
__global__ void __launch_bounds__(NT) matvec_f16_kernel(const float* __restrict__ x, const float* __restrict__ nw, const float* __restrict__ nbias, int K, int pro,
                                                        float eps, const void* __restrict__ Wv, int M, float* __restrict__ out) {
    using WT = typename std::conditional<F32W, float, uint16_t>::type;
    const WT* __restrict__ W = reinterpret_cast<const WT*>(Wv);
    CT_DYN_SMEM(smem_raw);   // the activation vector: K halves (F32 matrices: K floats)
    WT* xh = reinterpret_cast<WT*>(smem_raw);
    __shared__ double red[2][NT / 64];
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = wave_id(), j = tid & 3, quad = tid >> 2;
    // ---- prologue: ggml.c:10700-10716 (rms_norm: double sum, f32 mean, 1 / sqrtf) or ggml.c:10605-10654 (norm: f32 mean of a double sum, the centred
    // values' squares summed in double, f32 variance), ggml_mul with the norm weight (ggml_add with its bias), then fp16 ----
    float scale = 1.0f, mean = 0.0f;
    if (pro == PRO_LAYERNORM) {
        double s1 = 0.0;
        for (int i = tid; i < K; i += NT) s1 += (double)x[i];
        s1 = wave_sum(s1);
        if (lane == 0) red[0][wv] = s1;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) tot += red[0][w];
        mean = (float)(tot / (double)K);
        double s2 = 0.0;
        for (int i = tid; i < K; i += NT) { const float v = x[i] - mean; s2 += (double)(v * v); }
        s2 = wave_sum(s2);
        if (lane == 0) red[1][wv] = s2;
        __syncthreads();
        double tot2 = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) tot2 += red[1][w];
        const float variance = (float)(tot2 / (double)K);
        scale = 1.0f / sqrtf(variance + eps);
    } else if (pro == PRO_RMSNORM) {
        double s = 0.0;
        for (int i = tid; i < K; i += NT) { const float v = x[i]; s += (double)(v * v); }
        s = wave_sum(s);
        if (lane == 0) red[0][wv] = s;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) tot += red[0][w];
        const float ms = (float)(tot / (double)K);
        scale = 1.0f / sqrtf(ms + eps);
    }
    for (int i = tid; i < K; i += NT) {
        float v = x[i];
        if (pro == PRO_LAYERNORM) { v = ((v - mean) * scale) * nw[i]; if (nbias) v += nbias[i]; }
        else if (pro == PRO_RMSNORM) v = (v * scale) * nw[i];
        if constexpr (F32W) xh[i] = v;
        else xh[i] = f32_to_f16_bits(v);
    }
    __syncthreads();
    // ---- rows: a quad per row, NT / 4 rows per pass of the workgroup ----
    constexpr int PB = 8;   // 32-element steps in flight per lane
    constexpr int NV = F32W ? 2 : 1;   // 16-byte requests per step and lane (eight elements)
    const int steps = K >> 5;
    for (int row0 = (int)blockIdx.x * (NT / 4); row0 < M; row0 += (int)gridDim.x * (NT / 4)) {
        const int row = row0 + quad;
        const WT* wrow = W + (size_t)(row < M ? row : M - 1) * K + 8 * j;   // a quad past the last row re-reads it (nothing stored)
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        u32x4 buf[PB][NV];
        auto request = [&](int u, int st) {
            const WT* p = wrow + 32 * (st < steps ? st : steps - 1);
#pragma unroll
            for (int v = 0; v < NV; ++v) buf[u][v] = ld16(p + 4 * v * (F32W ? 1 : 2));
        };
        auto consume = [&](int u, int st) {
            float wf[8], xf[8];
            if constexpr (F32W) {
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const u32x4 xv = *(const u32x4*)(xh + 32 * st + 8 * j + 4 * v);
#pragma unroll
                    for (int l = 0; l < 4; ++l) { wf[4 * v + l] = bits_to_f32(buf[u][v][l]); xf[4 * v + l] = bits_to_f32(xv[l]); }
                }
            } else {
                unpack8_f16(buf[u][0], wf);
                unpack8_f16(*(const u32x4*)(xh + 32 * st + 8 * j), xf);
            }
#pragma unroll
            for (int l = 0; l < 8; ++l) acc[l] = fmaf(wf[l], xf[l], acc[l]);
        };
#pragma unroll
        for (int u = 0; u < PB; ++u) { request(u, u); sched_fence(); }   // issued in slot order: the loop's waits are counted against this order too
        int s0 = 0;
        for (; s0 + PB < steps; s0 += PB) {   // every step of this round exists; each slot is re-requested (clamped to the last step)
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                consume(u, s0 + u);
                request(u, s0 + u + PB);
                sched_fence();   // slot by slot: without it the scheduler gathers the round's waits at its top and the re-requests at its end
            }
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {   // the last round requests nothing
            if (s0 + u < steps) consume(u, s0 + u);
        }
        const float res = f16dot_reduce_exact(acc, j);
        if (j == 0 && row < M) out[row] = res;
    }
}

// Row r of the launch's concatenated outputs (job 0's rows, then job 1's, ...; gate/up: rows [0, F) = gate, [F, 2F) = up).
__global__ void __launch_bounds__(256) f16_epilogue_kernel(const MatvecArgs a, const float* __restrict__ tmp, int n_rows) {
    const int r = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r >= n_rows) return;
    if (a.gateup) {   // SiLU(gate) * up through the fp16 table (kernels_v9.h epilogue)
        const int F = a.job[0].w.M;
        if (r < F) a.out[r] = f16_bits_to_f32(a.silu_tab[f32_to_f16_bits(tmp[r])]) * tmp[F + r];
        return;
    }
    int jb = 0, row = r;
    if (a.njobs > 1 && row >= a.job[0].w.M) { row -= a.job[0].w.M; jb = 1; }
    if (a.njobs > 2 && jb == 1 && row >= a.job[1].w.M) { row -= a.job[1].w.M; jb = 2; }
    const int epi = jb == 2 ? a.job[2].epi : (jb == 1 ? a.job[1].epi : a.job[0].epi);
    const float res = tmp[r];
    if (epi == EPI_STORE) {
        a.out[row] = res;
    } else if (epi == EPI_ADD) {
        a.out[row] = res + a.res[row];
    } else if (epi == EPI_GELU) {   // the single-job epilogues of the falcon / gpt2 / mpt graphs, as kernels_v9.h writes them
        a.out[row] = f16_bits_to_f32(a.gelu_tab[f32_to_f16_bits(res)]);
    } else if (epi == EPI_ADD2) {
        a.out[row] = (res + a.res[row]) + a.res2[row];
    } else if (epi == EPI_BIAS_STORE) {
        a.out[row] = a.bias[row] + res;
    } else if (epi == EPI_BIAS_ADD) {
        a.out[row] = (a.bias[row] + res) + a.res[row];
    } else if (epi == EPI_BIAS_GELU) {
        a.out[row] = f16_bits_to_f32(a.gelu_tab[f32_to_f16_bits(a.bias[row] + res)]);
    } else if (epi == EPI_V) {
        a.vcache[(size_t)row * a.v_stride + *a.pos] = f32_to_f16_bits(res);
    } else if (epi == EPI_ROPE_Q || epi == EPI_ROPE_K) {   // normal-mode RoPE (ggml.c:12522-12539, the build's fma forms): rows 2i, 2i + 1 rotate together
        const int pos = *a.pos;
        const float other = tmp[r ^ 1];
        const int ip = (row % a.head_dim) >> 1;
        const float cs = a.rope_cs[((size_t)pos * (a.head_dim >> 1) + ip) * 2 + 0];
        const float sn = a.rope_cs[((size_t)pos * (a.head_dim >> 1) + ip) * 2 + 1];
        const float o = (row & 1) ? fmaf(res, cs, other * sn) : fmaf(res, cs, -(other * sn));
        if (epi == EPI_ROPE_Q) a.q_f16[row] = f32_to_f16_bits(o);
        else a.kcache[kcache_off(pos, row, a.head_dim, a.n_ctx)] = f32_to_f16_bits(o);
    }
}
