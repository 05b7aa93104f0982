// Q8_0 activation quantization for the 32-element block types (Q8_0, Q4_0 weights x Q8_0 activations), bit-identical to the
// reference's AVX2 build:
//   quantize_row_q8_0 (ggml.c:1208-1300 AVX2 path): per 32: d = amax/127 stored as fp16, id = 127/amax (0 when amax == 0),
//       q = round-half-even(x * id).
// This header keeps the workgroup prologue the prompt-chunk quantizer (kernels_pf.h) builds its token images with; the decode
// mat-vec of these types is generation 9 (kernels_v9.h: step9b / pro9b_*, LAYOUT_L9 records) — the systolic wave-to-wave form
// that lived here (402 tok/s on the 7B Q8_0 model, 536 with generation 9) was removed after the A/B.
// Layout LAYOUT_G4 (engine.cc:upload_matrix; read by the prompt-chunk kernels): per 8-row tile and group of 4 consecutive blocks
// one record, a lane's four dwords (its 4 elements of 4 blocks) one 16-byte load:
//   Q8_0 (1088 B): qs[r][l][i] 4 B at (r*8 + l)*16 + i*4        | d[r][i] fp16 at 1024 + r*8 + i*2
//   Q4_0 ( 576 B): qs[r][l&3][i] 4 B at (r*4 + (l&3))*16 + i*4  | d[r][i] fp16 at  512 + r*8 + i*2
#pragma once
#include "kernels_exact.h"

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
        if (b < ((nblk + 3) & ~3)) {   // blocks past the row's end inside its last group of four: zero quants, y.d = 0 (rows of Falcon-7B: 142 blocks)
            L.q8[((b >> 2) * 8 + l) * 4 + (b & 3)] = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | ((q3 & 0xff) << 24);
            if (l == 0) L.yd[b] = f16_bits_to_f32(f32_to_f16_bits(d));
        }
    }
    __syncthreads();
}
