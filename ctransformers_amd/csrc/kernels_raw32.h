// Q4_1 / Q5_0 / Q5_1 weight matrices (llama / falcon GGUF files of those ftypes: llama.cpp:4785-4790 quantizes every 2-D tensor but
// output.weight to the base type; legacy GGML files of gpt2 / starcoder / mpt with ftype 3 / 8 / 9): token steps only, on the FILE layout (DevMat::raw, LAYOUT_RAW32), bit-identical to the reference CPU build.
//
// Reference: ggml_compute_forward_mul_mat (ggml.c:11031-11245) quantizes the activation row to the weight type's vec_dot_type —
// Q8_0 for Q5_0, Q8_1 for Q4_1 / Q5_1 (type traits ggml.c:1700-1745) — and every output is one of
//   ggml_vec_dot_q4_1_q8_1  ggml.c:2699  (AVX2 :2770-2799)   acc[l] = fma(fp16(x.d) * y.d, (float)sumi[l], acc[l]);  summs = fma(fp16(x.m), y.s, summs)
//   ggml_vec_dot_q5_0_q8_0  ggml.c:2825  (AVX2 :2984-3006)   acc[l] = fma(fp16(x.d) * fp16(y.d), (float)sumi[l], acc[l])
//   ggml_vec_dot_q5_1_q8_1  ggml.c:3065  (AVX2 :3234-3259)   as Q4_1 with the fifth bit
// block after block; sumi[l] = the four products of elements 4l .. 4l+3 (element e < 16: low nibble of qs[e], e >= 16: high nibble of
// qs[e - 16]; fifth bit = bit e of qh; Q5_0 values are (q5 - 16)); result = hsum_float_8(acc) (+ summs).  `summs += m * s` is one
// fused multiply-add in the reference build (oracle/mirror.c; pinned against oracle/_ref).
// Activation blocks (AVX2 quantizers ggml.c:1208-1300, :1420-1476): d = amax / 127 (Q8_0: stored as fp16; Q8_1: f32), id = 127 / amax
// (0 for an all-zero block), q = round-half-even(x * id), Q8_1 s = d * (float)(sum of the quants).
//
//   matvec_raw32_kernel  prologue per workgroup: (RMSNorm * w | LayerNorm * w + b ->) activation blocks in LDS; 16 lanes per output row (AVX lane pair x
//                        block column: four blocks per round, chained through the quad), a ring of 32 blocks in flight per lane, every request
//                        unconditional (clamped block index); raw f32 results
//   f16_epilogue_kernel  (kernels_f16.h) the decode epilogues on those results
// These ftypes are on no BASELINE config: the point is that such files load and give the reference's bits; the layout-specific
// generations (kernels_v9.h) are what the measured types run on.
#pragma once
#include "kernels_f16.h"

template <int TYPE> DEV constexpr int raw32_block_bytes() { return TYPE == GT_Q4_1 ? 20 : (TYPE == GT_Q5_0 ? 22 : 24); }

struct Raw32Blk { uint32_t qs, qh, dm; };   // the lane's nibble word, the block's fifth bits, d | m << 16

template <int TYPE> DEV Raw32Blk raw32_load(const uint8_t* b, int l) {
    Raw32Blk R;
    if constexpr (TYPE == GT_Q4_1) {          // d m | qs[16]: 4-byte aligned (20-byte blocks, rows of whole blocks, 32-byte aligned tensors)
        R.dm = *(const uint32_t*)b;
        R.qh = 0;
        R.qs = *(const uint32_t*)(b + 4 + 4 * (l & 3));
    } else if constexpr (TYPE == GT_Q5_1) {   // d m | qh | qs[16]: 4-byte aligned
        R.dm = *(const uint32_t*)b;
        R.qh = *(const uint32_t*)(b + 4);
        R.qs = *(const uint32_t*)(b + 8 + 4 * (l & 3));
    } else {                                  // Q5_0: d | qh | qs[16]: 22-byte blocks are only 2-byte aligned
        const uint16_t* h = (const uint16_t*)b;
        R.dm = h[0];
        R.qh = (uint32_t)h[1] | ((uint32_t)h[2] << 16);
        const uint16_t* q = h + 3 + 2 * (l & 3);
        R.qs = (uint32_t)q[0] | ((uint32_t)q[1] << 16);
    }
    return R;
}

template <int TYPE, int NT>
__global__ void __launch_bounds__(NT) matvec_raw32_kernel(const float* __restrict__ x, const float* __restrict__ nw, const float* __restrict__ nbias,
                                                          int K, int pro, float eps, const uint8_t* __restrict__ W, int M, float* __restrict__ out) {
    constexpr bool Q81 = TYPE != GT_Q5_0;   // activation blocks are Q8_1
    constexpr int BB = raw32_block_bytes<TYPE>();
    CT_DYN_SMEM(smem_raw);   // K / 4 quant words | K / 32 block scales d | K / 32 block sums s
    int* aq = reinterpret_cast<int*>(smem_raw);
    const int nb = K >> 5;
    float* ad = reinterpret_cast<float*>(aq + (K >> 2));
    float* as = ad + nb;
    __shared__ double red[2][NT / 64];
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = wave_id(), l = tid & 7;
    // ---- prologue: ggml.c:10700-10716 (rms_norm: double sum, f32 mean, 1 / sqrtf) or ggml.c:10605-10654 (norm: f32 mean of a double sum, the
    // centred values' squares summed in double, f32 variance), ggml_mul with the norm weight (ggml_add with its bias), then the activation blocks ----
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
    for (int b0 = 0; b0 < nb; b0 += NT / 8) {   // eight lanes per block (lane l: elements 4l .. 4l+3); whole groups of eight lanes stay together
        const int b = b0 + (tid >> 3);
        const int bc = b < nb ? b : nb - 1;
        float4 t = *(const float4*)(x + bc * 32 + l * 4);
        if (pro != PRO_PLAIN) {
            const float4 w4 = *(const float4*)(nw + bc * 32 + l * 4);
            if (pro == PRO_LAYERNORM) { t.x -= mean; t.y -= mean; t.z -= mean; t.w -= mean; }
            t.x = (t.x * scale) * w4.x; t.y = (t.y * scale) * w4.y; t.z = (t.z * scale) * w4.z; t.w = (t.w * scale) * w4.w;
            if (pro == PRO_LAYERNORM && nbias) {
                const float4 b4 = *(const float4*)(nbias + bc * 32 + l * 4);
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
        int qsum = (q0 + q1) + (q2 + q3);
        qsum += lane_xor1(qsum);
        qsum += lane_xor2(qsum);
        qsum += lane_xor4(qsum);
        if (b < nb) {
            aq[b * 8 + l] = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | ((q3 & 0xff) << 24);
            if (l == 0) {
                ad[b] = Q81 ? d : f16_bits_to_f32(f32_to_f16_bits(d));
                as[b] = d * (float)qsum;
            }
        }
    }
    __syncthreads();
    // ---- rows: 16 lanes per row — lane = (row of the wave's four, AVX lane pair lq | lq + 4, block column c): a round is four consecutive
    // blocks, one per column; a lane's nibble word serves both AVX lanes of its pair (low / high nibbles: elements 4 lq .. and 16 + 4 lq ..);
    // the quad's lanes c = 0..3 run the reference's chains over the round in order (quad_chain4: gpu.h).  A column past the row's end
    // contributes fma(0, 0, acc) = acc (acc is never -0: it starts at +0).  Eight lanes per row (no columns) left a 4096-row site at two
    // waves per CU: 0.45 TB/s; 32 lanes per row (one AVX lane per lane) twice the load instructions per byte: 0.9 TB/s. ----
    constexpr int PB = 8;   // rounds (of four blocks) in flight per lane: a register ring, every slot re-requested as soon as it is consumed
    const int c = lane & 3, lq = (lane >> 2) & 3;
    constexpr int RPW = NT / 16;   // rows per pass of the workgroup
    for (int row0 = (int)blockIdx.x * RPW; row0 < M; row0 += (int)gridDim.x * RPW) {
        const int row = row0 + (tid >> 4);
        const uint8_t* wrow = W + (size_t)(row < M ? row : M - 1) * nb * BB;   // lanes past the last row re-read it (nothing stored)
        float accl = 0.0f, acch = 0.0f, summs = 0.0f;
        Raw32Blk R[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {   // issued in slot order (the fence): the loop's waits are counted against this order too
            const int b = 4 * u + c;
            R[u] = raw32_load<TYPE>(wrow + (size_t)(b < nb ? b : nb - 1) * BB, lq);
            sched_fence();
        }
        for (int b0 = 0; b0 < nb; b0 += 4 * PB) {   // no branch inside: hipcc counts the requests in flight (s_waitcnt vmcnt(N), kernels_attn9.h)
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int b = b0 + 4 * u + c;
                const bool live = b < nb;
                const int bc = live ? b : nb - 1;
                const int al = aq[bc * 8 + lq], ah = aq[bc * 8 + 4 + lq];
                const float adv = ad[bc], asv = Q81 ? as[bc] : 0.0f;   // unconditional (clamped index): a `live ? ad[bc] : 0` becomes a branch around the read
                int wl = (int)(R[u].qs & 0x0F0F0F0Fu), wh = (int)((R[u].qs >> 4) & 0x0F0F0F0Fu);
                int il, ih;
                if constexpr (TYPE == GT_Q4_1) {
                    il = sdot4(wl, al, 0); ih = sdot4(wh, ah, 0);
                } else {
                    const uint32_t hl = (R[u].qh >> (4 * lq)) & 0xFu, hh = (R[u].qh >> (16 + 4 * lq)) & 0xFu;
                    wl |= (int)(((hl * 0x00204081u) & 0x01010101u) << 4);   // bit k of the four -> bit 4 of byte k
                    wh |= (int)(((hh * 0x00204081u) & 0x01010101u) << 4);
                    if constexpr (TYPE == GT_Q5_1) { il = sdot4(wl, al, 0); ih = sdot4(wh, ah, 0); }
                    else {   // (q5 - 16) . a = q5 . a - 16 * (sum of a)
                        il = sdot4(wl, al, sdot4((int)0xF0F0F0F0u, al, 0));
                        ih = sdot4(wh, ah, sdot4((int)0xF0F0F0F0u, ah, 0));
                    }
                }
                const float dx = f16_bits_to_f32((uint16_t)(R[u].dm & 0xFFFFu));
                const float mx = f16_bits_to_f32((uint16_t)(R[u].dm >> 16));
                const int bn = b + 4 * PB;
                R[u] = raw32_load<TYPE>(wrow + (size_t)(bn < nb ? bn : nb - 1) * BB, lq);
                const float dv = live ? dx * adv : 0.0f;
                accl = quad_chain4(accl, dv, live ? (float)il : 0.0f);
                acch = quad_chain4(acch, dv, live ? (float)ih : 0.0f);
                if constexpr (Q81) summs = quad_chain4(summs, live ? mx : 0.0f, live ? asv : 0.0f);
                sched_fence();   // slot by slot: without it the scheduler hoists every slot's first use to the top of the round (one wait for all eight)
            }
        }
        // hsum_float_8 (ggml.c:609-615): ((x0 + x4) + (x2 + x6)) + ((x1 + x5) + (x3 + x7)); the pair (lq, lq + 4) is in this lane, lq at lane bits 2..3
        float r = acch + accl;
        r = r + lane_xor8(r);
        r = r + lane_xor4(r);
        if constexpr (Q81) r = r + summs;
        if ((lane & 15) == 0 && row < M) out[row] = r;
    }
}
