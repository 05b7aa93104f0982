// Round 6 probe for the order-free prompt kernels (kernels_mm8.h): operand pairing and result map of
//   v_mfma_i32_32x32x32_i8  (16 int8 per lane and operand, 16 int32 results per lane)
//   v_mfma_f32_32x32x16_f16 (8 halves per lane and operand)
// the float-addend form (C = 0x4B400000), and the issue rate of the inner-loop shapes the kernels use:
//   pure    back-to-back matrix instructions on two accumulator sets
//   mad     one matrix instruction + 16 v_mad_i32_i24 on its results (K-quants: acc += scale * p)
//   fma     one matrix instruction + 32 v_fma_f32 on its results (Q8_0: acc += (p * d_w) * d_a)
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mm8_probe tools/experiments/mfma_i8_32x32x32_probe.cpp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int mad24(int a, int b, int c) { int r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

__global__ void k_i8(const int8_t* A, const int8_t* B, int* D, int bias) {   // A[n][c][16], B[m][c][16]: n, m < 32, c < 2
    const int l = threadIdx.x;
    const i32x4 a = ((const i32x4*)A)[(l & 31) * 2 + (l >> 5)], b = ((const i32x4*)B)[(l & 31) * 2 + (l >> 5)];
    i32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = bias;
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[i];
}
__global__ void k_f16(const _Float16* A, const _Float16* B, float* D) {   // A[n][c][8], B[m][c][8]
    const int l = threadIdx.x;
    const f16x8 a = ((const f16x8*)A)[(l & 31) * 2 + (l >> 5)], b = ((const f16x8*)B)[(l & 31) * 2 + (l >> 5)];
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.0f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[i];
}

template <int MODE> __global__ void __launch_bounds__(512, 2) k_rate(int iters, int* out, unsigned long long* cyc) {
    const int l = threadIdx.x & 63;
    i32x4 a = {l, l + 1, l + 2, l + 3}, b = {l * 3, l * 5, l * 7, l * 9};
    i32x16 p0, p1, acc0, acc1;
    f32x16 f0, f1;
    for (int i = 0; i < 16; ++i) { p0[i] = 0; p1[i] = 0; acc0[i] = 0; acc1[i] = 0; f0[i] = 0.f; f1[i] = 0.f; }
    int sc = l & 63;
    const float dw = 0.5f + l, da = 0.25f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            p0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, p1, 0, 0, 0);
        } else if (MODE == 1) {
            const i32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            p0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, z, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc1[i] = mad24(p1[i], sc, acc1[i]);
            p1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, z, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc0[i] = mad24(p0[i], sc, acc0[i]);
            a[0] += 1; sc ^= 5;
        } else {
            i32x16 z;
            for (int i = 0; i < 16; ++i) z[i] = 0x4B400000;
            p0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, z, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) f1[i] = __builtin_fmaf(__builtin_fmaf(__builtin_bit_cast(float, p1[i]), dw, -12582912.f * dw), da, f1[i]);
            p1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, z, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) f0[i] = __builtin_fmaf(__builtin_fmaf(__builtin_bit_cast(float, p0[i]), dw, -12582912.f * dw), da, f0[i]);
            a[0] += 1;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < 16; ++i) s += p0[i] + p1[i] + acc0[i] + acc1[i] + (int)f0[i] + (int)f1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}

int main() {
    srand(11);
    int bad_total = 0;
    {
        std::vector<int8_t> hA(32 * 32), hB(32 * 32);
        for (auto& v : hA) v = (int8_t)(rand() % 255 - 127);
        for (auto& v : hB) v = (int8_t)(rand() % 255 - 127);
        int8_t *dA, *dB; int* dD; std::vector<int> hD(1024);
        hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
        hipMemcpy(dA, hA.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), 1024, hipMemcpyHostToDevice);
        for (int bias : {0, 0x4B400000}) {
            k_i8<<<1, 64>>>(dA, dB, dD, bias);
            hipMemcpy(hD.data(), dD, 4096, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int n = 0; n < 32; ++n) for (int m = 0; m < 32; ++m) {
                int s = 0;
                for (int kk = 0; kk < 32; ++kk) s += (int)hA[n * 32 + kk] * (int)hB[m * 32 + kk];
                if (bias) { float f; int v = hD[n * 32 + m]; memcpy(&f, &v, 4); if (f - 12582912.f != (float)s) ++bad; }
                else if (s != hD[n * 32 + m]) ++bad;
            }
            printf("mfma_i32_32x32x32_i8 pairing + D map (C = 0x%x): %s (%d mismatches)\n", bias, bad ? "FAIL" : "PASS", bad);
            bad_total += bad;
        }
    }
    {
        std::vector<_Float16> hA(32 * 16), hB(32 * 16);
        for (auto& v : hA) v = (_Float16)(float)(rand() % 4097 - 2048);
        for (auto& v : hB) v = (_Float16)(float)(rand() % 64);
        _Float16 *dA, *dB; float* dD; std::vector<float> hD(1024);
        hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
        hipMemcpy(dA, hA.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), 1024, hipMemcpyHostToDevice);
        k_f16<<<1, 64>>>(dA, dB, dD);
        hipMemcpy(hD.data(), dD, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int n = 0; n < 32; ++n) for (int m = 0; m < 32; ++m) {
            long s = 0;
            for (int kk = 0; kk < 16; ++kk) s += (long)(float)hA[n * 16 + kk] * (long)(float)hB[m * 16 + kk];
            if ((float)s != hD[n * 32 + m]) ++bad;
        }
        printf("mfma_f32_32x32x16_f16 pairing + D map, exact integer sums: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
        bad_total += bad;
    }
    int* dout; unsigned long long* dcyc;
    hipMalloc(&dout, 2048 * 512 * 4); hipMalloc(&dcyc, 64);
    const int iters = 20000;
    for (int wpb : {256, 512}) {
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k_rate<0><<<256, wpb>>>(iters, dout, dcyc);
                else if (mode == 1) k_rate<1><<<256, wpb>>>(iters, dout, dcyc);
                else k_rate<2><<<256, wpb>>>(iters, dout, dcyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c[4]; hipMemcpy(c, dcyc, 32, hipMemcpyDeviceToHost);
            const double mf = 2.0 * iters * (wpb / 64) * 256;   // matrix instructions of the launch
            printf("rate mode %d (%s), %d waves per SIMD: %.1f cycles (s_memtime ticks) per matrix instruction and wave, %.3f ms, %.0f TOP/s\n", mode,
                   mode == 0 ? "pure" : (mode == 1 ? "mfma + 16 mad24" : "mfma + 32 fma"), wpb / 256, (double)c[mode] / (2.0 * iters), ms, mf * 65536.0 / (ms * 1e-3) / 1e12);
        }
    }
    return bad_total != 0;
}
