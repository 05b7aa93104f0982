// Are the f16 matrix-core products of INTEGER-valued operands exact?  (kernels_pg.h relies on it: integer sums below 2^24 must come
// out of v_mfma_f32_16x16x32_f16 / v_mfma_f32_16x16x16_f16 exactly, whatever the internal summation order, also when chained
// through the accumulator.)  Also checks operand pairing and the D map:
//   A lane i: row i & 15, k-group i >> 4 (8 halves; 4 for x16);  B lane i: column i & 15, same k-group;  D lane i, reg j: D[4 * (i >> 4) + j][i & 15]
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_f16_exact tools/experiments/mfma_f16_exact.cpp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
// A[n][kg][8] , B[m][kg][8] integer-valued, C[n][m]; two chained products: D = A0*B0 + (A1*B1 + C)
__global__ void k32(const short* A, const short* B, const float* C, float* D, int nmat) {
    const int i = threadIdx.x;
    for (int t = blockIdx.x; t < nmat; t += gridDim.x) {
        h8 a0, b0, a1, b1;
        for (int e = 0; e < 8; ++e) {
            a0[e] = (_Float16)A[((size_t)(2 * t) * 64 + (i & 15) * 4 + (i >> 4)) * 8 + e];
            b0[e] = (_Float16)B[((size_t)(2 * t) * 64 + (i & 15) * 4 + (i >> 4)) * 8 + e];
            a1[e] = (_Float16)A[((size_t)(2 * t + 1) * 64 + (i & 15) * 4 + (i >> 4)) * 8 + e];
            b1[e] = (_Float16)B[((size_t)(2 * t + 1) * 64 + (i & 15) * 4 + (i >> 4)) * 8 + e];
        }
        f4 c;
        for (int j = 0; j < 4; ++j) c[j] = C[(size_t)t * 256 + (4 * (i >> 4) + j) * 16 + (i & 15)];
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c, 0, 0, 0);
        for (int j = 0; j < 4; ++j) D[(size_t)t * 256 + (4 * (i >> 4) + j) * 16 + (i & 15)] = c[j];
    }
}
__global__ void k16(const short* A, const short* B, float* D, int nmat) {   // A[n][kg][4], B[m][kg][4]
    const int i = threadIdx.x;
    for (int t = blockIdx.x; t < nmat; t += gridDim.x) {
        h4 a, b;
        for (int e = 0; e < 4; ++e) {
            a[e] = (_Float16)A[((size_t)t * 64 + (i & 15) * 4 + (i >> 4)) * 4 + e];
            b[e] = (_Float16)B[((size_t)t * 64 + (i & 15) * 4 + (i >> 4)) * 4 + e];
        }
        f4 c = {0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
        for (int j = 0; j < 4; ++j) D[(size_t)t * 256 + (4 * (i >> 4) + j) * 16 + (i & 15)] = c[j];
    }
}
static int rnd(int lo, int hi) { return lo + rand() % (hi - lo + 1); }
int main() {
    const int NM = 4096;
    srand(11);
    short* hA = (short*)malloc((size_t)2 * NM * 512 * 2); short* hB = (short*)malloc((size_t)2 * NM * 512 * 2);
    float* hC = (float*)malloc((size_t)NM * 256 * 4); float* hD = (float*)malloc((size_t)NM * 256 * 4);
    for (int t = 0; t < 2 * NM; ++t) {
        const int mode = (t / 2) % 4;   // 0: Q4_K/Q5_K-like (b in 0..1953), 1: Q6_K-like even (|b| <= 4096 even), 2: extremes same sign, 3: mixed
        for (int x = 0; x < 512; ++x) {
            int a = rnd(-127, 127), b;
            if (mode == 0) b = rnd(0, 63) * rnd(0, 31);
            else if (mode == 1) b = rnd(-32, 31) * (rnd(-128, 127) & ~1);
            else if (mode == 2) { a = (rand() & 7) ? 127 : rnd(100, 127); b = (t & 1) ? 2048 - (rand() & 1) : 2046 + (rand() & 1) * 2; }
            else { b = (rand() & 1) ? rnd(-32, 31) * (rnd(-128, 127) & ~1) : rnd(-32, 31) * (rand() & 1); }
            hA[(size_t)t * 512 + x] = (short)a; hB[(size_t)t * 512 + x] = (short)b;
        }
    }
    for (size_t x = 0; x < (size_t)NM * 256; ++x) hC[x] = (float)rnd(-4000000, 4000000);
    short *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, (size_t)2 * NM * 1024); hipMalloc(&dB, (size_t)2 * NM * 1024); hipMalloc(&dC, (size_t)NM * 1024); hipMalloc(&dD, (size_t)NM * 1024);
    hipMemcpy(dA, hA, (size_t)2 * NM * 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, (size_t)2 * NM * 1024, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, (size_t)NM * 1024, hipMemcpyHostToDevice);
    k32<<<256, 64>>>(dA, dB, dC, dD, NM);
    hipMemcpy(hD, dD, (size_t)NM * 1024, hipMemcpyDeviceToHost);
    long bad = 0, big = 0; double maxabs = 0;
    for (int t = 0; t < NM; ++t)
        for (int n = 0; n < 16; ++n) for (int m = 0; m < 16; ++m) {
            long long s = (long long)hC[(size_t)t * 256 + n * 16 + m];
            for (int h = 0; h < 2; ++h) for (int k = 0; k < 32; ++k)
                s += (long long)hA[((size_t)(2 * t + h) * 64 + n * 4 + k / 8) * 8 + k % 8] * hB[((size_t)(2 * t + h) * 64 + m * 4 + k / 8) * 8 + k % 8];
            if (s > 16777216 || s < -16777216) { ++big; continue; }   // outside the exactness contract
            if ((double)(s < 0 ? -s : s) > maxabs) maxabs = (double)(s < 0 ? -s : s);
            if ((float)s != hD[(size_t)t * 256 + n * 16 + m]) { if (bad < 5) printf("  x32 mismatch t=%d n=%d m=%d want %lld got %.1f\n", t, n, m, s, hD[(size_t)t * 256 + n * 16 + m]); ++bad; }
        }
    printf("mfma_f32_16x16x32_f16, 2 chained + C, integer operands: %s (%ld mismatches, %ld sums beyond 2^24 skipped, max |sum| %.0f)\n", bad ? "FAIL" : "PASS", bad, big, maxabs);
    // x16: min-term shape: a = 16-element sums (<= 2032), b = 6-bit mins
    for (size_t x = 0; x < (size_t)NM * 256; ++x) { hA[x] = (short)rnd(-2032, 2032); hB[x] = (short)((rand() & 3) ? rnd(0, 63) : 0); }
    hipMemcpy(dA, hA, (size_t)NM * 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, (size_t)NM * 512, hipMemcpyHostToDevice);
    k16<<<256, 64>>>(dA, dB, dD, NM);
    hipMemcpy(hD, dD, (size_t)NM * 1024, hipMemcpyDeviceToHost);
    long bad16 = 0;
    for (int t = 0; t < NM; ++t)
        for (int n = 0; n < 16; ++n) for (int m = 0; m < 16; ++m) {
            long long s = 0;
            for (int k = 0; k < 16; ++k) s += (long long)hA[((size_t)t * 64 + n * 4 + k / 4) * 4 + k % 4] * hB[((size_t)t * 64 + m * 4 + k / 4) * 4 + k % 4];
            if ((float)s != hD[(size_t)t * 256 + n * 16 + m]) ++bad16;
        }
    printf("mfma_f32_16x16x16_f16, integer operands: %s (%ld mismatches)\n", bad16 ? "FAIL" : "PASS", bad16);
    return (bad != 0) | (bad16 != 0);
}
