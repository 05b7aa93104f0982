// Checks the operand layout assumed by kernels_pfm.h for v_mfma_i32_16x16x32_i8 on gfx950:
//   A: lane i holds A[i & 15][8 * (i >> 4) .. + 7] (8 int8 in one 64-bit operand), B: lane i holds B[8 * (i >> 4) .. + 7][i & 15],
//   D: lane i, register j holds D[4 * (i >> 4) + j][i & 15].
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_i8_layout tools/experiments/mfma_i8_layout.cpp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int8_t* A, const int8_t* B, int* D) {
    const int i = threadIdx.x;
    long a = 0, b = 0;
    for (int e = 0; e < 8; ++e) {
        a |= (long)(uint8_t)A[(i & 15) * 32 + 8 * (i >> 4) + e] << (8 * e);
        b |= (long)(uint8_t)B[(8 * (i >> 4) + e) * 16 + (i & 15)] << (8 * e);
    }
    i32x4 c = {1, 2, 3, 4};
    c = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) D[(4 * (i >> 4) + j) * 16 + (i & 15)] = c[j] - (j + 1);
}
// 16x16x64 (gfx950): only the property kernels_pfm.h relies on — byte e of lane (n, q)'s A operand meets byte e of lane
// (m, q)'s B operand, whatever k it is called, and D keeps the 16x16 map: D[n][m] = sum_{q<4, e<16} A(n,q)[e] * B(m,q)[e].
typedef int i32x4v __attribute__((ext_vector_type(4)));
__global__ void k64(const int8_t* A, const int8_t* B, int* D) {   // A[n][q][16], B[m][q][16]
    const int i = threadIdx.x;
    i32x4v a, b;
    for (int w = 0; w < 4; ++w) {
        a[w] = ((const int*)A)[((i & 15) * 4 + (i >> 4)) * 4 + w];
        b[w] = ((const int*)B)[((i & 15) * 4 + (i >> 4)) * 4 + w];
    }
    i32x4v c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) D[(4 * (i >> 4) + j) * 16 + (i & 15)] = c[j];
}
static int test64() {
    int8_t hA[16 * 64], hB[16 * 64];
    for (auto& v : hA) v = (int8_t)(rand() % 255 - 127);
    for (auto& v : hB) v = (int8_t)(rand() % 255 - 127);
    int8_t *dA, *dB; int* dD; int hD[256];
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    k64<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int n = 0; n < 16; ++n) for (int m = 0; m < 16; ++m) {
        int s = 0;
        for (int kk = 0; kk < 64; ++kk) s += (int)hA[n * 64 + kk] * (int)hB[m * 64 + kk];
        if (s != hD[n * 16 + m]) ++bad;
    }
    printf("mfma_i32_16x16x64_i8 pairing + D map: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    return bad;
}
int main() {
    int8_t hA[16 * 32], hB[32 * 16];
    srand(7);
    for (auto& v : hA) v = (int8_t)(rand() % 255 - 127);
    for (auto& v : hB) v = (int8_t)(rand() % 255 - 127);
    int8_t *dA, *dB; int* dD; int hD[256];
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        int s = 0;
        for (int kk = 0; kk < 32; ++kk) s += (int)hA[m * 32 + kk] * (int)hB[kk * 16 + n];
        if (s != hD[m * 16 + n]) ++bad;
    }
    printf("mfma_i32_16x16x32_i8 layout: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    return (bad != 0) | (test64() != 0);
}
