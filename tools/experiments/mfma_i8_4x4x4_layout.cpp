// Probe: operand / result layout of v_mfma_i32_4x4x4_16b_i8 on gfx950 and exactness of the "float addend" trick.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_i8_probe tools/experiments/mfma_i8_4x4x4_layout.cpp && /tmp/mfma_i8_probe
// Hypothesis H1: lane = 4*b + m; A dword of lane = A_b[row m][k = byte 0..3]; B dword of lane = B_b[k = byte][col m];
//                result register i of lane (b, m) = D_b[i][m] = C + sum_k A_b[i][k] * B_b[k][m].
// With C = 0x4B400000 the result bits read as a float must equal 12582912 + sum exactly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const int* a, const int* b, int* d, int c0) {
    v4i c = {c0, c0, c0, c0};
    v4i r = __builtin_amdgcn_mfma_i32_4x4x4i8(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[4 * threadIdx.x + i] = r[i];
}
int main() {
    int ha[64], hb[64], hd[256];
    srand(7);
    for (int i = 0; i < 64; ++i) {
        uint32_t x = 0, y = 0;
        for (int k = 0; k < 4; ++k) { x |= (uint32_t)(uint8_t)(rand() % 256 - 128) << (8 * k); y |= (uint32_t)(uint8_t)(rand() % 255 - 127) << (8 * k); }
        ha[i] = (int)x; hb[i] = (int)y;
    }
    int *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    const int magic = 0x4B400000;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd, magic);
    hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    int bad1 = 0, bad2 = 0, badf = 0;
    for (int b = 0; b < 16; ++b)
        for (int m = 0; m < 4; ++m)
            for (int i = 0; i < 4; ++i) {
                int s1 = 0, s2 = 0;
                for (int kk = 0; kk < 4; ++kk) {
                    const int a1 = (int8_t)(ha[4 * b + i] >> (8 * kk)), b1 = (int8_t)(hb[4 * b + m] >> (8 * kk));
                    s1 += a1 * b1;                                                     // H1: D[i][m], register i of lane (b, m)
                    const int a2 = (int8_t)(ha[4 * b + m] >> (8 * kk)), b2 = (int8_t)(hb[4 * b + i] >> (8 * kk));
                    s2 += a2 * b2;                                                     // H2: transposed roles
                }
                const int got = hd[4 * (4 * b + m) + i];
                if (got != magic + s1) ++bad1;
                if (got != magic + s2) ++bad2;
                float f; memcpy(&f, &got, 4);
                if (f != 12582912.0f + (float)s1) ++badf;
            }
    printf("H1 mismatches %d / 256, H2 mismatches %d / 256, float-addend mismatches (under H1) %d\n", bad1, bad2, badf);
    return 0;
}
