// issue rate of the 4x4 matrix instructions (cycles per wave-instruction and SIMD, 1 / 2 / 4 waves per SIMD; independent x8 and dependent)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef int i4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define R8(S) S S S S S S S S
#define KERNEL(NAME, T, IND, DEPS)                                                                                           \
    template <int DEP> __global__ void __launch_bounds__(1024) NAME(int iters, unsigned long long* out, int* sink) {          \
        T a0, a1, a2, a3, a4, a5, a6, a7;                                                                                      \
        for (int k = 0; k < 4; ++k) { a0[k] = threadIdx.x + k; a1[k] = a0[k] + 1; a2[k] = a0[k] + 2; a3[k] = a0[k] + 3; a4[k] = a0[k] + 4; a5[k] = a0[k] + 5; a6[k] = a0[k] + 6; a7[k] = a0[k] + 7; } \
        int b = (int)threadIdx.x | 1, c = 0x01020304; float fb = 1.5f, fc = 0.25f;                                            \
        __syncthreads();                                                                                                      \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                           \
        for (int it = 0; it < iters; ++it) {                                                                                  \
            if (DEP) asm volatile(R8(R8(DEPS)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(fb), "v"(fc)); \
            else asm volatile(R8(IND) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(fb), "v"(fc)); \
        }                                                                                                                     \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                           \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                                      \
        if ((int)(a0[0] + a1[0] + a2[0] + a3[0] + a4[0] + a5[0] + a6[0] + a7[0]) == 0x7fffff12) sink[0] = 1;                   \
    }
#define I8(d) "v_mfma_i32_4x4x4_16b_i8 " d ", %8, %9, " d "\n"
#define F1(d) "v_mfma_f32_4x4x1_16b_f32 " d ", %10, %11, " d "\n"
#define I16(d) "v_mfma_i32_16x16x32_i8 " d ", %8, %9, " d "\n"
KERNEL(k_i8, i4, I8("%0") I8("%1") I8("%2") I8("%3") I8("%4") I8("%5") I8("%6") I8("%7"), I8("%0"))
KERNEL(k_f1, f4, F1("%0") F1("%1") F1("%2") F1("%3") F1("%4") F1("%5") F1("%6") F1("%7"), F1("%0"))
// the chunk step's mix: i8 product, f32 rank-1 product, 2 pk_add, 2 pk_fma (independent accumulators)
#define MIX(d, e) I8(d) F1(e) "v_pk_add_f32 " d ", " d ", " e "\n v_pk_fma_f32 " e ", " d ", " e ", " e "\n"
typedef float f2 __attribute__((ext_vector_type(2)));
template <int DEP> __global__ void __launch_bounds__(1024) k_mix(int iters, unsigned long long* out, int* sink) {
    f4 a0, a1, a2, a3;
    f2 p0, p1, p2, p3;
    for (int k = 0; k < 4; ++k) { a0[k] = threadIdx.x + k; a1[k] = a0[k] + 1; a2[k] = a0[k] + 2; a3[k] = a0[k] + 3; }
    for (int k = 0; k < 2; ++k) { p0[k] = threadIdx.x + k; p1[k] = p0[k] + 1; p2[k] = p0[k] + 2; p3[k] = p0[k] + 3; }
    int b = (int)threadIdx.x | 1, c = 0x01020304; float fb = 1.5f, fc = 0.25f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        asm volatile(R8(
            "v_mfma_i32_4x4x4_16b_i8 %0, %8, %9, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %10, %11, %1\n"
            "v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %5, %5, %4\n v_pk_fma_f32 %6, %4, %5, %6\n v_pk_fma_f32 %7, %5, %4, %7\n"
            "v_mfma_i32_4x4x4_16b_i8 %2, %8, %9, %2\n v_mfma_f32_4x4x1_16b_f32 %3, %10, %11, %3\n"
            "v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %5, %5, %4\n v_pk_fma_f32 %6, %4, %5, %6\n v_pk_fma_f32 %7, %5, %4, %7\n")
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(b), "v"(c), "v"(fb), "v"(fc));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if ((int)(a0[0] + a1[0] + a2[0] + a3[0] + p0[0] + p1[0] + p2[0] + p3[0]) == 0x7fffff12) sink[0] = 1;
}
typedef void (*kfn)(int, unsigned long long*, int*);
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned long long* out; int* sink;
    OK(hipMalloc(&out, 256 * 16 * 8)); OK(hipMalloc(&sink, 64));
    std::vector<unsigned long long> h(256 * 16);
    const int iters = 2000;
    struct { const char* n; kfn ind, dep; int per; } ops[] = {{"mfma_i32_4x4x4_16b_i8", k_i8<0>, k_i8<1>, 8}, {"mfma_f32_4x4x1_16b_f32", k_f1<0>, k_f1<1>, 8},
                                                             {"chunk step (2 mfma + 4 pk + nop)", k_mix<0>, nullptr, 16}};
    printf("cycles per wave-instruction (mix: per STEP of 2 mfma + 2 pk_add + 2 pk_fma) and SIMD at 1 / 2 / 4 waves per SIMD\n");
    for (auto& op : ops) {
        printf("%-34s |", op.n);
        for (int dep = 0; dep < 2; ++dep) {
            kfn f = dep ? op.dep : op.ind;
            if (!f) continue;
            for (int w = 1; w <= 4; w *= 2) {
                hipLaunchKernelGGL(f, dim3(256), dim3(64 * 4 * w), 0, 0, 10, out, sink);
                OK(hipDeviceSynchronize());
                hipLaunchKernelGGL(f, dim3(256), dim3(64 * 4 * w), 0, 0, iters, out, sink);
                OK(hipDeviceSynchronize());
                OK(hipMemcpy(h.data(), out, 256 * 16 * 8, hipMemcpyDeviceToHost));
                std::vector<unsigned long long> v;
                for (int bI = 0; bI < 256; ++bI) for (int k = 0; k < 4 * w; ++k) v.push_back(h[bI * 16 + k]);
                std::sort(v.begin(), v.end());
                const double per = dep ? 64.0 : (double)op.per;
                printf(" %7.2f", (double)v[v.size() / 2] / ((double)iters * per * w));
            }
            printf(" |");
        }
        printf("\n");
    }
    return 0;
}
