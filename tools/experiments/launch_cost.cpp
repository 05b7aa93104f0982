// Experiment (not part of the product): what a dependent kernel boundary inside a hipGraph costs on MI355X as a function of the launch
// shape — threads per workgroup, workgroups, dynamic LDS, kernel-argument bytes, registers — for kernels that exit at once.
//   hipcc -O3 --offload-arch=gfx950 -o launch_cost.bin launch_cost.cpp && ./launch_cost.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Big { int v[120]; };   // 480 bytes of kernel arguments
extern __shared__ unsigned char dyn[];
__global__ void __launch_bounds__(1024) k_small(int* p, int flag) { if (flag) p[threadIdx.x] = 1; }
__global__ void __launch_bounds__(1024) k_big(Big b, int* p, int flag) { if (flag) p[threadIdx.x] = b.v[flag & 63]; }
// a kernel that needs 128 VGPRs (forces the register allocation of the mat-vec)
__global__ void __launch_bounds__(1024) k_regs(int* p, int flag) {
    if (!flag) return;
    float a[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) a[i] = (float)p[i + threadIdx.x];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int i = 0; i < 96; ++i) a[i] = a[i] * a[(i + 7) % 96] + 1.0f;
    float s = 0;
#pragma unroll
    for (int i = 0; i < 96; ++i) s += a[i];
    p[threadIdx.x] = (int)s;
}
// kernels that only RESERVE registers (the highest one is named in a clobber list) and exit at once
#define KV(N, R) __global__ void __launch_bounds__(1024) k_v##N(int* p, int flag) { if (flag) { asm volatile("v_mov_b32 " R ", 0" ::: R); p[threadIdx.x] = 1; } }
KV(32, "v31") KV(48, "v47") KV(64, "v63") KV(72, "v71") KV(80, "v79") KV(96, "v95") KV(104, "v103") KV(112, "v111") KV(120, "v119") KV(128, "v127")
__global__ void __launch_bounds__(512) k_v256(int* p, int flag) { if (flag) { asm volatile("v_mov_b32 v255, 0" ::: "v255"); p[threadIdx.x] = 1; } }
__global__ void __launch_bounds__(256) k_v512(int* p, int flag) { if (flag) { asm volatile("v_mov_b32 v255, 0\n v_accvgpr_write_b32 a255, 0" ::: "v255", "a255"); p[threadIdx.x] = 1; } }
// touches one element per workgroup then stores: the minimal "real" dependency chain (load -> store)
__global__ void __launch_bounds__(1024) k_touch(int* p, int flag) { if (threadIdx.x == 0) p[blockIdx.x] = p[blockIdx.x + 4096] + 1; }

template <class F> static double run(F launch, int n, hipStream_t s) {
    hipGraph_t g; hipGraphExec_t ge;
    OK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; ++i) launch(s);
    OK(hipStreamEndCapture(s, &g));
    OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) OK(hipGraphLaunch(ge, s));
    OK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    const int reps = 10;
    OK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) OK(hipGraphLaunch(ge, s));
    OK(hipEventRecord(e1, s));
    OK(hipStreamSynchronize(s));
    float ms = 0; OK(hipEventElapsedTime(&ms, e0, e1));
    OK(hipGraphExecDestroy(ge)); OK(hipGraphDestroy(g));
    return (double)ms * 1e3 / (reps * n);
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    int* p; OK(hipMalloc(&p, 1 << 20)); OK(hipMemset(p, 0, 1 << 20));
    hipStream_t s; OK(hipStreamCreate(&s));
    OK(hipFuncSetAttribute((const void*)k_small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    OK(hipFuncSetAttribute((const void*)k_big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    OK(hipFuncSetAttribute((const void*)k_regs, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int n = 256;
    printf("us per dependent launch inside a hipGraph of %d launches (kernels exit at once)\n", n);
    const int shapes[][2] = {{256, 1024}, {256, 512}, {256, 256}, {256, 64}, {512, 512}, {1024, 256}, {64, 1024}, {128, 1024}, {2048, 128}};
    for (auto& sh : shapes) {
        for (int lds : {0, 44 * 1024, 150 * 1024}) {
            const int grid = sh[0], thr = sh[1];
            double t = run([&](hipStream_t st) { hipLaunchKernelGGL(k_small, dim3(grid), dim3(thr), lds, st, p, 0); }, n, s);
            printf("  grid %4d x %4d threads, dyn LDS %6d B, 16 B args            : %.3f\n", grid, thr, lds, t);
        }
    }
    Big b; for (int i = 0; i < 120; ++i) b.v[i] = i;
    for (int lds : {0, 44 * 1024}) {
        double t = run([&](hipStream_t st) { hipLaunchKernelGGL(k_big, dim3(256), dim3(1024), lds, st, b, p, 0); }, n, s);
        printf("  grid  256 x 1024 threads, dyn LDS %6d B, 496 B args           : %.3f\n", lds, t);
    }
    for (int lds : {0, 44 * 1024}) {
        double t = run([&](hipStream_t st) { hipLaunchKernelGGL(k_regs, dim3(256), dim3(1024), lds, st, p, 0); }, n, s);
        printf("  grid  256 x 1024 threads, dyn LDS %6d B, 128-VGPR kernel      : %.3f\n", lds, t);
    }
    {
        typedef void (*kf)(int*, int);
        struct { const char* n; kf f; int thr; } ks[] = {{"32", k_v32, 1024}, {"48", k_v48, 1024}, {"64", k_v64, 1024}, {"72", k_v72, 1024}, {"80", k_v80, 1024}, {"96", k_v96, 1024},
                                                         {"104", k_v104, 1024}, {"112", k_v112, 1024}, {"120", k_v120, 1024}, {"128", k_v128, 1024}, {"256", k_v256, 512}, {"512", k_v512, 256}};
        for (auto& k : ks) {
            for (int thr : {k.thr, 256, 64}) {
                if (thr > k.thr) continue;
                double t = run([&](hipStream_t st) { hipLaunchKernelGGL(k.f, dim3(256), dim3(thr), 0, st, p, 0); }, n, s);
                printf("  grid  256 x %4d threads, kernel reserving %3s VGPRs           : %.3f\n", thr, k.n, t);
            }
        }
    }
    for (auto& sh : shapes) {
        const int grid = sh[0], thr = sh[1];
        double t = run([&](hipStream_t st) { hipLaunchKernelGGL(k_touch, dim3(grid), dim3(thr), 0, st, p, 1); }, n, s);
        printf("  grid %4d x %4d threads, one dependent load -> store per WG    : %.3f\n", grid, thr, t);
    }
    return 0;
}
