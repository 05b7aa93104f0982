// Experiment (not part of the product): what does a grid-wide barrier cost on MI355X for a resident grid of 256 workgroups x 1024
// threads (one per CU: the decode mat-vec's shape), built from a device-scope counter with release / acquire fences — and do values
// written before it by workgroups on OTHER XCDs arrive (activations handed from phase to phase of a persistent token-step kernel)?
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/grid_barrier_probe tools/experiments/grid_barrier_probe.cpp && /tmp/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// Two-level form: the workgroups of an XCD (blockIdx % 8) meet at their own counter, the last of each XCD at a global one, the last of
// all publishes the round in a flag everybody polls (one writer).  ctr2: [8 x 16 (XCD counters, a cache line apart) | global | flag].
__device__ __forceinline__ void grid_barrier2(unsigned* ctr2, unsigned round, unsigned nwg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned xcd = blockIdx.x & 7u, per = (nwg + 7u - xcd) / 8u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned a = __hip_atomic_fetch_add(ctr2 + 16 * xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a == round * per + per - 1u) {
            const unsigned g = __hip_atomic_fetch_add(ctr2 + 128, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned nx = nwg < 8u ? nwg : 8u;
            if (g == round * nx + nx - 1u) __hip_atomic_store(ctr2 + 144, round + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(ctr2 + 144, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round + 1u) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// mode 0: barriers only; mode 1: every workgroup writes 16 floats (a function of the round) before the barrier and checks ALL
// workgroups' values after it (a 4096-float "activation vector" handed across the barrier)
__global__ void __launch_bounds__(1024) probe(unsigned* ctr, float* buf, int rounds, int mode, unsigned* errors, unsigned long long* cyc) {
    const unsigned nwg = gridDim.x;
    unsigned long long t0 = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) t0 = __builtin_readcyclecounter();
    unsigned bad = 0;
    for (int r = 0; r < rounds; ++r) {
        float* cur = buf + (size_t)(r & 1) * nwg * 16;
        if ((mode & 1) && threadIdx.x < 16) cur[blockIdx.x * 16 + threadIdx.x] = (float)(r * 7 + (int)blockIdx.x) + 0.25f * (float)threadIdx.x;
        if (mode >= 2) grid_barrier2(ctr, (unsigned)r, nwg); else grid_barrier(ctr, (unsigned)(r + 1) * nwg);
        if (mode & 1) {
            for (unsigned i = threadIdx.x; i < nwg * 16; i += 1024) {
                const float want = (float)(r * 7 + (int)(i >> 4)) + 0.25f * (float)(i & 15);
                if (cur[i] != want) ++bad;
            }
        }
    }
    if (bad) atomicAdd(errors, bad);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = __builtin_readcyclecounter() - t0;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    int cus = 0;
    OK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    unsigned* ctr; float* buf; unsigned* errors; unsigned long long* cyc;
    OK(hipMalloc(&ctr, 1024)); OK(hipMalloc(&buf, 2 * 256 * 16 * 4 * 2)); OK(hipMalloc(&errors, 4)); OK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    const int rounds = 2000;
    for (int mode = 0; mode < 4; ++mode)
        for (int grid : {64, 128, cus}) {
            OK(hipMemset(ctr, 0, 1024)); OK(hipMemset(errors, 0, 4));
            hipLaunchKernelGGL(probe, dim3(grid), dim3(1024), 0, 0, ctr, buf, 10, mode, errors, cyc);   // warm-up
            OK(hipDeviceSynchronize());
            OK(hipMemset(ctr, 0, 1024)); OK(hipMemset(errors, 0, 4));
            OK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(probe, dim3(grid), dim3(1024), 0, 0, ctr, buf, rounds, mode, errors, cyc);
            OK(hipEventRecord(e1, 0));
            OK(hipDeviceSynchronize());
            float ms = 0; OK(hipEventElapsedTime(&ms, e0, e1));
            unsigned herr = 0; unsigned long long hc = 0;
            OK(hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost)); OK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
            printf("mode %d (%s, %s), grid %3d x 1024: %.3f us per barrier round (%.0f cycles), stale values seen: %u\n", mode,
                   mode >= 2 ? "two-level" : "one counter", (mode & 1) ? "16 floats per workgroup written before, all read after" : "barrier only", grid, ms * 1e3 / rounds, (double)hc / rounds, herr);
        }
    return 0;
}
