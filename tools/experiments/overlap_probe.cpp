// Experiment (not part of the product): can two dependent kernels overlap their launch/startup cost on MI355X when they
// are put on two alternating streams of one hipGraph and the true dependency is carried by a device-scope counter?
// Each "op" = 256 workgroups x 512 threads that (1) spin until the previous op's counter reaches its target,
// (2) do ~WORK us of dependent ALU work, (3) release + signal.  Compared: A) one stream, stream-ordered (no counters),
// B) two alternating streams + counters.  Prints us per op.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/overlap_probe tools/experiments/overlap_probe.cpp && /tmp/overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(512) op_kernel(unsigned* ctr, int idx, unsigned target, int use_dep, int work, float* sink,
                                                  unsigned long long* stamps) {
    const bool lead = threadIdx.x == 0;
    unsigned long long t0 = 0;
    if (lead && blockIdx.x == 0) t0 = __builtin_readcyclecounter();
    if (use_dep && idx > 0) {
        if ((threadIdx.x & 63) == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(&ctr[idx - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 14))
                __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    float v = (float)threadIdx.x;
    for (int i = 0; i < work; ++i) v = __builtin_fmaf(v, 1.0000001f, 0.5f);
    if (v == 12345.678f) sink[threadIdx.x] = v;
    if (use_dep) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&ctr[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lead && blockIdx.x == 0 && stamps) { stamps[2 * idx] = t0; stamps[2 * idx + 1] = __builtin_readcyclecounter(); }
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int n_ops = 160, wgs = 256, reps = 20;
    const int work = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned* ctr; float* sink; unsigned long long* stamps;
    OK(hipMalloc(&ctr, n_ops * sizeof(unsigned)));
    OK(hipMalloc(&sink, 4096));
    OK(hipMalloc(&stamps, 2 * n_ops * sizeof(unsigned long long)));
    hipStream_t sa, sb;
    OK(hipStreamCreate(&sa)); OK(hipStreamCreate(&sb));
    hipEvent_t e0, e1, fork, join;
    OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    OK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); OK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    const unsigned target = wgs * 8;  // every wave signals
    for (int mode = 0; mode < 3; ++mode) {
        printf("mode %d capture...\n", mode);   // 0: one stream, no counters; 1: one stream + counters; 2: two streams + counters
        hipGraph_t g; hipGraphExec_t ge;
        OK(hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal));
        OK(hipMemsetAsync(ctr, 0, n_ops * sizeof(unsigned), sa));
        if (mode == 2) { OK(hipEventRecord(fork, sa)); OK(hipStreamWaitEvent(sb, fork, 0)); }
        for (int i = 0; i < n_ops; ++i) {
            hipStream_t s = (mode == 2 && (i & 1)) ? sb : sa;
            hipLaunchKernelGGL(op_kernel, dim3(wgs), dim3(512), 0, s, ctr, i, target, mode >= 1 ? 1 : 0, work, sink, stamps);
        }
        if (mode == 2) { OK(hipEventRecord(join, sb)); OK(hipStreamWaitEvent(sa, join, 0)); }
        OK(hipStreamEndCapture(sa, &g));
        OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 3; ++w) OK(hipGraphLaunch(ge, sa));
        OK(hipStreamSynchronize(sa));
        OK(hipEventRecord(e0, sa));
        for (int r = 0; r < reps; ++r) OK(hipGraphLaunch(ge, sa));
        OK(hipEventRecord(e1, sa));
        OK(hipStreamSynchronize(sa));
        float ms = 0; OK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(2 * n_ops);
        OK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        double in_kernel = 0; int overl = 0;
        for (int i = 0; i < n_ops; ++i) { in_kernel += (double)(h[2 * i + 1] - h[2 * i]); if (i && h[2 * i] < h[2 * i - 1]) ++overl; }
        printf("mode %d work %d: %.2f us per op (graph of %d ops); wg0 in-kernel avg %.0f clk; ops that started before the previous op's wg0 ended: %d\n",
               mode, work, ms * 1e3 / reps / n_ops, n_ops, in_kernel / n_ops, overl);
        OK(hipGraphExecDestroy(ge)); OK(hipGraphDestroy(g));
    }
    return 0;
}
