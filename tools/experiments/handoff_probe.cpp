// What a pipeline hop costs on this box: two streams (a producer "stage" and a consumer "stage"), a chain of H hops per iteration,
// each hop = a small producer kernel, the hand-off, a small consumer kernel.  Forms:
//   A  hipMemcpyPeerAsync + hipEventRecord / hipStreamWaitEvent          (round 4's pipeline.cc)
//   B  the producer kernel stores the rows itself; hipEventRecord / hipStreamWaitEvent
//   C  the producer kernel stores the rows and a sequence flag (system-scope release); the consumer STREAM waits with
//      hipStreamWaitValue32 (the command processor polls: no CU is occupied)
//   D  as C, but the flag is written by hipStreamWriteValue32 behind the producer kernel
//   E  the consumer kernel itself spins on the flag (occupies a wave while it waits)
// Build: hipcc -O2 --offload-arch=gfx950 -o handoff_probe handoff_probe.cpp ; run: ./handoff_probe [hops] [devA devB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every produce kernel busy-waits ~20 us (100 MHz wall clock) so that the host's launch calls run AHEAD of the device, as they do in
// the product (a stage's work per token is >= 150 us): what is measured is the device-side cost of the hand-off, not the API calls
__device__ int g_spin_ticks = 2000;
__global__ void produce(const float* __restrict__ src, float* __restrict__ dst, int n, unsigned* flag, unsigned seq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)g_spin_ticks) {}
    }
    if (i < n) __builtin_nontemporal_store(src[i] + 1.0f, dst + i);
    if (flag) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            // last workgroup to arrive publishes the flag
            __shared__ unsigned last;
            (void)last;
            unsigned* cnt = flag + 16;
            const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM);
            if (old == gridDim.x - 1) {
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}
// form C: the flag word and the arrival counter live in different allocations
__global__ void produce2(const float* __restrict__ src, float* __restrict__ dst, int n, unsigned* flag, unsigned* cnt, unsigned seq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)g_spin_ticks) {}
    }
    if (i < n) __builtin_nontemporal_store(src[i] + 1.0f, dst + i);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM);
        if (old == gridDim.x - 1) {
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void consume(const float* __restrict__ src, float* __restrict__ dst, int n, const unsigned* flag, unsigned seq) {
    if (flag) {
        if (threadIdx.x == 0) {
            while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = __builtin_nontemporal_load(src + i) * 0.5f;
}

int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 8;
    int ndev = 0;
    CK(hipGetDeviceCount(&ndev));
    const int dA = argc > 3 ? atoi(argv[2]) : 0, dB = argc > 3 ? atoi(argv[3]) : (ndev > 1 ? 1 : 0);
    printf("devices visible %d; producer on %d, consumer on %d; %d hops per iteration, 16 KiB per hop\n", ndev, dA, dB, H);
    const int n = 4096;
    hipStream_t sA, sB;
    float *xa, *xa2, *xb, *xb2;
    unsigned* flag;
    CK(hipSetDevice(dA)); CK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking));
    CK(hipMalloc(&xa, n * 4)); CK(hipMalloc(&xa2, n * 4)); CK(hipMemset(xa, 0, n * 4));
    CK(hipSetDevice(dB)); CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
    CK(hipMalloc(&xb, n * 4)); CK(hipMalloc(&xb2, n * 4));
    // the flag lives on the consumer's device, in signal memory (what hipStreamWaitValue32 requires)
    // hipStreamWaitValue32 wants signal memory: 8 bytes per allocation.  Two flags (one per direction) + the arrival counters of form C
    // in ordinary memory behind them.
    unsigned* sig[2] = {nullptr, nullptr};
    hipError_t ef = hipSuccess;
    for (int d = 0; d < 2; ++d) {
        hipError_t e = hipExtMallocWithFlags((void**)&sig[d], 8, hipMallocSignalMemory);
        if (e != hipSuccess) { ef = e; printf("hipMallocSignalMemory(8): %s\n", hipGetErrorString(e)); }
    }
    CK(hipMalloc((void**)&flag, 512));
    CK(hipMemset(flag, 0, 512));
    if (ef == hipSuccess) { CK(hipMemset(sig[0], 0, 8)); CK(hipMemset(sig[1], 0, 8)); }
    if (dA != dB) {
        int can = 0;
        CK(hipDeviceCanAccessPeer(&can, dA, dB));
        printf("peer access %d -> %d: %d\n", dA, dB, can);
        CK(hipSetDevice(dA)); (void)hipDeviceEnablePeerAccess(dB, 0); (void)hipGetLastError();
        CK(hipSetDevice(dB)); (void)hipDeviceEnablePeerAccess(dA, 0); (void)hipGetLastError();
    }
    CK(hipDeviceSynchronize());
    std::vector<hipEvent_t> ev(2 * H);
    for (auto& e : ev) { CK(hipSetDevice(dA)); CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
    unsigned seq = 0;
    const dim3 g(n / 256), b(256);
    auto run = [&](char form, int iters) -> double {
        CK(hipSetDevice(dA)); CK(hipStreamSynchronize(sA)); CK(hipSetDevice(dB)); CK(hipStreamSynchronize(sB));
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < iters; ++it) {
            for (int h = 0; h < H; ++h) {
                // hop: A's stream -> B's stream, then back B -> A (so that the chain is serial like a token through the stages)
                for (int dir = 0; dir < 2; ++dir) {
                    hipStream_t sp = dir ? sB : sA, sc = dir ? sA : sB;
                    const int dp = dir ? dB : dA, dc = dir ? dA : dB;
                    float* src = dir ? xb2 : xa2;      // the producer stage's own row
                    float* own = dir ? xb : xa;
                    float* dst = dir ? xa : xb;        // the consumer stage's hand-off buffer
                    float* cdst = dir ? xa2 : xb2;
                    hipEvent_t e = ev[2 * h + dir];
                    ++seq;
                    CK(hipSetDevice(dp));
                    if (form == 'A') {
                        produce<<<g, b, 0, sp>>>(src, own, n, nullptr, 0);
                        CK(hipMemcpyPeerAsync(dst, dc, own, dp, n * 4, sp));
                        CK(hipEventRecord(e, sp));
                        CK(hipSetDevice(dc));
                        CK(hipStreamWaitEvent(sc, e, 0));
                        consume<<<g, b, 0, sc>>>(dst, cdst, n, nullptr, 0);
                    } else if (form == 'B') {
                        produce<<<g, b, 0, sp>>>(src, dst, n, nullptr, 0);
                        CK(hipEventRecord(e, sp));
                        CK(hipSetDevice(dc));
                        CK(hipStreamWaitEvent(sc, e, 0));
                        consume<<<g, b, 0, sc>>>(dst, cdst, n, nullptr, 0);
                    } else if (form == 'C') {
                        produce2<<<g, b, 0, sp>>>(src, dst, n, sig[dir], flag + 32 * dir, seq);
                        CK(hipSetDevice(dc));
                        CK(hipStreamWaitValue32(sc, sig[dir], seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
                        consume<<<g, b, 0, sc>>>(dst, cdst, n, nullptr, 0);
                    } else if (form == 'D') {
                        produce<<<g, b, 0, sp>>>(src, dst, n, nullptr, 0);
                        CK(hipStreamWriteValue32(sp, sig[dir], seq, 0));
                        CK(hipSetDevice(dc));
                        CK(hipStreamWaitValue32(sc, sig[dir], seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
                        consume<<<g, b, 0, sc>>>(dst, cdst, n, nullptr, 0);
                    } else {
                        produce<<<g, b, 0, sp>>>(src, dst, n, flag + 32 * dir, seq);
                        CK(hipSetDevice(dc));
                        consume<<<g, b, 0, sc>>>(dst, cdst, n, flag + 32 * dir, seq);
                    }
                }
            }
            // one host sync per iteration, like a decode step
            CK(hipSetDevice(dA)); CK(hipStreamSynchronize(sA));
            CK(hipSetDevice(dB)); CK(hipStreamSynchronize(sB));
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        return us / iters;
    };
    // the floor: the same kernels on ONE stream, no hand-off
    {
        CK(hipSetDevice(dA));
        auto t0 = std::chrono::steady_clock::now();
        const int iters = 200;
        for (int it = 0; it < iters; ++it) {
            for (int h = 0; h < 2 * H; ++h) { produce<<<g, b, 0, sA>>>(xa2, xa, n, nullptr, 0); consume<<<g, b, 0, sA>>>(xa, xa2, n, nullptr, 0); }
            CK(hipStreamSynchronize(sA));
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
        printf("floor: %d x (produce + consume) on one stream, one sync: %.1f us per iteration, %.2f us per pair\n", 2 * H, us, us / (2 * H));
    }
    for (char form : {'A', 'B', 'C', 'D', 'E'}) {
        if ((form == 'C' || form == 'D') && ef != hipSuccess) { printf("form %c skipped (no signal memory)\n", form); continue; }
        run(form, 20);
        const double us = run(form, 200);
        printf("form %c: %.1f us per iteration of %d hops -> %.2f us per hop (incl. its two small kernels)\n", form, us, 2 * H, us / (2 * H));
        fflush(stdout);
    }
    return 0;
}
