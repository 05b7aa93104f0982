// Experiment (not part of the product): decode-sized weight streams (9.4 / 28 / 51 / 107 MB per launch, HBM-cold, launches chained in
// a hipGraph like a token step) on 256 x 1024-thread workgroups reading 1152-byte records —
//   A  register ring (4 records per wave), the record consumed by a few ALU instructions          = kernels_v9.h without block math
//   B  the same + W dependent-ish vector instructions per record (the cost class of the step's)   = kernels_v9.h
//   C  loader / consumer: two loader waves per workgroup copy records into LDS by LDS-DMA (global_load_lds_dwordx4), fourteen consumer
//      waves read them from LDS and run the same W instructions; LDS counters per consumer (filled / consumed)
// Prints us per launch and TB/s for each size and form, and checks C's checksum against A's (a stale LDS read would show).
//   hipcc -O3 --offload-arch=gfx950 -o stream_probe2.bin stream_probe2.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int REC = 1152;

// W instructions of the 3.3-cycle class on the record's words (v_mad_i32_i24 chains over four accumulators)
template <int W> __device__ __forceinline__ unsigned work(u32x4 b, u32x4 h, unsigned acc) {
    int a0 = (int)b[0], a1 = (int)b[1], a2 = (int)b[2], a3 = (int)b[3];
    const int m0 = (int)(h[0] & 0xffff), m1 = (int)(h[1] & 0xffff);
#pragma unroll
    for (int i = 0; i < W / 4; ++i) {
        asm volatile("v_mad_i32_i24 %0, %0, %4, %5\n v_mad_i32_i24 %1, %1, %5, %4\n v_mad_i32_i24 %2, %2, %4, %5\n v_mad_i32_i24 %3, %3, %5, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m0), "v"(m1));
    }
    return acc + (unsigned)(a0 ^ a1 ^ a2 ^ a3) + h[2] + h[3];
}

// PAT: which records a wave walks — 0: recs_per_wave consecutive records; S > 0: units of S consecutive records, unit u of the launch
// belongs to workgroup u % grid, wave (u / grid) % 16 (the decode mat-vec's distribution: S = 4 at K = 4096, 11 at K = 11008)
template <int W, int PAT>
__global__ void __launch_bounds__(1024) ring_kernel(const unsigned char* base, int recs_per_wave, unsigned* sink) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = gridDim.x;
    const size_t stride_units = (size_t)G * 16;
    const unsigned char* p = PAT == 0 ? base + ((size_t)(blockIdx.x * 16 + wv) * recs_per_wave) * REC
                                      : base + ((size_t)(blockIdx.x + G * wv) * PAT) * REC;
    u32x4 body[4], hdr[4];
    int left = recs_per_wave, s = 0;
    auto issue = [&](int k) __attribute__((always_inline)) {
        body[k] = __builtin_nontemporal_load((const u32x4*)(p + lane * 16));
        hdr[k] = __builtin_nontemporal_load((const u32x4*)(p + 1024 + (lane >> 3) * 16));
        if (left > 1) {
            --left; p += REC;
            if (PAT > 0 && ++s == PAT) { s = 0; p += (stride_units - 1) * PAT * REC; }
        }
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) issue(k);
    unsigned acc = 0;
    for (int st = 0; st < recs_per_wave; st += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32x4 b = body[k], h = hdr[k];
            unsigned v = work<W>(b, h, 0);
            asm volatile("" : "+v"(v));
            issue(k);
            if (st + k < recs_per_wave) acc += v;
        }
    }
    atomicAdd(&sink[(blockIdx.x * 16 + wv) & 1023], acc);
}

// ---- loader / consumer -----------------------------------------------------------------------------------------------------------
constexpr int NC = 14, NL = 2, CPL = NC / NL, D = 4, LAG = 14;   // consumers, loaders, consumers per loader, slots per consumer, records in flight per loader
struct LcLds {
    unsigned char ring[NC * D * REC];
    unsigned filled[16], consumed[16];
};
__device__ __forceinline__ void glds16(const void* g, unsigned lds_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n s_mov_b32 m0, %2\n s_nop 0\n global_load_lds_dwordx4 %1, off nt\n s_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_off) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

template <int W>
__global__ void __launch_bounds__(1024) lc_kernel(const unsigned char* base, int R, unsigned* sink) {   // R records per consumer wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LcLds& S = *reinterpret_cast<LcLds*>(smem);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 16) { S.filled[threadIdx.x] = 0u; S.consumed[threadIdx.x] = 0u; }
    __syncthreads();
    if (wv >= NC) {   // ---- loader ----
        const int L = wv - NC;
        const int total = CPL * R;
        auto where = [&](int idx, int& c, int& r) { c = L * CPL + idx % CPL; r = idx / CPL; };
        auto publish = [&](int idx) {
            int c, r; where(idx, c, r);
            if (lane == 0) __hip_atomic_store(&S.filled[c], (unsigned)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        for (int idx = 0; idx < total; ++idx) {
            int c, r; where(idx, c, r);
            if (r >= D) {   // the slot is free once the consumer has finished record r - D
                while (__hip_atomic_load(&S.consumed[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)(r - D + 1)) __builtin_amdgcn_s_sleep(1);
            }
            const unsigned char* src = base + (((size_t)(blockIdx.x * NC + c)) * R + r) * REC;
            const unsigned slot = (unsigned)((c * D + (r % D)) * REC);
            glds16(src + lane * 16, slot);
            if (lane < 8) glds16(src + 1024 + lane * 16, slot + 1024u);
            if (idx >= LAG) { vm_wait<2 * LAG>(); publish(idx - LAG); }
        }
        vm_wait<0>();
        for (int idx = total > LAG ? total - LAG : 0; idx < total; ++idx) publish(idx);
        return;
    }
    // ---- consumer ----
    unsigned acc = 0;
    for (int r = 0; r < R; ++r) {
        while (__hip_atomic_load(&S.filled[wv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= (unsigned)r) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const unsigned char* slot = S.ring + (size_t)(wv * D + (r % D)) * REC;
        const u32x4 b = *(const u32x4*)(slot + lane * 16);
        const u32x4 h = *(const u32x4*)(slot + 1024 + (lane >> 3) * 16);
        unsigned v = work<W>(b, h, 0);
        asm volatile("" : "+v"(v));
        acc += v;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&S.consumed[wv], (unsigned)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    atomicAdd(&sink[(blockIdx.x * 16 + wv) & 1023], acc);
}

template <class F> static double run_graph(F launch, int n_launch, hipStream_t s) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n_launch; ++i) launch(i, s);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return (double)ms * 1e3 / (3.0 * n_launch);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t bytes = (size_t)2 << 30;
    unsigned char* buf; unsigned* sink;
    CK(hipMalloc(&buf, bytes + (64 << 20))); CK(hipMalloc(&sink, 4096));
    {   // non-trivial contents (a checksum over zeros proves nothing)
        std::vector<unsigned> h(bytes / 4);
        unsigned x = 12345u;
        for (size_t i = 0; i < h.size(); ++i) { x = x * 1664525u + 1013904223u; h[i] = x; }
        CK(hipMemcpy(buf, h.data(), bytes, hipMemcpyHostToDevice));
    }
    hipStream_t s; CK(hipStreamCreate(&s));
    CK(hipFuncSetAttribute((const void*)lc_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LcLds)));
    CK(hipFuncSetAttribute((const void*)lc_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LcLds)));
    CK(hipFuncSetAttribute((const void*)lc_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LcLds)));
    printf("us per launch (TB/s) inside a hipGraph of 16 launches over different windows of a 2 GiB buffer; records of 1152 B\n");
    const double sizes_mb[] = {9.4, 28.3, 50.7, 107.5};
    for (double mb : sizes_mb) {
        // records per wave so that 4096 (ring) / 3584 (consumer) waves cover about `mb`; equal bytes for both forms: lcm-ish rounding
        const int rpw_ring = (int)(mb * 1e6 / REC / 4096 + 0.5);
        const int rpw_lc = (int)(mb * 1e6 / REC / (256 * NC) + 0.5);
        const size_t win_ring = (size_t)4096 * rpw_ring * REC, win_lc = (size_t)256 * NC * rpw_lc * REC;
        const size_t stride = ((std::max(win_ring, win_lc) + (16 << 20)) >> 20) << 20;
        const int nl = (int)std::min<size_t>(16, bytes / stride);
        auto tb = [&](size_t w, double us) { return (double)w / us / 1e6; };
        printf("%6.1f MB:", mb);
#define RING(Wv, Pv) do { const int rp = Pv ? ((rpw_ring + Pv - 1) / Pv) * Pv : rpw_ring; \
        double us = run_graph([&](int i, hipStream_t st) { hipLaunchKernelGGL((ring_kernel<Wv, Pv>), dim3(256), dim3(1024), 0, st, buf + (size_t)i * stride, rp, sink); }, nl, s); \
        printf("  ring W=%-2d pat %-2d %6.2f us (%.2f)", Wv, Pv, us, (double)4096 * rp * REC / us / 1e6); } while (0)
#define LC(Wv) do { double us = run_graph([&](int i, hipStream_t st) { hipLaunchKernelGGL((lc_kernel<Wv>), dim3(256), dim3(1024), sizeof(LcLds), st, buf + (size_t)i * stride, rpw_lc, sink); }, nl, s); \
        printf("  | loader/consumer W=%-2d %6.2f us (%.2f)", Wv, us, tb(win_lc, us)); } while (0)
        RING(0, 0); RING(64, 0); RING(0, 1); RING(64, 1); RING(0, 4); RING(64, 4); RING(0, 11); RING(64, 11);
        LC(0); LC(64);
        printf("\n");
    }
    // checksum: the loader/consumer form over one window against the ring form over the same bytes (W = 0: plain sums)
    {
        const int rpw_lc = 28, rpw_ring = 0;
        (void)rpw_ring;
        std::vector<unsigned> a(1024), b(1024);
        CK(hipMemset(sink, 0, 4096));
        hipLaunchKernelGGL((lc_kernel<0>), dim3(256), dim3(1024), sizeof(LcLds), s, buf, rpw_lc, sink);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(a.data(), sink, 4096, hipMemcpyDeviceToHost));
        unsigned long long ta = 0; for (unsigned v : a) ta += v;
        // the same bytes: 256 * 14 * 28 records = 7 * (4096 waves x 3.5) -> use the ring kernel with 3584 waves' worth: grid 224 x 16 waves x 28 records
        CK(hipMemset(sink, 0, 4096));
        hipLaunchKernelGGL((ring_kernel<0, 0>), dim3(224), dim3(1024), 0, s, buf, rpw_lc, sink);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(b.data(), sink, 4096, hipMemcpyDeviceToHost));
        unsigned long long tbb = 0; for (unsigned v : b) tbb += v;
        printf("checksum over the same %d records: loader/consumer %llu, register ring %llu -> %s\n", 256 * NC * rpw_lc, ta & 0xffffffffull, tbb & 0xffffffffull,
               (ta & 0xffffffffull) == (tbb & 0xffffffffull) ? "equal" : "DIFFERENT (stale LDS read?)");
    }
    return 0;
}
