// Experiment (not part of the product), round 6, VERDICT item 3 step 1: can an LDS-DMA loader in the guide's geometry stream this repo's decode weights at
// >= 6 TB/s?  (MI355X_MICROARCH.md rows `ldsdma-fill`, `nt-weights`, `engine-vs-launches`: one loader wave + three consumer waves per CU, a ring of
// 8 x 16 KiB slots, nt policy: 6.4 - 6.8 TB/s chip-wide.  The round-4 probe (stream_probe2.cpp) handed over 1152-byte records one at a time between 2 loader
// and 14 consumer waves and reached 2.1 - 2.4.)
//   grid = NWG workgroups (256: one per CU; 512: two per CU) of 256 threads; workgroup g streams its own contiguous run of `nslots` 16 KiB slots — the
//   weight records of a row group are contiguous in LAYOUT_L9 (quant.h), so a slot is 14.2 consecutive 1152-byte records; slot edges need not be record edges
//   for the question asked here.
//   loader  = wave 3: per slot 16 pieces of 1 KiB (global_load_lds_dwordx4, 16 B per lane), LAG slots in flight (vmcnt), publishes `filled`;
//             waits for the slot's previous tenant to be consumed by all three consumers.
//   consumer = waves 0..2: wait `filled`, read a third of the slot from LDS (ds_read_b128) + W vector instructions per 16 B, publish `consumed[wave]`.
// Prints us per launch and TB/s inside a hipGraph of up to 16 launches over different windows of a 2 GiB buffer (HBM-cold), and a checksum against a plain
// register-load kernel over the same bytes.
//   hipcc -O3 --offload-arch=gfx950 -o engine_probe.bin engine_probe.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int SLOT = 16384, RING = 8, PIECES = SLOT / 1024;

template <bool NT> __device__ __forceinline__ void glds16(const void* g, unsigned lds_off) {
    if (NT) asm volatile("s_mov_b32 m0, %1\n s_nop 0\n global_load_lds_dwordx4 %0, off nt" ::"v"(g), "s"(lds_off) : "memory", "m0");
    else asm volatile("s_mov_b32 m0, %1\n s_nop 0\n global_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_off) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

struct Ctl { unsigned filled, consumed[3]; };

template <int W> __device__ __forceinline__ unsigned work(u32x4 b, unsigned acc) {
    int a0 = (int)b[0], a1 = (int)b[1], a2 = (int)b[2], a3 = (int)b[3];
#pragma unroll
    for (int i = 0; i < W / 4; ++i)
        asm volatile("v_mad_i32_i24 %0, %0, %1, %2\n v_mad_i32_i24 %1, %1, %2, %3\n v_mad_i32_i24 %2, %2, %3, %0\n v_mad_i32_i24 %3, %3, %0, %1"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    return acc + (unsigned)a0 + (unsigned)a1 + (unsigned)a2 + (unsigned)a3;
}

// LAG: slots the loader keeps in flight before it publishes the oldest (vmcnt counts pieces: 16 per slot, the counter holds 63)
template <int W, bool NT, int LAG>
__global__ void __launch_bounds__(256) engine_kernel(const unsigned char* base, int nslots, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Ctl& C = *reinterpret_cast<Ctl*>(smem + RING * SLOT);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) { C.filled = 0u; C.consumed[0] = C.consumed[1] = C.consumed[2] = 0u; }
    __syncthreads();
    const unsigned char* src = base + (size_t)blockIdx.x * nslots * SLOT;
    if (wv == 3) {   // ---- loader ----
        __builtin_amdgcn_s_setprio(1);
        for (int s = 0; s < nslots; ++s) {
            if (s >= RING) {
                const unsigned need = (unsigned)(s - RING + 1);
                while (__hip_atomic_load(&C.consumed[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need ||
                       __hip_atomic_load(&C.consumed[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need ||
                       __hip_atomic_load(&C.consumed[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
            }
            const unsigned char* g = src + (size_t)s * SLOT + lane * 16;
            const unsigned off = (unsigned)((s % RING) * SLOT);
#pragma unroll
            for (int p = 0; p < PIECES; ++p) glds16<NT>(g + p * 1024, off + p * 1024u);
            if (s >= LAG) {
                vm_wait<PIECES * LAG>();
                if (lane == 0) __hip_atomic_store(&C.filled, (unsigned)(s - LAG + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        vm_wait<0>();
        if (lane == 0) __hip_atomic_store(&C.filled, (unsigned)nslots, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    // ---- consumer: pieces wv, wv + 3, .. of every slot ----
    unsigned acc = 0;
    for (int s = 0; s < nslots; ++s) {
        while (__hip_atomic_load(&C.filled, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= (unsigned)s) __builtin_amdgcn_s_sleep(1);
        const unsigned char* slot = smem + (size_t)(s % RING) * SLOT;
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            if (p % 3 != wv) continue;
            const u32x4 b = *(const u32x4*)(slot + p * 1024 + lane * 16);
            acc = work<W>(b, acc);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&C.consumed[wv], (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    atomicAdd(&sink[(blockIdx.x * 4 + wv) & 1023], acc);
}

// the same bytes through registers: 256-thread workgroups, four 16-byte loads in flight per lane (what kernels_v9.h does, without the block arithmetic)
template <int W>
__global__ void __launch_bounds__(256) plain_kernel(const unsigned char* base, int nslots, unsigned* sink) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* src = base + (size_t)blockIdx.x * nslots * SLOT;
    unsigned acc = 0;
    // wave wv takes the pieces p with p % 4 == wv of every slot; in the engine's sum order nothing matters: sums of words are commutative
    for (int s = 0; s < nslots; ++s) {
        u32x4 b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) b[k] = __builtin_nontemporal_load((const u32x4*)(src + (size_t)s * SLOT + (4 * k + wv) * 1024 + lane * 16));
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = work<W>(b[k], acc);
    }
    atomicAdd(&sink[(blockIdx.x * 4 + wv) & 1023], acc);
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
    {
        std::vector<unsigned> h(bytes / 4);
        unsigned x = 12345u;
        for (size_t i = 0; i < h.size(); ++i) { x = x * 1664525u + 1013904223u; h[i] = x; }
        CK(hipMemcpy(buf, h.data(), bytes, hipMemcpyHostToDevice));
    }
    hipStream_t s; CK(hipStreamCreate(&s));
    const int lds = RING * SLOT + 64;
#define OPTIN(K) CK(hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, lds))
    OPTIN((engine_kernel<0, true, 2>)); OPTIN((engine_kernel<0, true, 3>)); OPTIN((engine_kernel<0, false, 2>)); OPTIN((engine_kernel<16, true, 2>));
    OPTIN((engine_kernel<0, true, 1>));
    printf("us per launch (TB/s), hipGraph of <= 16 launches over different windows of a 2 GiB buffer; slots of 16 KiB, ring of 8, 1 loader + 3 consumer waves\n");
    const double sizes_mb[] = {9.4, 28.3, 50.7, 107.5, 430.0};
    for (int nwg : {256, 512, 1024}) {
        printf("-- %d workgroups (%d per CU; LDS %d KB each)\n", nwg, nwg / 256, lds / 1024);
        for (double mb : sizes_mb) {
            const int nslots = std::max(1, (int)(mb * 1e6 / SLOT / nwg + 0.5));
            const size_t win = (size_t)nwg * nslots * SLOT;
            const size_t stride = ((win + (16 << 20)) >> 20) << 20;
            const int nl = (int)std::max<size_t>(1, std::min<size_t>(16, bytes / stride));
            printf("%6.1f MB (%3d slots per workgroup, %2d launches):", (double)win / 1e6, nslots, nl);
#define ENG(Wv, NTv, LAGv) do { if (nwg == 256 || RING * SLOT * 2 <= 160 * 1024) { \
            double us = run_graph([&](int i, hipStream_t st) { hipLaunchKernelGGL((engine_kernel<Wv, NTv, LAGv>), dim3(nwg), dim3(256), lds, st, buf + (size_t)i * stride, nslots, sink); }, nl, s); \
            printf("  engine W=%-2d %s lag %d %7.2f us (%.2f)", Wv, NTv ? "nt " : "def", LAGv, us, (double)win / us / 1e6); } } while (0)
#define PLAIN(Wv) do { double us = run_graph([&](int i, hipStream_t st) { hipLaunchKernelGGL((plain_kernel<Wv>), dim3(nwg), dim3(256), 0, st, buf + (size_t)i * stride, nslots, sink); }, nl, s); \
            printf("  | registers W=%-2d %7.2f us (%.2f)", Wv, us, (double)win / us / 1e6); } while (0)
            ENG(0, true, 2); ENG(0, true, 3); ENG(0, true, 1); ENG(0, false, 2); ENG(16, true, 2);
            PLAIN(0); PLAIN(16);
            printf("\n");
        }
    }
    {   // checksum over one window: a stale LDS read would show
        const int nslots = 12;
        std::vector<unsigned> a(1024), b(1024);
        CK(hipMemset(sink, 0, 4096));
        hipLaunchKernelGGL((engine_kernel<0, true, 2>), dim3(256), dim3(256), lds, s, buf, nslots, sink);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(a.data(), sink, 4096, hipMemcpyDeviceToHost));
        CK(hipMemset(sink, 0, 4096));
        hipLaunchKernelGGL((plain_kernel<0>), dim3(256), dim3(256), 0, s, buf, nslots, sink);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(b.data(), sink, 4096, hipMemcpyDeviceToHost));
        unsigned ta = 0, tb = 0;
        for (unsigned v : a) ta += v;
        for (unsigned v : b) tb += v;
        printf("checksum over the same %d slots: engine %u, registers %u -> %s\n", 256 * nslots, ta, tb, ta == tb ? "equal" : "DIFFERENT (stale LDS read?)");
    }
    return 0;
}
