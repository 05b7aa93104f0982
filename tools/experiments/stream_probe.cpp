// What read bandwidth do 256 x 1024-thread workgroups reach on 1152-byte records (1 KiB of nibbles + 128 B of headers),
// as a function of how the header is fetched and how deep each wave's register ring is?  (round 2: the generation-7 mat-vec
// streamed at 3.3 TB/s although its loads were issued early and counted exactly; this isolates the memory side.)
//   mode 0: body only (1 KiB per wave instruction)                      -> ceiling of the pattern
//   mode 1: body + header replicated over the 8 lanes of a slot (64 active lanes, 8 distinct 16-B pieces)   [generation 7]
//   mode 2: body + header by lanes 0..7 only (exec-masked)
//   mode 3: body + header as ONE 4-byte load per lane (lane p*8+g reads dword g&3 of slot p: 128 B, every lane active)
// units: a wave streams `spu` consecutive records (one unit), then jumps to its next unit (interleaved over waves/CUs).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int D>
__global__ void __launch_bounds__(1024) probe(const unsigned char* base, int n_units, int spu, unsigned* sink) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int stride = gridDim.x * 16, first = blockIdx.x + gridDim.x * wv;
    const int nu = first < n_units ? (n_units - first + stride - 1) / stride : 0;
    const size_t unit_bytes = (size_t)spu * 1152;
    int left = nu * spu, s = 0, it = first;
    const unsigned char* p = base + (size_t)(nu > 0 ? first : 0) * unit_bytes;
    u32x4 body[D], hdr[D];
    unsigned h1[D];
    auto issue = [&](int k) __attribute__((always_inline)) {
        body[k] = __builtin_nontemporal_load((const u32x4*)(p + 128 + lane * 16));
        if (MODE == 1) hdr[k] = __builtin_nontemporal_load((const u32x4*)(p + (lane >> 3) * 16));
        if (MODE == 2) { if (lane < 8) hdr[k] = __builtin_nontemporal_load((const u32x4*)(p + lane * 16)); }
        if (MODE == 3) h1[k] = __builtin_nontemporal_load((const unsigned*)(p + (lane >> 3) * 16 + (lane & 3) * 4));
        if (left > 1) { --left; p += 1152; if (++s == spu) { s = 0; it += stride; p = base + (size_t)it * unit_bytes; } }
    };
#pragma unroll
    for (int k = 0; k < D; ++k) issue(k);
    unsigned acc = 0;
    const int total = nu * spu;
    for (int st = 0; st < total; st += D) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            unsigned v = body[k][0] ^ body[k][1] ^ body[k][2] ^ body[k][3];
            if (MODE == 1 || MODE == 2) v ^= hdr[k][0] ^ hdr[k][3];
            if (MODE == 3) v ^= h1[k];
            asm volatile("" : "+v"(v));
            acc += v;
            issue(k);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int D> double run(const unsigned char* buf, size_t bytes, int spu, unsigned* sink, int grid) {
    const int n_units = (int)(bytes / ((size_t)spu * 1152));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<MODE, D>), dim3(grid), dim3(1024), 0, 0, buf, n_units, spu, sink);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<MODE, D>), dim3(grid), dim3(1024), 0, 0, buf, n_units, spu, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double used = (double)n_units * spu * (MODE == 0 ? 1024.0 : 1152.0);
    return used * reps / (ms * 1e-3) / 1e12;
}

int main() {
    const size_t bytes = (size_t)1 << 30;   // 1 GiB: far beyond the 256 MB memory-side cache
    unsigned char* buf; unsigned* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, bytes));
    for (int spu : {4, 16, 64}) {
        printf("spu=%2d  body-only D2 %.2f D4 %.2f | replicated hdr D2 %.2f D4 %.2f | 8-lane hdr D2 %.2f D4 %.2f | dword hdr D2 %.2f D4 %.2f  TB/s\n", spu,
               run<0, 2>(buf, bytes, spu, sink, 256), run<0, 4>(buf, bytes, spu, sink, 256),
               run<1, 2>(buf, bytes, spu, sink, 256), run<1, 4>(buf, bytes, spu, sink, 256),
               run<2, 2>(buf, bytes, spu, sink, 256), run<2, 4>(buf, bytes, spu, sink, 256),
               run<3, 2>(buf, bytes, spu, sink, 256), run<3, 4>(buf, bytes, spu, sink, 256));
    }
    printf("grid 512: body-only D4 spu4 %.2f, replicated D4 spu4 %.2f\n", run<0, 4>(buf, bytes, 4, sink, 512), run<1, 4>(buf, bytes, 4, sink, 512));
    return 0;
}
