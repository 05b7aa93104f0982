// TEST INFRASTRUCTURE ONLY.  Functional CPU emulation of the small HIP subset this repo's kernels use, so that the
// kernel *logic* (indexing, reductions, bit unpacking, numerics contract) can be exercised by `-m "not gpu"` tests
// in a container without a GPU.  A workgroup's threads run as cooperative fibers (ucontext); wave collectives
// (__shfl*, __ballot) and __syncthreads() are rendezvous points.  It is compiled ONLY into
// tests/emu/_build/libctransformers_emu.so (g++ -DCT_EMU); the product library (hipcc) contains none of it and
// the product loader never falls back to it.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <math.h>
#include <functional>
#include <immintrin.h>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

extern dim3 threadIdx, blockIdx, blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

namespace emu {
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void syncthreads();
uint64_t wave_exchange(uint64_t v, int src_lane);
uint64_t wave_ballot(int pred);
void wave_sync();                  // rendezvous of the live lanes of the calling wave (wave-private LDS exchange)
void spin_yield();                 // a polling loop hands the CPU to the other fibers of the block
void dyn_smem_reserve(size_t n);   // dynamic LDS of the next launch (zero-filled per workgroup is NOT guaranteed, as on hardware)
unsigned char* dyn_smem();
}  // namespace emu

static inline void __syncthreads() { emu::syncthreads(); }

template <class T> static inline T emu_shfl_any(T v, int src) {
    static_assert(sizeof(T) <= 8, "shuffle payload too large");
    uint64_t u = 0;
    memcpy(&u, &v, sizeof(T));
    u = emu::wave_exchange(u, src);
    T r;
    memcpy(&r, &u, sizeof(T));
    return r;
}
template <class T> static inline T __shfl(T v, int src) { return emu_shfl_any(v, src & 63); }
template <class T> static inline T __shfl_xor(T v, int mask) {
    unsigned tid = threadIdx.x + threadIdx.y * blockDim.x + threadIdx.z * blockDim.x * blockDim.y;
    return emu_shfl_any(v, (int)((tid & 63) ^ (unsigned)mask) & 63);
}
static inline unsigned long long __ballot(int pred) { return emu::wave_ballot(pred); }

static inline int __builtin_amdgcn_sdot4(int a, int b, int c, bool) {
    int8_t x[4], y[4];
    memcpy(x, &a, 4);
    memcpy(y, &b, 4);
    return c + x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
}
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }

// ---- minimal hip host API shim (host memory stands in for device memory) --------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
typedef struct { double t; } *hipEvent_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetDeviceCount(int* n) { const char* e = getenv("CT_EMU_DEVICES"); *n = e && atoi(e) > 0 ? atoi(e) : 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256); return *p ? 0 : 1; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
template <class P> static inline hipError_t hipMalloc(P** p, size_t n) { return hipMalloc((void**)p, n); }
template <class P> static inline hipError_t hipHostMalloc(P** p, size_t n, unsigned f = 0) { return hipHostMalloc((void**)p, n, f); }

#define CT_LAUNCH(kernel, grid, block, stream, ...) emu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })
#define CT_LAUNCH_DYN(kernel, grid, block, smem, stream, ...) \
    do { emu::dyn_smem_reserve(smem); emu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); }); } while (0)
#define CT_DYN_SMEM(name) unsigned char* name = emu::dyn_smem()
#define CT_SMEM_OPTIN(fn, bytes) ((void)(fn), (bytes) <= 160 * 1024)

constexpr bool kConcurrentLaunches = false;   // the fiber scheduler is one-launch-at-a-time: host code must not launch from several threads
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
enum { hipDeviceAttributeMultiprocessorCount = 0 };
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) {   // emulate a 4-CU chip (CT_EMU_CUS: another count, for launch shapes that depend on it)
    const char* e = getenv("CT_EMU_CUS");
    *v = (e && atoi(e) > 0) ? atoi(e) : 4;
    return hipSuccess;
}
