// Device dialect shared by every kernel in this directory.  Product build: hipcc --offload-arch=gfx950 (CDNA4,
// wave64).  The only other build is the test-only CPU emulation (g++ -DCT_EMU, tests/emu/) which re-implements the
// handful of names below so kernel logic can be checked without a GPU; nothing of it is linked into the product.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef CT_EMU
#include "emu_runtime.h"
#else
#include <hip/hip_runtime.h>
#define CT_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, (grid), (block), 0, (stream), __VA_ARGS__)
// dynamic LDS (more than the 64 KB a static __shared__ declaration may take): size in bytes at launch, CT_DYN_SMEM in the kernel
#define CT_LAUNCH_DYN(kernel, grid, block, smem, stream, ...) hipLaunchKernelGGL(kernel, (grid), (block), (smem), (stream), __VA_ARGS__)
#define CT_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define CT_SMEM_OPTIN(fn, bytes) \
    (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)) == hipSuccess)
constexpr bool kConcurrentLaunches = true;   // host threads may launch on different devices at the same time (pipeline.cc: stages load concurrently)
#endif

#define DEV __device__ __forceinline__
constexpr int kWave = 64;  // CDNA wavefront width; hard-coded on purpose (MI355X_MICROARCH.md "wave = 64 not 32")

// ---- fp16 bit conversions (IEEE binary16, round-to-nearest-even; == x86 F16C used by the reference build) -----------
#ifdef CT_EMU
static inline uint16_t f32_to_f16_bits(float f) { return (uint16_t)_cvtss_sh(f, 0); }
static inline float f16_bits_to_f32(uint16_t h) { return _cvtsh_ss(h); }
#else
__host__ __device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    // Pin the f32 value in a VGPR first: without this hipcc (ROCm 7.2) folds `(half)(a*b)` / `(half)fma(a,b,c)` into
    // v_fma_mixlo_f16, which rounds the EXACT product once to fp16 instead of f32-then-fp16 like the reference's
    // F16C conversion; the two differ exactly when the f32 result is an fp16 tie (seen on hardware: a softmax
    // probability 0x3cd9b000 came out 0x26cd instead of 0x26ce).  -ffp-contract=off does not stop this fold.
    asm volatile("" : "+v"(f));
#endif
    _Float16 h = (_Float16)f;  // v_cvt_f16_f32 on device, RNE on both sides
    return __builtin_bit_cast(uint16_t, h);
}
__host__ __device__ __forceinline__ float f16_bits_to_f32(uint16_t b) {
    return (float)__builtin_bit_cast(_Float16, b);
}
#endif

DEV int lane_id() { return (int)(threadIdx.x & 63); }
DEV int wave_id() { return (int)(threadIdx.x >> 6); }

template <class T> DEV T wave_sum(T v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
DEV float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

DEV int sdot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }
// Eight independent dot4 with zero accumulators in ONE statement: hipcc lowers sdot4(a, b, 0) to `v_mov v, 0` + the VOP2
// accumulate form `v_dot4c`; the VOP3P form takes the constant directly.  The trailing `s_nop 2` is part of the contract: on
// gfx940+ a DOT result needs 3 wait states before a non-DOT VALU may read it, and hipcc pads nothing inside or right after an
// asm statement it cannot see into (found the hard way: a lone asm dot4 feeding v_mul_i32_i24 read stale registers).
#ifdef CT_EMU
static inline void dot4x8(int (&d)[8], const int (&w)[8], const int (&a)[8]) {
    for (int i = 0; i < 8; ++i) d[i] = __builtin_amdgcn_sdot4(w[i], a[i], 0, false);
}
// d[i] = dot4(w[i], a[i]) + dot4(bias, a[i])   (Q6_K: bias = 0xE0E0E0E0 = -32 per byte)
static inline void dot4x8_bias(int (&d)[8], const int (&w)[8], const int (&a)[8], int bias) {
    for (int i = 0; i < 8; ++i) d[i] = __builtin_amdgcn_sdot4(w[i], a[i], __builtin_amdgcn_sdot4(bias, a[i], 0, false), false);
}
// four independent dot4 (kernels_v9.h, 32-block types): d[i] = dot4(w[i], a[i]) [+ c[i]]
static inline void dot4x4(int (&d)[4], const int (&w)[4], const int (&a)[4]) {
    for (int i = 0; i < 4; ++i) d[i] = __builtin_amdgcn_sdot4(w[i], a[i], 0, false);
}
static inline void dot4x4_add(int (&d)[4], const int (&w)[4], const int (&a)[4], const int (&c)[4]) {
    for (int i = 0; i < 4; ++i) d[i] = __builtin_amdgcn_sdot4(w[i], a[i], c[i], false);
}
// d[i] = dot4(w[i], a[i]) + c[i]
static inline void dot4x8_add(int (&d)[8], const int (&w)[8], const int (&a)[8], const int (&c)[8]) {
    for (int i = 0; i < 8; ++i) d[i] = __builtin_amdgcn_sdot4(w[i], a[i], c[i], false);
}
// d[i] = dot4(w[i], a[i]) + c, c wave-uniform (VOP3P form with the addend in a scalar register: one instruction per dot)
static inline void dot4x8_acc(int (&d)[8], const int (&w)[8], const int (&a)[8], int c) {
    for (int i = 0; i < 8; ++i) d[i] = __builtin_amdgcn_sdot4(w[i], a[i], c, false);
}
#else
DEV void dot4x8(int (&d)[8], const int (&w)[8], const int (&a)[8]) {
    asm("v_dot4_i32_i8 %0, %8, %16, 0\n\tv_dot4_i32_i8 %1, %9, %17, 0\n\tv_dot4_i32_i8 %2, %10, %18, 0\n\tv_dot4_i32_i8 %3, %11, %19, 0\n\t"
        "v_dot4_i32_i8 %4, %12, %20, 0\n\tv_dot4_i32_i8 %5, %13, %21, 0\n\tv_dot4_i32_i8 %6, %14, %22, 0\n\tv_dot4_i32_i8 %7, %15, %23, 0\n\t"
        "s_nop 2"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
        : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]),
          "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
}
DEV void dot4x8_bias(int (&d)[8], const int (&w)[8], const int (&a)[8], int bias) {
    asm("v_dot4_i32_i8 %0, %24, %16, 0\n\tv_dot4_i32_i8 %1, %24, %17, 0\n\tv_dot4_i32_i8 %2, %24, %18, 0\n\tv_dot4_i32_i8 %3, %24, %19, 0\n\t"
        "v_dot4_i32_i8 %4, %24, %20, 0\n\tv_dot4_i32_i8 %5, %24, %21, 0\n\tv_dot4_i32_i8 %6, %24, %22, 0\n\tv_dot4_i32_i8 %7, %24, %23, 0\n\t"
        "v_dot4_i32_i8 %0, %8, %16, %0\n\tv_dot4_i32_i8 %1, %9, %17, %1\n\tv_dot4_i32_i8 %2, %10, %18, %2\n\tv_dot4_i32_i8 %3, %11, %19, %3\n\t"
        "v_dot4_i32_i8 %4, %12, %20, %4\n\tv_dot4_i32_i8 %5, %13, %21, %5\n\tv_dot4_i32_i8 %6, %14, %22, %6\n\tv_dot4_i32_i8 %7, %15, %23, %7\n\t"
        "s_nop 2"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
        : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]),
          "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(bias));
}
DEV void dot4x4(int (&d)[4], const int (&w)[4], const int (&a)[4]) {
    asm("v_dot4_i32_i8 %0, %4, %8, 0\n\tv_dot4_i32_i8 %1, %5, %9, 0\n\tv_dot4_i32_i8 %2, %6, %10, 0\n\tv_dot4_i32_i8 %3, %7, %11, 0\n\ts_nop 2"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
        : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
}
DEV void dot4x4_add(int (&d)[4], const int (&w)[4], const int (&a)[4], const int (&c)[4]) {
    asm("v_dot4_i32_i8 %0, %4, %8, %12\n\tv_dot4_i32_i8 %1, %5, %9, %13\n\tv_dot4_i32_i8 %2, %6, %10, %14\n\tv_dot4_i32_i8 %3, %7, %11, %15\n\ts_nop 2"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
        : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]));
}
DEV void dot4x8_add(int (&d)[8], const int (&w)[8], const int (&a)[8], const int (&c)[8]) {
    asm("v_dot4_i32_i8 %0, %8, %16, %24\n\tv_dot4_i32_i8 %1, %9, %17, %25\n\tv_dot4_i32_i8 %2, %10, %18, %26\n\tv_dot4_i32_i8 %3, %11, %19, %27\n\t"
        "v_dot4_i32_i8 %4, %12, %20, %28\n\tv_dot4_i32_i8 %5, %13, %21, %29\n\tv_dot4_i32_i8 %6, %14, %22, %30\n\tv_dot4_i32_i8 %7, %15, %23, %31\n\t"
        "s_nop 2"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
        : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]),
          "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
          "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]));
}
DEV void dot4x8_acc(int (&d)[8], const int (&w)[8], const int (&a)[8], int c) {
    const int cs = __builtin_amdgcn_readfirstlane(c);
    asm("v_dot4_i32_i8 %0, %8, %16, %24\n\tv_dot4_i32_i8 %1, %9, %17, %24\n\tv_dot4_i32_i8 %2, %10, %18, %24\n\tv_dot4_i32_i8 %3, %11, %19, %24\n\t"
        "v_dot4_i32_i8 %4, %12, %20, %24\n\tv_dot4_i32_i8 %5, %13, %21, %24\n\tv_dot4_i32_i8 %6, %14, %22, %24\n\tv_dot4_i32_i8 %7, %15, %23, %24\n\t"
        "s_nop 2"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
        : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]),
          "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "s"(cs));
}
#endif
// 24-bit integer multiply (full rate; v_mul_lo_u32 is quarter rate).  All products on the hot path fit: |a|,|b| < 2^23.
#ifdef CT_EMU
static inline int mul24(int a, int b) { return a * b; }
static inline int uniform_int(int v) { return v; }
static inline uint32_t alignbit32(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (s & 31)); }
static inline uint32_t bfe32(uint32_t v, int off, int width) { return (v >> off) & ((1u << width) - 1u); }
#else
DEV uint32_t alignbit32(uint32_t hi, uint32_t lo, uint32_t s) { return __builtin_amdgcn_alignbit(hi, lo, s); }
DEV uint32_t bfe32(uint32_t v, int off, int width) { return __builtin_amdgcn_ubfe(v, off, width); }
DEV int mul24(int a, int b) { return __mul24(a, b); }
// Tell the compiler a value is wave-uniform (it is: derived from the wave index) so it lives in an SGPR and branches on
// it are scalar (guide T20: anything derived from threadIdx is divergent to the compiler).
DEV int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif

// ---- round-4 step primitives (measured issue costs, tools/experiments/gen_valu_rate.py on MI355X, cycles per wave-instruction and SIMD at
// four waves per SIMD: v_and / v_lshrrev / v_add_u32 / v_mul_f32 / v_fma_f32 / v_fmac_f32 / v_mov 2.0; everything with a DPP or SDWA
// modifier, every other VOP3 (v_bfe, v_mad_i32_i24, v_add3, v_dot4, v_perm, v_alignbit, v_lshl_add, v_cndmask with an SGPR mask), v_mul_i32_i24, the
// conversions, v_max_f32 and the 64-bit forms 3.1-3.3; v_rcp 6.1) -------------------------------------------------------------------------------
// mad24: a * b + c on 24-bit operands as ONE v_mad_i32_i24 (hipcc re-associates mul24 + add chains into v_mul x 2 + v_add3: three
// instructions of the 3.3-cycle class for two products instead of two).
// lshr64_lo: low word of ((hi:lo) >> sh), sh = 0..63 per lane: one v_lshrrev_b64 when hi:lo sit in consecutive registers.
// lane_up4_add: v + (the value of lane + 4 inside its row of 16; lanes 12..15 of a row add zero) — ONE v_add_u32_dpp row_shl:4.
// mix_mul_f16lo / mix_mulneg_f16hi: fp16(h & 0xFFFF) * f resp. -(fp16(h >> 16) * f) rounded once to f32: v_fma_mix_f32 converts the
// half operand on the fly (one instruction for conversion + multiply; the addend -0.0 keeps the product's sign of zero).
#ifdef CT_EMU
static inline int mad24(int a, int b, int c) { return a * b + c; }
static inline uint32_t lshr64_lo(uint32_t hi, uint32_t lo, int sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 63)); }
static inline int lane_up4_add(int v) {
    const int lane = (int)(threadIdx.x & 63u);
    const int o = __shfl(v, (lane & 15) + 4 < 16 ? lane + 4 : lane);
    return v + ((lane & 15) + 4 < 16 ? o : 0);
}
static inline float mix_mul_f16lo(uint32_t h, float f) { return _cvtsh_ss((uint16_t)(h & 0xFFFFu)) * f; }
static inline float mix_mulneg_f16hi(uint32_t h, float f) { return -f * _cvtsh_ss((uint16_t)(h >> 16)); }
#else
DEV int mad24(int a, int b, int c) { int r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
DEV uint32_t lshr64_lo(uint32_t hi, uint32_t lo, int sh) { return (uint32_t)((((unsigned long long)hi << 32) | lo) >> (sh & 63)); }
DEV int lane_up4_add(int v) { return v + __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0xF, true); }   // row_shl:4, bound_ctrl
DEV float mix_mul_f16lo(uint32_t h, float f) {
    float r;
    const uint32_t nz = 0x80000000u;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(f), "s"(nz));
    return r;
}
DEV float mix_mulneg_f16hi(uint32_t h, float f) {
    float r;
    const uint32_t nz = 0x80000000u;
    asm("v_fma_mix_f32 %0, %1, -%2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(f), "s"(nz));
    return r;
}
#endif

// ---- 16-byte streaming load (weights are read once per token: non-temporal, `nt-weights` row of the guide) ----------
#ifdef CT_EMU
struct u32x4 {
    uint32_t v[4];
    uint32_t operator[](int i) const { return v[i]; }
    uint32_t& operator[](int i) { return v[i]; }
};
static inline u32x4 ld_stream16(const void* p) { u32x4 r; memcpy(&r, p, 16); return r; }
static inline u32x4 ld16(const void* p) { u32x4 r; memcpy(&r, p, 16); return r; }
#else
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
DEV u32x4 ld_stream16(const void* p) { return __builtin_nontemporal_load((const u32x4*)p); }
DEV u32x4 ld16(const void* p) { return *(const u32x4*)p; }
#endif

// fma8_hh / fma8_hf: acc[k] = fma(fp16 element k of a, fp16 element k of b | f[k], acc[k]) for the eight halves of a 16-byte operand, as eight
// v_fma_mix_f32: the half operands are converted by the instruction itself (exactly, as v_cvt_f32_f16 does) and the fma rounds once — the
// same floats as conversions followed by fmaf, in 8 instructions instead of 16 + 8 (hh) / 8 + 8 (hf).  (hipcc folds some of these itself,
// not all: the decode attention's V*P loop ran eight conversions + four packed fmas per 32 positions.)
#ifdef CT_EMU
static inline void fma8_hh(float (&acc)[8], const u32x4& a, const u32x4& b) {
    for (int k = 0; k < 4; ++k) {
        acc[2 * k] = fmaf(_cvtsh_ss((uint16_t)(a[k] & 0xFFFFu)), _cvtsh_ss((uint16_t)(b[k] & 0xFFFFu)), acc[2 * k]);
        acc[2 * k + 1] = fmaf(_cvtsh_ss((uint16_t)(a[k] >> 16)), _cvtsh_ss((uint16_t)(b[k] >> 16)), acc[2 * k + 1]);
    }
}
static inline void fma8_hf(float (&acc)[8], const u32x4& a, const float (&f)[8]) {
    for (int k = 0; k < 4; ++k) {
        acc[2 * k] = fmaf(_cvtsh_ss((uint16_t)(a[k] & 0xFFFFu)), f[2 * k], acc[2 * k]);
        acc[2 * k + 1] = fmaf(_cvtsh_ss((uint16_t)(a[k] >> 16)), f[2 * k + 1], acc[2 * k + 1]);
    }
}
#else
DEV void fma8_hh(float (&acc)[8], const u32x4& a, const u32x4& b) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(acc[2 * k]) : "v"(a[k]), "v"(b[k]));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * k + 1]) : "v"(a[k]), "v"(b[k]));
    }
}
DEV void fma8_hf(float (&acc)[8], const u32x4& a, const float (&f)[8]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[2 * k]) : "v"(a[k]), "v"(f[2 * k]));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[2 * k + 1]) : "v"(a[k]), "v"(f[2 * k + 1]));
    }
}
#endif


// fma4_hf: the same for the four halves of an 8-byte operand
#ifdef CT_EMU
static inline void fma4_hf(float (&acc)[4], uint32_t a0, uint32_t a1, const float (&f)[4]) {
    acc[0] = fmaf(_cvtsh_ss((uint16_t)(a0 & 0xFFFFu)), f[0], acc[0]);
    acc[1] = fmaf(_cvtsh_ss((uint16_t)(a0 >> 16)), f[1], acc[1]);
    acc[2] = fmaf(_cvtsh_ss((uint16_t)(a1 & 0xFFFFu)), f[2], acc[2]);
    acc[3] = fmaf(_cvtsh_ss((uint16_t)(a1 >> 16)), f[3], acc[3]);
}
#else
DEV void fma4_hf(float (&acc)[4], uint32_t a0, uint32_t a1, const float (&f)[4]) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[0]) : "v"(a0), "v"(f[0]));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[1]) : "v"(a0), "v"(f[1]));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[2]) : "v"(a1), "v"(f[2]));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[3]) : "v"(a1), "v"(f[3]));
}
#endif

// ---- int8 matrix core: D[16][16] += A[16][32] * B[32][16] (v_mfma_i32_16x16x32_i8) ---------------------------------------
// Operand layout (checked on hardware by tools/experiments/mfma_i8_layout.cpp): lane i gives A[i & 15][8q .. 8q+7] and
// B[8q .. 8q+7][i & 15] (q = i >> 4) as eight int8 in one 64-bit operand, and holds D[4q + j][i & 15] in register j.
#ifdef CT_EMU
struct i32x4 {
    int v[4];
    int operator[](int i) const { return v[i]; }
    int& operator[](int i) { return v[i]; }
};
static inline i32x4 mfma_i8_16x16x32(uint64_t a, uint64_t b, i32x4 c) {
    const int lane = (int)(threadIdx.x & 63), q = lane >> 4, n = lane & 15;
    uint64_t bk[4];
    for (int kc = 0; kc < 4; ++kc) bk[kc] = emu_shfl_any(b, n + 16 * kc);
    for (int j = 0; j < 4; ++j) {
        for (int kc = 0; kc < 4; ++kc) {
            const uint64_t am = emu_shfl_any(a, 4 * q + j + 16 * kc);
            for (int e = 0; e < 8; ++e) c[j] += (int)(int8_t)(am >> (8 * e)) * (int)(int8_t)(bk[kc] >> (8 * e));
        }
    }
    return c;
}
// 16x16x64 (gfx950): 16 bytes per lane and operand.  Only the pairing matters to the callers: byte e of lane (n, q)'s A
// meets byte e of lane (m, q)'s B, D[n][m] sums all 64 products and keeps the 16x16 result map (also hardware-checked).
static inline i32x4 mfma_i8_16x16x64(u32x4 a, u32x4 b, i32x4 c) {
    const int lane = (int)(threadIdx.x & 63), q = lane >> 4, n = lane & 15;
    const uint64_t a01 = (uint64_t)a[0] | ((uint64_t)a[1] << 32), a23 = (uint64_t)a[2] | ((uint64_t)a[3] << 32);
    const uint64_t b01 = (uint64_t)b[0] | ((uint64_t)b[1] << 32), b23 = (uint64_t)b[2] | ((uint64_t)b[3] << 32);
    uint64_t bk[4][2];
    for (int kc = 0; kc < 4; ++kc) { bk[kc][0] = emu_shfl_any(b01, n + 16 * kc); bk[kc][1] = emu_shfl_any(b23, n + 16 * kc); }
    for (int j = 0; j < 4; ++j) {
        for (int kc = 0; kc < 4; ++kc) {
            const uint64_t am[2] = {emu_shfl_any(a01, 4 * q + j + 16 * kc), emu_shfl_any(a23, 4 * q + j + 16 * kc)};
            for (int w = 0; w < 2; ++w)
                for (int e = 0; e < 8; ++e) c[j] += (int)(int8_t)(am[w] >> (8 * e)) * (int)(int8_t)(bk[kc][w] >> (8 * e));
        }
    }
    return c;
}
// v_mfma_i32_4x4x4_16b_i8: sixteen independent 4x4 blocks with K = 4.  Lane 4b + m gives row m of block b's A and column m of
// its B as four int8 each, and holds D_b[i][m] = C + sum_k A_b[i][k] * B_b[k][m] in register i (checked on hardware by
// tools/experiments/mfma_i8_4x4x4_layout.cpp, together with the float-addend form C = 0x4B400000).
static inline i32x4 mfma_i8_4x4x4(int a, int b, i32x4 c) {
    const int lane = (int)(threadIdx.x & 63), blk = lane >> 2;
    for (int i = 0; i < 4; ++i) {
        const int am = emu_shfl_any(a, 4 * blk + i);
        for (int e = 0; e < 4; ++e) c[i] += (int)(int8_t)(am >> (8 * e)) * (int)(int8_t)(b >> (8 * e));
    }
    return c;
}
// two u16 lanes of `a` times the two u16 lanes of `s` (low 16 bits each): v_pk_mul_lo_u16
static inline uint32_t pk_mul_u16(uint32_t a, uint32_t s) {
    return (((a & 0xFFFFu) * (s & 0xFFFFu)) & 0xFFFFu) | (((a >> 16) * (s >> 16)) << 16);
}
#else
typedef int i32x4 __attribute__((ext_vector_type(4)));
DEV i32x4 mfma_i8_16x16x32(uint64_t a, uint64_t b, i32x4 c) {
    return __builtin_amdgcn_mfma_i32_16x16x32_i8((long)a, (long)b, c, 0, 0, 0);
}
DEV i32x4 mfma_i8_16x16x64(u32x4 a, u32x4 b, i32x4 c) {
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), c, 0, 0, 0);
}
DEV i32x4 mfma_i8_4x4x4(int a, int b, i32x4 c) { return __builtin_amdgcn_mfma_i32_4x4x4i8(a, b, c, 0, 0, 0); }
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
DEV uint32_t pk_mul_u16(uint32_t a, uint32_t s) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, s));
}
#endif

// ---- f16 matrix core on INTEGER-valued operands (kernels_pg.h) -------------------------------------------------------------
// v_mfma_f32_16x16x32_f16 / v_mfma_f32_16x16x16_f16 with every operand an integer that fp16 holds exactly: all products and
// partial sums are integers below 2^24, so the f32 result is exact whatever the internal summation order — it equals
// (float) of the integer sum (checked on hardware by tools/experiments/mfma_f16_exact.cpp, also chained through C).
// Pairing (all the callers rely on): half e of lane (n, g)'s A meets half e of lane (m, g)'s B (n, m = lane & 15, g = lane >> 4);
// lane (m, g) holds D[4g + j][m] in register j.
#ifdef CT_EMU
struct f32x4 {
    float v[4];
    float operator[](int i) const { return v[i]; }
    float& operator[](int i) { return v[i]; }
};
static inline double emu_h2_dot(uint32_t a, uint32_t b) {
    return (double)f16_bits_to_f32((uint16_t)(a & 0xFFFF)) * (double)f16_bits_to_f32((uint16_t)(b & 0xFFFF)) +
           (double)f16_bits_to_f32((uint16_t)(a >> 16)) * (double)f16_bits_to_f32((uint16_t)(b >> 16));
}
static inline f32x4 mfma_f16_16x16x16(uint64_t a, uint64_t b, f32x4 c) {
    const int lane = (int)(threadIdx.x & 63), g = lane >> 4, n = lane & 15;
    uint64_t bk[4];
    for (int kc = 0; kc < 4; ++kc) bk[kc] = emu_shfl_any(b, n + 16 * kc);
    for (int j = 0; j < 4; ++j) {
        double s = (double)c[j];
        for (int kc = 0; kc < 4; ++kc) {
            const uint64_t am = emu_shfl_any(a, 4 * g + j + 16 * kc);
            s += emu_h2_dot((uint32_t)am, (uint32_t)bk[kc]) + emu_h2_dot((uint32_t)(am >> 32), (uint32_t)(bk[kc] >> 32));
        }
        c[j] = (float)s;
    }
    return c;
}
static inline f32x4 mfma_f16_16x16x32(u32x4 a, u32x4 b, f32x4 c) {
    const uint64_t a01 = (uint64_t)a[0] | ((uint64_t)a[1] << 32), a23 = (uint64_t)a[2] | ((uint64_t)a[3] << 32);
    const uint64_t b01 = (uint64_t)b[0] | ((uint64_t)b[1] << 32), b23 = (uint64_t)b[2] | ((uint64_t)b[3] << 32);
    return mfma_f16_16x16x16(a23, b23, mfma_f16_16x16x16(a01, b01, c));   // exact integer sums: any grouping gives the same f32
}
static inline uint32_t emu_pack_h2(float lo, float hi) { return (uint32_t)f32_to_f16_bits(lo) | ((uint32_t)f32_to_f16_bits(hi) << 16); }
// packed fp16: a * b + c with one rounding, a * b
static inline uint32_t pk_fma_f16(uint32_t a, uint32_t b, uint32_t c) {
    const double lo = (double)f16_bits_to_f32((uint16_t)a) * (double)f16_bits_to_f32((uint16_t)b) + (double)f16_bits_to_f32((uint16_t)c);
    const double hi = (double)f16_bits_to_f32((uint16_t)(a >> 16)) * (double)f16_bits_to_f32((uint16_t)(b >> 16)) + (double)f16_bits_to_f32((uint16_t)(c >> 16));
    return emu_pack_h2((float)lo, (float)hi);   // callers' results are small integers: exact in float and in fp16
}
static inline uint32_t pk_mul_f16(uint32_t a, uint32_t b) { return pk_fma_f16(a, b, 0u); }
static inline uint32_t h2_from_int(int v) { return emu_pack_h2((float)v, (float)v); }   // |v| <= 2048: exact
#else
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
DEV f32x4 mfma_f16_16x16x32(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
DEV f32x4 mfma_f16_16x16x16(uint64_t a, uint64_t b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
}
DEV uint32_t pk_fma_f16(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), __builtin_bit_cast(f16x2, c)));
}
DEV uint32_t pk_mul_f16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, a) * __builtin_bit_cast(f16x2, b)); }
DEV uint32_t h2_from_int(int v) { const _Float16 h = (_Float16)(short)v; const f16x2 r = {h, h}; return __builtin_bit_cast(uint32_t, r); }
#endif

// ---- 32x32 matrix cores of the order-free prompt kernels (kernels_mm8.h) -------------------------------------------------------------
// v_mfma_i32_32x32x32_i8: D[32][32] (+)= A[32][32] * B[32][32], 16 int8 per lane and operand.  Pairing (hardware-checked,
// tools/experiments/mfma_i8_32x32x32_probe.cpp): byte e of lane (n, c)'s A meets byte e of lane (m, c)'s B (n, m = lane & 31, c = lane >> 5;
// which k the hardware calls it is irrelevant to an integer sum); lane (m, h) holds D[(i & 3) + 8 (i >> 2) + 4 h][m] in register i.
// mfma_i8_32x32x32_bias: the same product on top of C = 0x4B400000 in every element — the result bits READ AS FLOATS are 1.5 * 2^23 + sum
// (|sum| < 2^22), the integer sum as an exact float without a conversion.
// v_mfma_f32_32x32x16_f16 on integer-valued halves (every product and partial sum an integer below 2^24: exact in any order): 8 halves per lane
// and operand, same pairing and result map.
#ifdef CT_EMU
struct i32x16 {
    int v[16];
    int operator[](int i) const { return v[i]; }
    int& operator[](int i) { return v[i]; }
};
struct f32x16 {
    float v[16];
    float operator[](int i) const { return v[i]; }
    float& operator[](int i) { return v[i]; }
};
static inline i32x16 mfma_i8_32x32x32_c(u32x4 a, u32x4 b, int c0) {
    const int lane = (int)(threadIdx.x & 63), m = lane & 31, h = lane >> 5;
    const uint64_t a01 = (uint64_t)a[0] | ((uint64_t)a[1] << 32), a23 = (uint64_t)a[2] | ((uint64_t)a[3] << 32);
    const uint64_t b01 = (uint64_t)b[0] | ((uint64_t)b[1] << 32), b23 = (uint64_t)b[2] | ((uint64_t)b[3] << 32);
    uint64_t bk[2][2];
    for (int c = 0; c < 2; ++c) { bk[c][0] = emu_shfl_any(b01, m + 32 * c); bk[c][1] = emu_shfl_any(b23, m + 32 * c); }
    i32x16 d;
    for (int i = 0; i < 16; ++i) {
        const int n = (i & 3) + 8 * (i >> 2) + 4 * h;
        int s = c0;
        for (int c = 0; c < 2; ++c) {
            const uint64_t am[2] = {emu_shfl_any(a01, n + 32 * c), emu_shfl_any(a23, n + 32 * c)};
            for (int w = 0; w < 2; ++w)
                for (int e = 0; e < 8; ++e) s += (int)(int8_t)(am[w] >> (8 * e)) * (int)(int8_t)(bk[c][w] >> (8 * e));
        }
        d[i] = s;
    }
    return d;
}
static inline i32x16 mfma_i8_32x32x32(u32x4 a, u32x4 b) { return mfma_i8_32x32x32_c(a, b, 0); }
static inline i32x16 mfma_i8_32x32x32_acc(u32x4 a, u32x4 b, i32x16 c) {
    const i32x16 p = mfma_i8_32x32x32_c(a, b, 0);
    for (int i = 0; i < 16; ++i) c[i] += p[i];
    return c;
}
static inline i32x16 mfma_i8_32x32x32_bias(u32x4 a, u32x4 b) { return mfma_i8_32x32x32_c(a, b, 0x4B400000); }
static inline f32x16 mfma_f16_32x32x16_acc(u32x4 a, u32x4 b, f32x16 c);
static inline f32x16 mfma_f16_32x32x16(u32x4 a, u32x4 b) {
    f32x16 z;
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    return mfma_f16_32x32x16_acc(a, b, z);
}
// (the sum of the sixteen products and C in double, rounded once: what the hardware does inside the instruction is not specified beyond "f32 accumulate" —
// callers are the order-free kernels, whose results do not depend on it beyond an f32 rounding)
static inline f32x16 mfma_f16_32x32x16_acc(u32x4 a, u32x4 b, f32x16 c0) {
    const int lane = (int)(threadIdx.x & 63), m = lane & 31, h = lane >> 5;
    const uint64_t a01 = (uint64_t)a[0] | ((uint64_t)a[1] << 32), a23 = (uint64_t)a[2] | ((uint64_t)a[3] << 32);
    const uint64_t b01 = (uint64_t)b[0] | ((uint64_t)b[1] << 32), b23 = (uint64_t)b[2] | ((uint64_t)b[3] << 32);
    uint64_t bk[2][2];
    for (int c = 0; c < 2; ++c) { bk[c][0] = emu_shfl_any(b01, m + 32 * c); bk[c][1] = emu_shfl_any(b23, m + 32 * c); }
    f32x16 d;
    for (int i = 0; i < 16; ++i) {
        const int n = (i & 3) + 8 * (i >> 2) + 4 * h;
        double s = (double)c0[i];
        for (int c = 0; c < 2; ++c) {
            const uint64_t am[2] = {emu_shfl_any(a01, n + 32 * c), emu_shfl_any(a23, n + 32 * c)};
            for (int w = 0; w < 2; ++w)
                for (int e = 0; e < 4; ++e)
                    s += (double)f16_bits_to_f32((uint16_t)(am[w] >> (16 * e))) * (double)f16_bits_to_f32((uint16_t)(bk[c][w] >> (16 * e)));
        }
        d[i] = (float)s;
    }
    return d;
}
#else
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
DEV i32x16 mfma_i8_32x32x32(u32x4 a, u32x4 b) {
    const i32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4v, a), __builtin_bit_cast(i32x4v, b), z, 0, 0, 0);
}
DEV i32x16 mfma_i8_32x32x32_acc(u32x4 a, u32x4 b, i32x16 c) {
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4v, a), __builtin_bit_cast(i32x4v, b), c, 0, 0, 0);
}
DEV i32x16 mfma_i8_32x32x32_bias(u32x4 a, u32x4 b) {
    constexpr int K = 0x4B400000;
    const i32x16 z = {K, K, K, K, K, K, K, K, K, K, K, K, K, K, K, K};
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4v, a), __builtin_bit_cast(i32x4v, b), z, 0, 0, 0);
}
DEV f32x16 mfma_f16_32x32x16(u32x4 a, u32x4 b) {
    typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, a), __builtin_bit_cast(f16x8v, b), z, 0, 0, 0);
}
DEV f32x16 mfma_f16_32x32x16_acc(u32x4 a, u32x4 b, f32x16 c) {
    typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, a), __builtin_bit_cast(f16x8v, b), c, 0, 0, 0);
}
#endif

// ---- f32 matrix core, K = 1 (kernels_pf.h: the block-scale products of the Q8_0 / Q4_0 prompt chunks) --------------------------------
// v_mfma_f32_4x4x1_16b_f32: sixteen independent rank-1 updates D_b[i][m] = C + A_b[i] * B_b[m] (the lane roles of the int8 4x4x4 form:
// lane 4b + m gives A_b[m] and B_b[m], holds D_b[i][m] in register i).  With C = 0 and operands whose product is exact in f32 (two
// fp16 values) the result is that product, bit for bit what v_mul_f32 gives.
#ifdef CT_EMU
static inline f32x4 mfma_f32_4x4x1(float a, float b, f32x4 c) {
    const int lane = (int)(threadIdx.x & 63), blk = lane >> 2;
    for (int i = 0; i < 4; ++i) {
        const float am = emu_shfl_any(a, 4 * blk + i);
        c[i] = fmaf(am, b, c[i]);
    }
    return c;
}
#else
DEV f32x4 mfma_f32_4x4x1(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
#endif

// ---- cross-lane moves without the LDS crossbar where the ISA allows it ----------------------------------------------
// __shfl_xor always lowers to ds_bpermute_b32 (address VGPR + LDS pipe, ~100 cycles dependent latency).  Butterflies
// inside a row of 16 lanes can use DPP modifiers instead: quad_perm for xor 1/2, row_half_mirror o quad_perm(3,2,1,0)
// for xor 4 (7-p then 3-q  ==  p^4), row_ror:8 for xor 8.  xor 16 uses ds_swizzle (no address register), xor 32 bpermute.
#ifdef CT_EMU
template <class T> static inline T lane_xor1(T v) { return __shfl_xor(v, 1); }
template <class T> static inline T lane_xor2(T v) { return __shfl_xor(v, 2); }
template <class T> static inline T lane_xor4(T v) { return __shfl_xor(v, 4); }
template <class T> static inline T lane_xor8(T v) { return __shfl_xor(v, 8); }
template <class T> static inline T lane_xor16(T v) { return __shfl_xor(v, 16); }
template <class T> static inline T lane_xor32(T v) { return __shfl_xor(v, 32); }
#else
template <int CTRL> DEV int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <class T, int CTRL> DEV T dpp_any(T v) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "dpp payload");
    if constexpr (sizeof(T) == 4) {
        int i = __builtin_bit_cast(int, v);
        i = dpp_i32<CTRL>(i);
        return __builtin_bit_cast(T, i);
    } else {
        long long l = __builtin_bit_cast(long long, v);
        int lo = (int)(l & 0xffffffffll), hi = (int)(l >> 32);
        lo = dpp_i32<CTRL>(lo);
        hi = dpp_i32<CTRL>(hi);
        l = ((long long)hi << 32) | (unsigned)lo;
        return __builtin_bit_cast(T, l);
    }
}
template <class T> DEV T lane_xor1(T v) { return dpp_any<T, 0xB1>(v); }                      // quad_perm [1,0,3,2]
template <class T> DEV T lane_xor2(T v) { return dpp_any<T, 0x4E>(v); }                      // quad_perm [2,3,0,1]
template <class T> DEV T lane_xor4(T v) { return dpp_any<T, 0x1B>(dpp_any<T, 0x141>(v)); }  // row_half_mirror, then quad_perm [3,2,1,0]
template <class T> DEV T lane_xor8(T v) { return dpp_any<T, 0x128>(v); }                     // row_ror:8
template <class T> DEV T lane_xor16(T v) {
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));  // bit mode: xor 16
    } else {
        return __shfl_xor(v, 16);
    }
}
template <class T> DEV T lane_xor32(T v) { return __shfl_xor(v, 32); }
#endif

// ---- generation-9 mat-vec primitives (kernels_v9.h) ---------------------------------------------------------------------------
// quad_bcast<K>: every lane of a quad (lanes 4q .. 4q+3) reads lane 4q + K's value — DPP quad_perm:[K,K,K,K], one VALU move that
// hipcc folds into the consuming VOP2 (v_fmac_f32_dpp) where it can.  wave_read_lane: v_readlane_b32 with a wave-uniform index.
// ld_stream8: 8-byte streaming load.  bfe_i32: sign-extending bit-field extract.
#ifdef CT_EMU
template <int K, class T> static inline T quad_bcast(T v) { return __shfl(v, (int)((threadIdx.x & 63u) & ~3u) + K); }
static inline float wave_read_lane(float v, int src) { return __shfl(v, src); }
struct u32x2 {
    uint32_t v[2];
    uint32_t operator[](int i) const { return v[i]; }
    uint32_t& operator[](int i) { return v[i]; }
};
static inline u32x2 ld_stream8(const void* p) { u32x2 r; memcpy(&r, p, 8); return r; }
static inline u32x2 ld8(const void* p) { u32x2 r; memcpy(&r, p, 8); return r; }
static inline uint32_t ld_stream4(const void* p) { uint32_t r; memcpy(&r, p, 4); return r; }
static inline int bfe_i32(uint32_t v, int off, int width) { return (int)(v << (32 - off - width)) >> (32 - width); }
static inline uint32_t pack_low_bytes(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return (a & 0xFFu) | ((b & 0xFFu) << 8) | ((c & 0xFFu) << 16) | (d << 24);
}
#else
template <int K, class T> DEV T quad_bcast(T v) { return dpp_any<T, K * 0x55>(v); }
DEV float wave_read_lane(float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
DEV u32x2 ld_stream8(const void* p) { return __builtin_nontemporal_load((const u32x2*)p); }
DEV u32x2 ld8(const void* p) { return *(const u32x2*)p; }
DEV uint32_t ld_stream4(const void* p) { return __builtin_nontemporal_load((const uint32_t*)p); }
DEV int bfe_i32(uint32_t v, int off, int width) { return __builtin_amdgcn_sbfe((int)v, off, width); }
// the low bytes of four words as one word (three v_perm_b32)
DEV uint32_t pack_low_bytes(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const uint32_t ab = __builtin_amdgcn_perm(b, a, 0x0C0C0400u);   // byte 0 of a, byte 0 of b, 0, 0
    const uint32_t cd = __builtin_amdgcn_perm(d, c, 0x04000C0Cu);   // 0, 0, byte 0 of c, byte 0 of d
    return ab | cd;
}
#endif

// The reference's four chain steps of one record, acc = fma(d_c, s_c, acc) for c = 0..3 in order, operands from the quad's lanes
// c = 0..3 (every lane of the quad runs the same chain): v_mov_b32_dpp broadcasts d_c, v_fmac_f32_dpp takes s_c through its own
// quad_perm — 8 instructions instead of the 10 of separate broadcasts (the compiler SLP-packs those into v_pk_fma_f32 + 16 moves
// when two chains run side by side).  v_fmac_f32 is the fused multiply-add (one rounding), like fmaf.  The leading s_nop covers
// the VALU-write -> DPP-read hazard for d and s (hipcc does not look into the statement).
#ifdef CT_EMU
static inline float quad_chain4(float acc, float d, float s) {
    acc = fmaf(quad_bcast<0>(d), quad_bcast<0>(s), acc);
    acc = fmaf(quad_bcast<1>(d), quad_bcast<1>(s), acc);
    acc = fmaf(quad_bcast<2>(d), quad_bcast<2>(s), acc);
    acc = fmaf(quad_bcast<3>(d), quad_bcast<3>(s), acc);
    return acc;
}
#else
DEV float quad_chain4(float acc, float d, float s) {
    float t;
    asm("s_nop 1\n\t"
        "v_mov_b32_dpp %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %3, %1 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %3, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %3, %1 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %3, %1 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
        : "+v"(acc), "=&v"(t) : "v"(d), "v"(s));
    return acc;
}
#endif

template <class T> DEV T wave_sum_fast(T v) {
    v += lane_xor1(v); v += lane_xor2(v); v += lane_xor4(v); v += lane_xor8(v); v += lane_xor16(v); v += lane_xor32(v);
    return v;
}

// ---- wave-private LDS exchange: all lanes of a wave wrote, all lanes of the same wave read ---------------------------------
// DS operations of one wave are executed in issue order; what is needed is that the compiler keeps the order and that the
// written data has left the store queue: a workgroup-scope release/acquire pair on the LDS address space (lowers to
// `s_waitcnt lgkmcnt(0)`, no vmcnt — weight loads stay in flight) around a scheduling barrier.
// One dword through the scalar data cache (lgkmcnt), load and wait in one statement: a device scalar such as the cursor position,
// which hipcc otherwise fetches with a vector load followed by `s_waitcnt vmcnt(0)` — draining every weight load in flight.
#ifdef CT_EMU
static inline int sload_i32(const int* p) { return *p; }
static inline void sload_i32x4(const int* p, int (&v)[4]) { v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3]; }
static inline int sload_i32x4_and(const int* p, int (&v)[4], const int* q) { v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3]; return *q; }
#else
// four consecutive dwords, one scalar load
DEV void sload_i32x4(const int* p, int (&v)[4]) {
    typedef int i32x4s __attribute__((ext_vector_type(4)));
    i32x4s r;
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
    v[0] = r[0]; v[1] = r[1]; v[2] = r[2]; v[3] = r[3];
}
// four consecutive dwords and one more from another line: both loads in flight together, one wait
DEV int sload_i32x4_and(const int* p, int (&v)[4], const int* q) {
    typedef int i32x4s __attribute__((ext_vector_type(4)));
    i32x4s r;
    int w;
    asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r), "=&s"(w) : "s"(p), "s"(q) : "memory");
    v[0] = r[0]; v[1] = r[1]; v[2] = r[2]; v[3] = r[3];
    return w;
}
DEV int sload_i32(const int* p) {
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
#endif

// kernarg_touch<BYTES>: one batch of scalar loads, a dword of every 64-byte line of the kernel-argument segment, at the top of a kernel.  hipcc fetches
// the fields of a by-value argument struct where they are first used, and each cold line is a round trip to L2: the fused QKV + attention launch read its
// 620 bytes in seven places one after the other and reached its first weight request 2 200 cycles after the plain mat-vec launch did.  After the touch
// every later read hits the scalar cache.
#ifdef CT_EMU
template <int BYTES> static inline void kernarg_touch() {}
#else
template <int BYTES> DEV void kernarg_touch() {
    typedef __attribute__((address_space(4))) const uint32_t* kptr;
    kptr kp = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int N = (BYTES + 63) / 64;
    uint32_t t[N + 1];
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = kp[16 * i];
    t[N] = kp[(BYTES - 4) / 4];
#pragma unroll
    for (int i = 0; i <= N; ++i) asm volatile("" ::"s"(t[i]));
}
#endif

// opaque_int: the value, but the compiler cannot see where it came from — what is derived from it is recomputed, not kept in registers.
#ifdef CT_EMU
static inline int opaque_int(int v) { return v; }
#else
DEV int opaque_int(int v) { asm volatile("" : "+v"(v)); return v; }
#endif

// ---- LDS arrival counters: waves of one workgroup meet without a workgroup barrier (kernels_v9.h prologue) ------------------------
#ifdef CT_EMU
DEV void lds_signal(unsigned* ctr, int lane, unsigned inc) {
    emu::wave_sync();   // a wave's LDS writes (several lanes may have written) happen before its count moves: lockstep on the hardware
    if (lane == 0) *ctr += inc;
}
DEV void lds_wait_ge(const unsigned* ctr, unsigned target) {
    while (*(const volatile unsigned*)ctr < target) emu::spin_yield();
}
#else
DEV void lds_signal(unsigned* ctr, int lane, unsigned inc) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's LDS writes are visible before the count moves
    if (lane == 0) __hip_atomic_fetch_add(ctr, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
DEV void lds_wait_ge(const unsigned* ctr, unsigned target) {
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
#endif

// sched_fence: nothing is moved across this point by the instruction scheduler.  sgpr_const: a constant the compiler keeps in a scalar
// register instead of re-materialising it as a literal operand.
#ifdef CT_EMU
static inline void sched_fence() {}
static inline uint32_t sgpr_const(uint32_t v) { return v; }
#else
DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
DEV uint32_t sgpr_const(uint32_t v) { uint32_t r; asm volatile("s_mov_b32 %0, %1" : "=s"(r) : "i"(v)); return r; }
#endif

// Two-wide f32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, full rate on gfx950): each half is the IEEE operation.
#ifdef CT_EMU
struct F32x2 { float x, y; };
static inline F32x2 pk2(float x, float y) { return F32x2{x, y}; }
static inline F32x2 pk_fma_f32(F32x2 a, F32x2 b, F32x2 c) { return F32x2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
static inline F32x2 pk_mul_f32(F32x2 a, F32x2 b) { return F32x2{a.x * b.x, a.y * b.y}; }
static inline F32x2 pk_add_f32(F32x2 a, F32x2 b) { return F32x2{a.x + b.x, a.y + b.y}; }
static inline float pk_lo(F32x2 a) { return a.x; }
static inline float pk_hi(F32x2 a) { return a.y; }
#else
typedef float F32x2 __attribute__((ext_vector_type(2)));
DEV F32x2 pk2(float x, float y) { return F32x2{x, y}; }
DEV F32x2 pk_fma_f32(F32x2 a, F32x2 b, F32x2 c) { return __builtin_elementwise_fma(a, b, c); }
DEV F32x2 pk_mul_f32(F32x2 a, F32x2 b) { return a * b; }
DEV F32x2 pk_add_f32(F32x2 a, F32x2 b) { return a + b; }
DEV float pk_lo(F32x2 a) { return a[0]; }
DEV float pk_hi(F32x2 a) { return a[1]; }
#endif

// Four f32 chain steps a[j] = fma(d[j], s[j], a[j]) as two v_pk_fma_f32 (each half is the IEEE fma), pinned where they are written:
// the asm names the two register pairs, so the steps are neither split into four scalar fmas nor moved to the end of the block.
#ifdef CT_EMU
static inline void chain4(float (&a)[4], const float (&d)[4], const f32x4& s) {
    for (int j = 0; j < 4; ++j) a[j] = fmaf(d[j], s[j], a[j]);
}
static inline uint32_t vgpr_const(uint32_t v) { return v; }
#else
typedef float f32x2 __attribute__((ext_vector_type(2)));
DEV void chain4(float (&a)[4], const float (&d)[4], const f32x4& s) {
    f32x2 a0 = {a[0], a[1]}, a1 = {a[2], a[3]};
    const f32x2 d0 = {d[0], d[1]}, d1 = {d[2], d[3]}, s0 = {s[0], s[1]}, s1 = {s[2], s[3]};
    a0 = __builtin_elementwise_fma(d0, s0, a0);
    a1 = __builtin_elementwise_fma(d1, s1, a1);
    asm volatile("" : "+v"(a0), "+v"(a1));
    a[0] = a0[0]; a[1] = a0[1]; a[2] = a1[0]; a[3] = a1[1];
}
// a constant held in a vector register (gfx9 VOP3 takes one scalar / constant operand: v_and_or_b32 needs the other one here)
DEV uint32_t vgpr_const(uint32_t v) { uint32_t r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "i"(v)); return r; }
#endif

// ---- LDS-DMA: 16 bytes per lane from global memory straight into LDS at (wave-uniform base) + 16 * lane, no VGPRs --------------
// Issued as one asm statement (M0 = destination base is written in the statement that reads it).  hipcc does not count it:
// its own `s_waitcnt vmcnt(N)` stay correct (memory operations retire in order, an uncounted younger one can only make them wait
// longer) and the DATA needs vm_wait<N>() by hand, then a barrier, before any lane reads it.
#ifdef CT_EMU
static inline void glds16(const void* gsrc, unsigned char* lds_wave_base) { memcpy(lds_wave_base + 16 * (threadIdx.x & 63), gsrc, 16); }
static inline void glds4(const void* gsrc, unsigned char* lds_wave_base) { memcpy(lds_wave_base + 4 * (threadIdx.x & 63), gsrc, 4); }
static inline void glds16_s(const void* sbase, uint32_t voff, unsigned char* lds_wave_base) { memcpy(lds_wave_base + 16 * (threadIdx.x & 63), (const unsigned char*)sbase + voff, 16); }
static inline void glds4_s(const void* sbase, uint32_t voff, unsigned char* lds_wave_base) { memcpy(lds_wave_base + 4 * (threadIdx.x & 63), (const unsigned char*)sbase + voff, 4); }
template <int N> static inline void vm_wait() {}
template <int N> static inline void sleep_cycles() {}
#else
DEV void glds4(const void* gsrc, unsigned char* lds_wave_base) {   // 4 bytes per lane: 256 B per wave instruction
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)lds_wave_base);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
DEV void glds16(const void* gsrc, unsigned char* lds_wave_base) {
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)lds_wave_base);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// The same with the source as (wave-uniform base in scalar registers) + (32-bit lane offset): no 64-bit vector address arithmetic per piece.
DEV void glds16_s(const void* sbase, uint32_t voff, unsigned char* lds_wave_base) {
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)lds_wave_base);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(dst), "s"(sbase) : "memory");
}
DEV void glds4_s(const void* sbase, uint32_t voff, unsigned char* lds_wave_base) {
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)lds_wave_base);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(dst), "s"(sbase) : "memory");
}
template <int N> DEV void sleep_cycles() { __builtin_amdgcn_s_sleep(N); }   // N * 64 clocks
template <int N> DEV void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }
#endif

// Scheduling fence on values: they are computed before this point, nothing that depends on them moves above it.
#ifdef CT_EMU
static inline void reg_fence(float&, float&, float&, float&) {}
#else
DEV void reg_fence(float& a, float& b, float& c, float& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
#endif

#ifdef CT_EMU
static inline void wave_lds_sync() { emu::wave_sync(); }
#else
DEV void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#endif

#ifdef CT_EMU
static inline unsigned long long clock64_dev() { return 0; }
#else
DEV unsigned long long clock64_dev() { return __builtin_amdgcn_s_memtime(); }
#endif

// ---- device-scope ("agent": all XCDs of the chip) words that workgroups of one launch hand to each other ------------------------------
// The L2 of an XCD is not coherent with the other seven: a plain store stays in the writer's L2 until the kernel ends.  These are
// RELAXED device-scope atomics (performed at the memory side, no cache maintenance): an acquire / release fence at this scope is a
// write-back + invalidate of the XCD's whole L2 — measured in round 5 at ~13 us when the 256 workgroups of a head launch each issued one
// (profiles/r05_head_fold.txt).  Ordering comes from the RETURN VALUES instead: an atomic whose result has arrived has been performed.
#ifdef CT_EMU
static inline unsigned agent_add_u32(unsigned* p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }   // workgroups run one after the other
static inline unsigned long long agent_max_u64(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; if (v > o) *p = v; return o; }
static inline unsigned agent_ld_u32(const unsigned* p) { return *p; }
static inline void agent_st_u32(unsigned* p, unsigned v) { *p = v; }
static inline void agent_st_u64(unsigned long long* p, unsigned long long v) { *p = v; }
#else
DEV unsigned agent_add_u32(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV unsigned long long agent_max_u64(unsigned long long* p, unsigned long long v) { return __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV unsigned agent_ld_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void agent_st_u32(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void agent_st_u64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
