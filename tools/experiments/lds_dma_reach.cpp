// How far does the LDS-DMA destination base (M0) reach?  global_load_lds_dwordx4 into LDS byte offsets 0, 60 KB, 70 KB, 100 KB, 150 KB of a
// 160 KB dynamic allocation, read back with ds_read.  (kernels_pg.h keeps its stage buffers below 64 KB unless this says it need not.)
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_dma_reach tools/experiments/lds_dma_reach.cpp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
__global__ void k(const u4* src, u4* out, int off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 160 * 1024 / 16; i += 64) ((u4*)smem)[i] = u4{0xdeadbeefu, 0, 0, 0};
    __syncthreads();
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(smem + off));
    uint32_t keep;
    const u4* g = src + lane;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    __syncthreads();
    __builtin_amdgcn_s_sleep(20);
    __syncthreads();
    out[lane] = ((u4*)(smem + off))[lane];
}
int main() {
    u4 h[64], *d, *o, r[64];
    for (int i = 0; i < 64; ++i) h[i] = u4{(uint32_t)i, 0x1111u * i, 7u, 9u};
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof h);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int offs[] = {0, 60 * 1024, 70 * 1024, 100 * 1024, 150 * 1024};
    for (int off : offs) {
        hipMemset(o, 0, sizeof h);
        k<<<1, 64, 160 * 1024>>>(d, o, off);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 64; ++i) bad += r[i][0] != h[i][0] || r[i][1] != h[i][1] || r[i][2] != 7u;
        printf("LDS-DMA to byte offset %6d: %s (%d lanes wrong, first word read 0x%x, launch %s)\n", off, bad ? "WRONG" : "ok", bad, r[0][0], hipGetErrorString(e));
    }
    return 0;
}
