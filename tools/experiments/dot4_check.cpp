// Does the VOP3P form `v_dot4_i32_i8 d, a, b, 0` (inline asm) give what __builtin_amdgcn_sdot4(a, b, 0) gives on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(const int* a, const int* b, int* o1, int* o2) {
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    int r;
    asm("v_dot4_i32_i8 %0, %1, %2, 0" : "=v"(r) : "v"(a[i]), "v"(b[i]));
    o1[i] = r;
    o2[i] = __builtin_amdgcn_sdot4(a[i], b[i], 0, false);
}
int main() {
    const int n = 4096;
    int *a, *b, *o1, *o2;
    hipMallocManaged(&a, n * 4); hipMallocManaged(&b, n * 4); hipMallocManaged(&o1, n * 4); hipMallocManaged(&o2, n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) { a[i] = rand() ^ (rand() << 16); b[i] = rand() ^ (rand() << 16); }
    a[0] = 0x0F0F0F0F; b[0] = (int)0x80FF7F01; a[1] = (int)0xE0E0E0E0; b[1] = 0x7F7F7F7F;
    k<<<n / 256, 256>>>(a, b, o1, o2);
    hipDeviceSynchronize();
    int bad = 0;
    for (int i = 0; i < n; ++i) if (o1[i] != o2[i]) { if (bad < 5) printf("i=%d a=%08x b=%08x asm=%d builtin=%d\n", i, a[i], b[i], o1[i], o2[i]); ++bad; }
    printf("mismatches: %d of %d\n", bad, n);
    return 0;
}
