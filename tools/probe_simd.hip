// prints, for 512-thread workgroups, which SIMD each wave landed on (HW_REG_HW_ID)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hwid;
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 8 * 4);
    hipLaunchKernelGGL(k, dim3(64), dim3(512), 0, 0, d);
    unsigned h[64 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 6; ++b) { printf("block %d: ", b); for (int w = 0; w < 8; ++w) printf("w%d:simd%u(cu%u,wave%u) ", w, (h[b*8+w] >> 4) & 3, (h[b*8+w] >> 8) & 15, h[b*8+w] & 15); printf("\n"); }
    return 0;
}
