// In-place residual update x[i] += v[i] over N floats: (a) load + add + store, (b) returnless fp32 atomic add (the read-modify-write happens at the L2, the
// wave never waits for x).  Rows of 64 consecutive floats per wave instruction, as a GEMM epilogue would issue them.  Prints ms and the x-side GB/s (8 B per element).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(256) void rmw(float* __restrict__ x, const float* __restrict__ v, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] += v[i & 0xfffff];
}
__global__ __launch_bounds__(256) void atom(float* __restrict__ x, const float* __restrict__ v, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(x + i), v[i & 0xfffff]);
}
__global__ __launch_bounds__(256) void wr(float* __restrict__ x, const float* __restrict__ v, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] = v[i & 0xfffff];
}
int main() {
    const long n = 393216L * 1536;
    float *x, *v; hipMalloc(&x, n * 4); hipMalloc(&v, (1 << 20) * 4);
    hipMemset(x, 0, n * 4); hipMemset(v, 0, (1 << 20) * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"load+add+store", "atomic add (no return)", "store only"};
    for (int grid : {2048, 8192, 65536}) for (int k = 0; k < 3; ++k) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 3; ++it) {
                if (k == 0) hipLaunchKernelGGL(rmw, dim3(grid), dim3(256), 0, 0, x, v, n);
                else if (k == 1) hipLaunchKernelGGL(atom, dim3(grid), dim3(256), 0, 0, x, v, n);
                else hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, x, v, n);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3; if (ms < best) best = ms;
        }
        printf("grid %6d %-24s %.3f ms  %.0f GB/s (x side, 8 B/elem; store only: 4 B)\n", grid, names[k], best, (k == 2 ? 4.0 : 8.0) * n / best / 1e6);
    }
    return 0;
}
