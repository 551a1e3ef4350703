// Fill-rate probe: how many bytes per clock can ONE CU pull out of L2 - into registers (buffer_load_dwordx4) and into LDS by DMA (buffer_load_dwordx4 ... lds)?
// Every workgroup streams its own window of a buffer small enough to stay in the L2s (2 MB per XCD), `rep` times; 1 or 2 workgroups per CU (LDS size decides),
// 4 or 8 waves.  Build: hipcc --offload-arch=gfx950 -O3 tools/probes/fill_rate.hip -o /tmp/fill_rate ; run: /tmp/fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int v4i;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE, int DEPTH>      // MODE 0: to registers, 1: LDS-DMA; DEPTH: 1 KiB pieces in flight per wave
__global__ void fill_kernel(const char* buf, long window, int rep, int lds_pad, unsigned* sink) {
    extern __shared__ __attribute__((aligned(1024))) char sm[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const char* base = buf + (long)(blockIdx.x % 64) * window;          // 64 windows shared by the grid: L2-resident
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)window, 0x00020000);
    const int pieces = (int)(window >> 10);
    v4i acc = {0, 0, 0, 0};
    for (int r = 0; r < rep; ++r) {
        for (int p0 = wave * DEPTH; p0 < pieces; p0 += nw * DEPTH) {
            if (MODE == 0) {
                v4i t[DEPTH];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) t[d] = (v4i)__builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (p0 + d) * 1024, 0);
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) acc ^= t[d];
            } else {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(sm + ((wave * DEPTH + d) & 31) * 1024), 16, lane * 16, (p0 + d) * 1024, 0, 0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
            }
        }
    }
    if (MODE == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc[0] = ((int*)sm)[lane + lds_pad * 0]; }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) sink[0] = 1;
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* buf, unsigned* sink, int threads, size_t lds, int wgs_per_cu) {
    const long window = 256 << 10;       // 256 KB per window, 64 windows = 16 MB (2 MB per XCD)
    const int rep = 64, grid = 256 * wgs_per_cu;
    hipFuncSetAttribute((const void*)fill_kernel<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((fill_kernel<MODE, DEPTH>), dim3(grid), dim3(threads), lds, 0, buf, window, 4, 0, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill_kernel<MODE, DEPTH>), dim3(grid), dim3(threads), lds, 0, buf, window, rep, 0, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * window * rep;
    printf("%-34s %d waves x %d WG/CU, %2d KiB in flight per wave: %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU (2.1 GHz)\n", name, threads / 64, wgs_per_cu, DEPTH, ms,
           bytes / ms / 1e9, bytes / (ms * 1e-3) / 256.0 / 2.1e9);
}

int main() {
    char* buf; unsigned* sink;
    hipMalloc(&buf, 64 << 20); hipMemset(buf, 1, 64 << 20); hipMalloc(&sink, 64);
    run<0, 4>("registers (buffer_load_dwordx4)", buf, sink, 256, 32768, 1);
    run<0, 8>("registers (buffer_load_dwordx4)", buf, sink, 256, 32768, 1);
    run<0, 8>("registers (buffer_load_dwordx4)", buf, sink, 512, 32768, 1);
    run<0, 8>("registers (buffer_load_dwordx4)", buf, sink, 256, 32768, 2);
    run<1, 4>("LDS-DMA (buffer_load ... lds)", buf, sink, 256, 32768, 1);
    run<1, 8>("LDS-DMA (buffer_load ... lds)", buf, sink, 256, 32768, 1);
    run<1, 8>("LDS-DMA (buffer_load ... lds)", buf, sink, 512, 65536, 1);
    run<1, 8>("LDS-DMA (buffer_load ... lds)", buf, sink, 256, 32768, 2);
    run<1, 16>("LDS-DMA (buffer_load ... lds)", buf, sink, 256, 32768, 2);
    return 0;
}
