// Measurement aid of bench.py (ABI 19), not part of the model path: the rate the matrix pipes SUSTAIN on this device for the caller's operand values.
//
// Why.  The 2.5 PFLOP/s bf16 figure is 256 CUs x 4 SIMDs x 1024 flop per clock at 2.4 GHz.  MI355X clocks to its power budget: a stream of
// v_mfma_f32_16x16x32_bf16 (the product GEMM's MFMA) on random operands is granted ~1.7-1.9 GHz, the same stream on zeros ~2.3-2.4 GHz
// (profiles/r03_gemm_power.txt, profiles/r05_pipe_overlap_ubench.txt, MI355X_MICROARCH.md "DVFS give-back").  bench.py therefore reports, beside
// roofline.frac (against the 2.4 GHz peak, as the contract asks), what a register-fed MFMA stream with no memory system behind it reaches on the same
// box in the same run on operands of the bench's kind - the ceiling a GEMM kernel can approach by scheduling alone.
//
// Kernel: one wave per SIMD-slot, 4 x 4 register tile of 16x16 blocks (the 8-wave GEMM's wave tile is 8 x 4): 4 A and 4 B fragments loaded once from the
// caller's buffer, then `iters` rounds of 2 k-steps x 16 MFMAs on 16 independent accumulators - back-to-back issue, nothing else in the loop.
#include "cvar_common.h"

typedef short pb_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float pb_f32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void probe_mfma_bf16_kernel(const pb_bf16x8_t* __restrict__ ops, int iters, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 4 + (threadIdx.x >> 6)) & 15;        // 16 different fragment sets across the waves of the chip
    pb_bf16x8_t a[2][4], b[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[k][i] = ops[((wave * 16 + k * 8 + i) * 64) + lane];
            b[k][i] = ops[((wave * 16 + k * 8 + 4 + i) * 64) + lane];
        }
    pb_f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = pb_f32x4_t{0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k][i], b[k][j], acc[i][j], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f && sink) sink[0] = s;                            // keeps the accumulators alive; practically never true
    if (sink && blockIdx.x == 0 && threadIdx.x == 0) sink[1] = (float)(t1 - t0);     // shader cycles of one wave's loop: cycles / wall time = the clock the launch ran at
}

// The same stream on v_mfma_f32_32x32x16_bf16 (ABI 20; the MFMA of the conv tiles and of the guide's 2 495 TFLOP/s microbenchmark): 2 x 2 register tile of 32x32 blocks, 4 independent
// accumulators of 16 registers, `iters` rounds of 2 k-steps x 4 MFMAs x 2 = the same flop count per round as the 16x16x32 kernel.  Together with the board clock
// (bench.py samples AMD SMI beside both) it tells issue rate from clock: a 16x16x32 stream cannot use more than ~80-94 % of the pipe a 32x32x16 stream fills
// (MI355X_MICROARCH.md instruction table: ~5 vs ~8 cycles per CU for half the flops).
typedef float pb_f32x16_t __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void probe_mfma_bf16_32_kernel(const pb_bf16x8_t* __restrict__ ops, int iters, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 4 + (threadIdx.x >> 6)) & 15;
    pb_bf16x8_t a[2][2], b[2][2];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[k][i] = ops[((wave * 16 + k * 8 + i) * 64) + lane];
            b[k][i] = ops[((wave * 16 + k * 8 + 4 + i) * 64) + lane];
        }
    pb_f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k][i], b[k][j], acc[i][j], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f && sink) sink[0] = s;
    if (sink && blockIdx.x == 0 && threadIdx.x == 0) { sink[1] = (float)(t1 - t0); sink[2] = (float)(r1 - r0); }     // s_memtime ticks / s_memrealtime ticks (100 MHz) of one wave's loop
}
extern "C" int cvar_probe_mfma_bf16_32x32(const void* operands, int64_t operand_bytes, int iters, float* sink /* 3 floats */, void* stream) {
    if (!operands || operand_bytes < 16 * 16 * 64 * 16 || iters <= 0) return CVAR_EINVAL;
    if ((uintptr_t)operands & 15) return CVAR_EINVAL;
    hipLaunchKernelGGL(probe_mfma_bf16_32_kernel, dim3(512), dim3(256), 0, as_stream(stream), (const pb_bf16x8_t*)operands, iters, sink);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// operands: >= 16 * 16 * 64 * 16 B = 256 KB of bf16 values (the caller chooses them: the bench's randn, zeros, ...).  Launches 256 CUs x 2 workgroups of 4 waves
// (two waves per SIMD, as the product GEMM runs).  flop of the launch = cvar_probe_mfma_flops(iters).  sink (optional, 2 floats): [1] = shader cycles of one wave's loop.
extern "C" int cvar_probe_mfma_bf16(const void* operands, int64_t operand_bytes, int iters, float* sink, void* stream) {
    if (!operands || operand_bytes < 16 * 16 * 64 * 16 || iters <= 0) return CVAR_EINVAL;
    if ((uintptr_t)operands & 15) return CVAR_EINVAL;
    hipLaunchKernelGGL(probe_mfma_bf16_kernel, dim3(512), dim3(256), 0, as_stream(stream), (const pb_bf16x8_t*)operands, iters, sink);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}
extern "C" double cvar_probe_mfma_flops(int iters) { return 512.0 * 4.0 * (double)iters * 32.0 * 16384.0; }
