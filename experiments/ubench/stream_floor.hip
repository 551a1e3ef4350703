// Microbenchmark (round 5): what does ONE dependent launch that streams a d24 weight matrix cost on this box, with nothing but the stream in it?
// The small-M GEMM of a B = 1 generation (cvar_gemm_skinny_kernel, M = 4: 9.0 us per qkv call of 14.2 MB in a 24-call graph, tools/skinny_bench.py) against
// the same grid (N / 16 workgroups x 8 waves, a wave's six 1 KiB weight loads all in flight at once) doing only: load - xor-reduce - one store per wave.
//   variant 0: weights only                      variant 1: + every wave first reads 12 KB that the PREVIOUS launch wrote (the activations' dependency)
//   variant 2: as 1 + six workgroup barriers + an LDS round trip per wave (the skeleton of the shipped kernel's K loop and cross-wave reduction)
// 24 launches on 24 different matrices (340 MB: from HBM, not from the Infinity Cache) in one hipGraph, like the tool above.  us per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int VAR, int AUX>
__global__ __launch_bounds__(512) void stream_kernel(const v4i* __restrict__ W, int K16 /* 16-byte chunks per weight row */, const v4i* __restrict__ act_in, v4i* __restrict__ act_out) {
    __shared__ v4i red[8][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16;
    v4i acc = {0, 0, 0, 0};
    if (VAR >= 1) {
        // 12 KB of the predecessor's output (M = 4 rows x 1536 bf16): 768 chunks of 16 B, every workgroup reads all of them
#pragma unroll
        for (int i = 0; i < 2; ++i) { const int c = tid + 512 * i; if (c < 768) { const v4i a = act_in[c]; acc ^= a; } }
    }
    // wave w takes k-steps w, w + 8, ... (32 elements = 4 chunks per row and step); lane = (row l15, chunk kq)
    const int l15 = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x7fffffff, 0x00020000);
    v4i w[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const int chunk = (8 * b + wave) * 4 + kq;
        w[b] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((n0 + l15) * K16 + chunk) * 16, 0, AUX);
    }
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        acc ^= w[b];
        if (VAR == 2) __syncthreads();
        if (VAR == 5) __builtin_amdgcn_s_barrier();          // bare s_barrier: no memory fence in front of it
    }
    if (VAR >= 2) {
        red[wave][lane] = acc;
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int j = 1; j < 8; ++j) acc ^= red[j][lane];
        }
    }
    if (VAR >= 2 ? wave == 0 : true) {
        // (variants without the reduction: EVERY wave stores, otherwise the compiler sinks the loads of waves 1-7 into dead code and the launch streams an eighth)
        if (lane < 16) act_out[(blockIdx.x * 16 + lane + (VAR >= 2 ? 0 : 97 * wave)) % 768] = acc;       // 12 KB of "output" for the successor
    }
}

// variant 3: the barrier-free form of the small-M GEMM - ONE wave per workgroup owns 16 output columns over the whole K: 48 weight fragments (1 KiB each) and the
// 48 activation fragments (rows >= M: out-of-range = zeros, no traffic) all in flight, 48 v_mfma_f32_16x16x32_bf16, one 16-byte store per lane.  No LDS, no barrier.
typedef __attribute__((ext_vector_type(8))) __bf16 bfv8;
typedef float f4 __attribute__((ext_vector_type(4)));
template <int AUX, int KS>
__global__ __launch_bounds__(64) void onewave_kernel(const v4i* __restrict__ W, int K16, const v4i* __restrict__ act_in, int M, float* __restrict__ out, int N) {
    const int lane = threadIdx.x, l15 = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void*)act_in, 0, M * K16 * 16, 0x00020000);
    v4i w[KS], a[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) w[s] = __builtin_amdgcn_raw_buffer_load_b128(wr, ((n0 + l15) * K16 + 4 * s + kq) * 16, 0, AUX);
#pragma unroll
    for (int s = 0; s < KS; ++s) a[s] = __builtin_amdgcn_raw_buffer_load_b128(ar, (l15 * K16 + 4 * s + kq) * 16, 0, 0);
    __builtin_amdgcn_sched_barrier(0);                 // every load is issued before the first MFMA (the compiler otherwise chains load - wait - MFMA with 32 registers)
    f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfv8, w[s]), __builtin_bit_cast(bfv8, a[s]), acc, 0, 0, 0);
    if (l15 < M) *(f4*)(out + (long)l15 * N + n0 + 4 * kq) = acc;
}

template <int AUX>
static double run1w(const char* name, v4i* Wall, size_t mat_chunks, int N, int M, v4i* a0, float* o, int depth) {
    hipStream_t st; hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < depth; ++i)
        hipLaunchKernelGGL((onewave_kernel<AUX, 48>), dim3(N / 16), dim3(64), 0, st, Wall + (size_t)i * mat_chunks, 192, (i & 1) ? (const v4i*)(o) : a0, M, (i & 1) ? (float*)a0 : o, N);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps / depth;
    printf("%-78s N=%5d M=%5d  %6.2f us per launch  %6.2f TB/s of weights\n", name, N, M, us, (double)N * 192 * 16 / us / 1e6);
    return us;
}

template <int VAR, int AUX>
static double run(const char* name, v4i* Wall, size_t mat_chunks, int N, int K16, v4i* a0, v4i* a1, int depth) {
    hipStream_t st; hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < depth; ++i)
        hipLaunchKernelGGL((stream_kernel<VAR, AUX>), dim3(N / 16), dim3(512), 0, st, Wall + (size_t)i * mat_chunks, K16, (i & 1) ? a1 : a0, (i & 1) ? a0 : a1);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps / depth;
    printf("%-78s N=%5d K=%5d  %6.2f us per launch  %6.2f TB/s of weights\n", name, N, K16 * 8, us, (double)N * K16 * 16 / us / 1e6);
    return us;
}

int main() {
    const int depth = 24;
    const size_t mat_chunks = (size_t)6144 * 192;                  // the largest matrix of a block: 6144 x 1536 bf16 = 18.9 MB
    v4i* W; hipMalloc(&W, mat_chunks * 16 * depth);
    hipMemset(W, 1, mat_chunks * 16 * depth);
    v4i *a0, *a1; hipMalloc(&a0, 768 * 16); hipMalloc(&a1, 768 * 16); hipMemset(a0, 0, 768 * 16); hipMemset(a1, 0, 768 * 16);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>("0: weight stream only, default cache policy", W, mat_chunks, 4608, 192, a0, a1, depth);
        run<0, 2>("0: weight stream only, nt loads", W, mat_chunks, 4608, 192, a0, a1, depth);
        run<1, 0>("1: + 12 KB of the predecessor's output read by every workgroup", W, mat_chunks, 4608, 192, a0, a1, depth);
        run<1, 2>("1: the same, nt weight loads", W, mat_chunks, 4608, 192, a0, a1, depth);
        run<2, 0>("2: + six workgroup barriers and a cross-wave LDS reduction", W, mat_chunks, 4608, 192, a0, a1, depth);
        run<4, 0>("4: as 1 + ONLY the cross-wave LDS reduction (two workgroup barriers), no barrier in the loop", W, mat_chunks, 4608, 192, a0, a1, depth);
        run<5, 0>("5: as 2 with bare s_barrier in the loop (no vmcnt(0) fence)", W, mat_chunks, 4608, 192, a0, a1, depth);
        run<2, 0>("2: the same on the fc1 matrix (6144 x 1536)", W, mat_chunks, 6144, 192, a0, a1, depth);
        run<2, 0>("2: the same on the proj matrix (1536 x 1536)", W, mat_chunks, 1536, 192, a0, a1, depth);
    }
    {
        float* o; hipMalloc(&o, 16 * 6144 * 4 * 2); hipMemset(o, 0, 16 * 6144 * 4 * 2);
        v4i* ab; hipMalloc(&ab, 16 * 6144 * 4 * 2); hipMemset(ab, 0, 16 * 6144 * 4 * 2);
        for (int rep = 0; rep < 2; ++rep) {
            run1w<0>("3: one wave per 16 columns, whole K, no LDS / barrier (qkv), M = 4", W, mat_chunks, 4608, 4, ab, o, depth);
            run1w<2>("3: the same, nt weight loads", W, mat_chunks, 4608, 4, ab, o, depth);
            run1w<0>("3: the same, M = 16", W, mat_chunks, 4608, 16, ab, o, depth);
            run1w<0>("3: fc1 matrix, M = 16", W, mat_chunks, 6144, 16, ab, o, depth);
            run1w<0>("3: proj matrix, M = 16 (96 waves)", W, mat_chunks, 1536, 16, ab, o, depth);
        }
    }
    return 0;
}
