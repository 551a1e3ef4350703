// Training-side kernels of the teacher-forced transformer step (SURVEY.md 8a rows A5 backward / A20):
// gated residual + its gradient, GELU forward/backward, adaLN-modulated LayerNorm backward, column sums, fused
// cross-entropy forward+backward, embedding scatter, SiLU backward, fused AdamW and the gradient-norm reduction.
// All reductions are two-level with a fixed order (no atomics): results are bit-reproducible.
#include "cvar_common.h"

constexpr int RED_S = 8;      // row segments of the per-sequence column reductions (scalar fallbacks)
// The vector kernels split every sequence into red_segments(R, l) row segments so that R * segments blocks fill the chip
// (8 segments x 32 sequences left 3/4 of the wave slots empty and the kernels ran at 2.2-2.8 TB/s).
#ifndef CVAR_RED_TARGET
#define CVAR_RED_TARGET 512       // ln_modulate_bwd: blocks of 4 waves at 2 waves per SIMD -> 512 blocks are exactly one resident round
#endif
#ifndef CVAR_GG_TARGET
#define CVAR_GG_TARGET 256        // gated_grad: x ceil(C/512) blocks of 2 waves
#endif
constexpr int RED_S_MAX = 64;
static inline int red_segments(int R, int l, int target) { return max(1, min(min(RED_S_MAX, (target + R - 1) / max(R, 1)), (l + 3) / 4)); }
static inline int red_segments_max(int R) { return max(RED_S, red_segments(R, 1 << 20, max(CVAR_RED_TARGET, CVAR_GG_TARGET))); }
// floats of workspace the per-sequence reductions (cvar_gated_grad, cvar_ln_modulate_bwd) may use for M = R * l rows
extern "C" int64_t cvar_train_ws_floats(int64_t M, int R, int C) {
    if (M <= 0 || R <= 0 || C <= 0) return 0;
    return 2 * M + 2 * (int64_t)red_segments_max(R) * R * C;
}

// x[m,c] += gate[r,c] * rowscale[r] * f[m,c]          (x + drop_path(gamma * f(x)), basic_var.py:208-209)
template <typename T>
__global__ void gate_residual_kernel(float* __restrict__ x, const T* __restrict__ f, const float* __restrict__ gate, long ldg,
                                     int gate_rows, const float* __restrict__ rowscale, long M, int C) {
    const long nvec = M * (C / 4);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const long m = i / (C / 4);
        const int c = (int)(i % (C / 4)) * 4;
        const long r = m / gate_rows;
        const float rs = rowscale ? rowscale[r] : 1.0f;
        const f32x4_t g = *(const f32x4_t*)(gate + r * ldg + c);
        f32x4_t xv = *(f32x4_t*)(x + m * C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[e] += (g[e] * rs) * Elem<T>::ld(f + m * C + c + e);
        *(f32x4_t*)(x + m * C + c) = xv;
    }
}

extern "C" int cvar_gate_residual(float* x, const void* f, int dtype, const float* gate, int64_t ldg, int gate_rows,
                                  const float* rowscale, int64_t M, int C, void* stream) {
    if (!x || !f || !gate || M <= 0 || C <= 0 || gate_rows <= 0) return CVAR_EINVAL;
    if (C % 4 || ldg % 4) return CVAR_EUNSUPPORTED;
    dim3 grid((unsigned)min((int64_t)8192, (M * (C / 4) + 255) / 256)), block(256);
    if (dtype == CVAR_BF16) hipLaunchKernelGGL(gate_residual_kernel<bf16_t>, grid, block, 0, as_stream(stream), x, (const bf16_t*)f, gate, (long)ldg, gate_rows, rowscale, (long)M, C);
    else if (dtype == CVAR_F32) hipLaunchKernelGGL(gate_residual_kernel<float>, grid, block, 0, as_stream(stream), x, (const float*)f, gate, (long)ldg, gate_rows, rowscale, (long)M, C);
    else return CVAR_EUNSUPPORTED;
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// ---- GELU(tanh) forward / backward ---------------------------------------------------------------------------
template <typename T, bool BWD>
__global__ void gelu_kernel(const T* __restrict__ a, T* __restrict__ io, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float t = Elem<T>::ld(a + i);
        if (BWD) Elem<T>::st(io + i, Elem<T>::ld(io + i) * gelu_tanh_grad(t));
        else Elem<T>::st(io + i, gelu_tanh_f(t));
    }
}
// h = gelu(a)
extern "C" int cvar_gelu(const void* a, void* h, int dtype, int64_t n, void* stream) {
    if (!a || !h || n <= 0) return CVAR_EINVAL;
    dim3 grid((unsigned)min((int64_t)8192, (n + 255) / 256)), block(256);
    if (dtype == CVAR_BF16) hipLaunchKernelGGL((gelu_kernel<bf16_t, false>), grid, block, 0, as_stream(stream), (const bf16_t*)a, (bf16_t*)h, (long)n);
    else if (dtype == CVAR_F32) hipLaunchKernelGGL((gelu_kernel<float, false>), grid, block, 0, as_stream(stream), (const float*)a, (float*)h, (long)n);
    else return CVAR_EUNSUPPORTED;
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}
// dh <- dh * gelu'(a)
extern "C" int cvar_gelu_bwd(const void* a, void* dh, int dtype, int64_t n, void* stream) {
    if (!a || !dh || n <= 0) return CVAR_EINVAL;
    dim3 grid((unsigned)min((int64_t)8192, (n + 255) / 256)), block(256);
    if (dtype == CVAR_BF16) hipLaunchKernelGGL((gelu_kernel<bf16_t, true>), grid, block, 0, as_stream(stream), (const bf16_t*)a, (bf16_t*)dh, (long)n);
    else if (dtype == CVAR_F32) hipLaunchKernelGGL((gelu_kernel<float, true>), grid, block, 0, as_stream(stream), (const float*)a, (float*)dh, (long)n);
    else return CVAR_EUNSUPPORTED;
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// ---- per-sequence column reductions ----------------------------------------------------------------------------
// grid (ceil(C/256), R, RED_S): thread = channel, block = one row segment of one sequence; partial[s][r][k][c]
template <typename T>
__global__ __launch_bounds__(256) void gated_grad_kernel(const float* __restrict__ dx, const T* __restrict__ f, const float* __restrict__ gate,
                                                        long ldg, const float* __restrict__ rowscale, T* __restrict__ df,
                                                        float* __restrict__ partial, int R, int l, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y, s = blockIdx.z;
    if (c >= C) return;
    const int seg = (l + RED_S - 1) / RED_S;
    const int t0 = s * seg, t1 = min(l, t0 + seg);
    const float g = gate[(long)r * ldg + c] * (rowscale ? rowscale[r] : 1.0f);
    float acc = 0.f;
    for (int t = t0; t < t1; ++t) {
        const long i = ((long)r * l + t) * C + c;
        const float d = dx[i];
        acc += d * Elem<T>::ld(f + i);
        Elem<T>::st(df + i, d * g);
    }
    partial[((long)s * R + r) * C + c] = acc;
}
// four channels per thread (16-byte dx loads, 8-byte f / df accesses); per-channel summation order over the tokens is unchanged -> bit-identical
__global__ __launch_bounds__(128) void gated_grad_vec_kernel(const float* __restrict__ dx, const bf16_t* __restrict__ f, const float* __restrict__ gate,
                                                            long ldg, const float* __restrict__ rowscale, bf16_t* __restrict__ df,
                                                            float* __restrict__ partial, int R, int l, int C, int nseg) {
    const int c = (blockIdx.x * 128 + threadIdx.x) * 4;
    const int r = blockIdx.y, s = blockIdx.z;
    if (c >= C) return;
    const int seg = (l + nseg - 1) / nseg;
    const int t0 = s * seg, t1 = min(l, t0 + seg);
    const float rs = rowscale ? rowscale[r] : 1.0f;
    const f32x4_t g4 = *(const f32x4_t*)(gate + (long)r * ldg + c);
    float g[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = g4[e] * rs;
#pragma unroll 4
    for (int t = t0; t < t1; ++t) {
        const long i = ((long)r * l + t) * C + c;
        const f32x4_t d = *(const f32x4_t*)(dx + i);
        const bf16x4_t fq = *(const bf16x4_t*)(f + i);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[e] += d[e] * bf16_to_f32((bf16_t)fq[e]); o[e] = d[e] * g[e]; }
        *(bf16x4_t*)(df + i) = pack_bf16x4(o);
    }
    const f32x4_t a4 = {acc[0], acc[1], acc[2], acc[3]};
    *(f32x4_t*)(partial + ((long)s * R + r) * C + c) = a4;
}
// out[r*ldo + c] = scale_r * sum_s partial[s][r][c]
__global__ void red_finalize_kernel(const float* __restrict__ partial, float* __restrict__ out, long ldo, const float* __restrict__ rowscale,
                                    int R, int C, int nseg) {
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= C) return;
    float a = 0.f;
    for (int s = 0; s < nseg; ++s) a += partial[((long)s * R + r) * C + c];
    out[(long)r * ldo + c] = a * (rowscale ? rowscale[r] : 1.0f);
}

// df = dx * gate * rowscale (dtype);  dgate[r, c] = rowscale[r] * sum_{m in r} dx[m,c] * f[m,c]
// ws: cvar_train_ws_floats(R * l, R, C) floats; ws_floats = what the caller allocated (checked)
extern "C" int cvar_gated_grad(const float* dx, const void* f, int dtype, const float* gate, int64_t ldg, const float* rowscale,
                               void* df, float* dgate, int64_t ldo, int R, int l, int C, float* ws, int64_t ws_floats, void* stream) {
    if (!dx || !f || !gate || !df || !dgate || !ws || R <= 0 || l <= 0 || C <= 0) return CVAR_EINVAL;
    if (ws_floats < cvar_train_ws_floats((int64_t)R * l, R, C)) return CVAR_EINVAL;       // ABI 16: the workspace size is part of the call
    dim3 grid(cdiv(C, 256), R, RED_S), block(256);
    const bool vec = dtype == CVAR_BF16 && C % 4 == 0 && ldg % 4 == 0 && ((((uintptr_t)dx | (uintptr_t)gate | (uintptr_t)ws) & 15) == 0) &&
                     ((((uintptr_t)f | (uintptr_t)df) & 7) == 0);
    const int nseg = vec ? red_segments(R, l, CVAR_GG_TARGET) : RED_S;
    if (vec) hipLaunchKernelGGL(gated_grad_vec_kernel, dim3(cdiv(C, 512), R, nseg), dim3(128), 0, as_stream(stream), dx, (const bf16_t*)f, gate, (long)ldg, rowscale,
                                (bf16_t*)df, ws, R, l, C, nseg);
    else if (dtype == CVAR_BF16) hipLaunchKernelGGL(gated_grad_kernel<bf16_t>, grid, block, 0, as_stream(stream), dx, (const bf16_t*)f, gate, (long)ldg, rowscale, (bf16_t*)df, ws, R, l, C);
    else if (dtype == CVAR_F32) hipLaunchKernelGGL(gated_grad_kernel<float>, grid, block, 0, as_stream(stream), dx, (const float*)f, gate, (long)ldg, rowscale, (float*)df, ws, R, l, C);
    else return CVAR_EUNSUPPORTED;
    hipLaunchKernelGGL(red_finalize_kernel, dim3(cdiv(C, 256), R), block, 0, as_stream(stream), ws, dgate, (long)ldo, rowscale, R, C, nseg);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// ---- adaLN LayerNorm backward ----------------------------------------------------------------------------------
// y = xhat * (1 + s) + b, xhat = (x - mu) * rstd.   Row kernel: dx_out = dx_in + rstd * (g - mean(g) - xhat * mean(g*xhat)),
// g = dy * (1 + s); also stores (mu, rstd) per row for the column kernel.
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_row_kernel(const float* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ scale,
                                                        long ld_ada, int rows_per, const float* __restrict__ dx_in, float* __restrict__ dx_out,
                                                        float* __restrict__ stats, int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (long)row * C;
    const T* dyr = dy + (long)row * C;
    const float* sc = scale + (long)(row / rows_per) * ld_ada;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mu; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    float sg = 0.f, sgx = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float g = Elem<T>::ld(dyr + c) * (1.0f + sc[c]);
        const float xh = (xr[c] - mu) * rstd;
        sg += g; sgx += g * xh;
    }
    const float mg = wave_sum(sg) / (float)C, mgx = wave_sum(sgx) / (float)C;
    for (int c = lane; c < C; c += 64) {
        const float g = Elem<T>::ld(dyr + c) * (1.0f + sc[c]);
        const float xh = (xr[c] - mu) * rstd;
        const float d = rstd * (g - mg - xh * mgx);
        dx_out[(long)row * C + c] = (dx_in ? dx_in[(long)row * C + c] : 0.f) + d;
    }
    if (lane == 0) { stats[2 * (long)row] = mu; stats[2 * (long)row + 1] = rstd; }
}
// The same row kernel with the row held in registers (x, dy, 1 + s as NV float4 vectors per lane, C <= NV * 256): one pass over memory
// instead of four dword-strided ones (the scalar form re-read the 6 KB row through L2 four times and ran at 2.4 TB/s effective).
template <typename T, int NV, bool TAIL>
__global__ __launch_bounds__(256) void ln_bwd_row_vec_kernel(const float* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ scale,
                                                            long ld_ada, int rows_per, const float* __restrict__ dx_in, float* __restrict__ dx_out,
                                                            float* __restrict__ stats, int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (long)row * C;
    const T* dyr = dy + (long)row * C;
    const float* sc = scale + (long)(row / rows_per) * ld_ada;
    f32x4_t xv[NV], gv[NV], din[NV];
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        const bool ok = !(TAIL && i == NV - 1) || c < C;
        xv[i] = ok ? *(const f32x4_t*)(xr + c) : zero4;
        const f32x4_t s4 = ok ? *(const f32x4_t*)(sc + c) : zero4;
        f32x4_t d4 = zero4;
        if (ok) {
            if constexpr (sizeof(T) == 2) {
                const bf16x4_t dq = *(const bf16x4_t*)(dyr + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) d4[e] = bf16_to_f32((bf16_t)dq[e]);
            } else {
                d4 = *(const f32x4_t*)(dyr + c);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[i][e] = d4[e] * (1.0f + s4[e]);
        din[i] = (ok && dx_in) ? *(const f32x4_t*)(dx_in + (long)row * C + c) : zero4;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        const bool ok = !(TAIL && i == NV - 1) || c < C;
#pragma unroll
        for (int e = 0; e < 4; ++e) { xv[i][e] = ok ? xv[i][e] - mu : 0.f; q += xv[i][e] * xv[i][e]; }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { xv[i][e] *= rstd; sg += gv[i][e]; sgx += gv[i][e] * xv[i][e]; }
    const float mg = wave_sum(sg) / (float)C, mgx = wave_sum(sgx) / (float)C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (TAIL && i == NV - 1 && c >= C) continue;
        f32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = din[i][e] + rstd * (gv[i][e] - mg - xv[i][e] * mgx);
        *(f32x4_t*)(dx_out + (long)row * C + c) = o;
    }
    if (lane == 0) { stats[2 * (long)row] = mu; stats[2 * (long)row + 1] = rstd; }
}
template <typename T>
static bool ln_bwd_row_vec_launch(const float* x, const T* dy, const float* scale, long ld_ada, int rows_per, const float* dx_in, float* dx_out, float* stats,
                                  int M, int C, float eps, hipStream_t st) {
    if (C % 4 || C > 2048 || ld_ada % 4 || (((uintptr_t)x | (uintptr_t)scale | (uintptr_t)dx_out | (uintptr_t)dx_in) & 15) || ((uintptr_t)dy & 7)) return false;
    const int nv = (C + 255) / 256;
    const bool tail = (C % 256) != 0;
    const dim3 grid(cdiv(M, 4)), block(256);
#define CVAR_LNB(NVV)                                                                                                                          \
    case NVV:                                                                                                                                  \
        if (tail) hipLaunchKernelGGL((ln_bwd_row_vec_kernel<T, NVV, true>), grid, block, 0, st, x, dy, scale, ld_ada, rows_per, dx_in, dx_out, stats, M, C, eps); \
        else hipLaunchKernelGGL((ln_bwd_row_vec_kernel<T, NVV, false>), grid, block, 0, st, x, dy, scale, ld_ada, rows_per, dx_in, dx_out, stats, M, C, eps);     \
        return true;
    switch (nv) { CVAR_LNB(1) CVAR_LNB(2) CVAR_LNB(3) CVAR_LNB(4) CVAR_LNB(5) CVAR_LNB(6) CVAR_LNB(7) CVAR_LNB(8) default: return false; }
#undef CVAR_LNB
}
// column kernel: partial[s][r][0][c] = sum dy * xhat, partial[s][r][1][c] = sum dy
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_col_kernel(const float* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ stats,
                                                        float* __restrict__ partial, int R, int l, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y, s = blockIdx.z;
    if (c >= C) return;
    const int seg = (l + RED_S - 1) / RED_S;
    const int t0 = s * seg, t1 = min(l, t0 + seg);
    float a = 0.f, b = 0.f;
    for (int t = t0; t < t1; ++t) {
        const long m = (long)r * l + t;
        const float d = Elem<T>::ld(dy + m * C + c);
        a += d * ((x[m * C + c] - stats[2 * m]) * stats[2 * m + 1]);
        b += d;
    }
    partial[(((long)s * R + r) * 2 + 0) * C + c] = a;
    partial[(((long)s * R + r) * 2 + 1) * C + c] = b;
}
// four channels per thread; same per-channel order over the tokens -> bit-identical to ln_bwd_col_kernel
__global__ __launch_bounds__(128) void ln_bwd_col_vec_kernel(const float* __restrict__ x, const bf16_t* __restrict__ dy, const float* __restrict__ stats,
                                                            float* __restrict__ partial, int R, int l, int C, int nseg) {
    const int c = (blockIdx.x * 128 + threadIdx.x) * 4;
    const int r = blockIdx.y, s = blockIdx.z;
    if (c >= C) return;
    const int seg = (l + nseg - 1) / nseg;
    const int t0 = s * seg, t1 = min(l, t0 + seg);
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int t = t0; t < t1; ++t) {
        const long m = (long)r * l + t;
        const f32x4_t xv = *(const f32x4_t*)(x + m * C + c);
        const bf16x4_t dq = *(const bf16x4_t*)(dy + m * C + c);
        const float mu = stats[2 * m], rstd = stats[2 * m + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = bf16_to_f32((bf16_t)dq[e]);
            a[e] += d * ((xv[e] - mu) * rstd);
            b[e] += d;
        }
    }
    const f32x4_t a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    *(f32x4_t*)(partial + (((long)s * R + r) * 2 + 0) * C + c) = a4;
    *(f32x4_t*)(partial + (((long)s * R + r) * 2 + 1) * C + c) = b4;
}

#ifndef CVAR_LNF_LEAN
#define CVAR_LNF_LEAN 0
#endif
// Row and column parts in ONE pass (bf16 dy): a block owns a row segment of one sequence, its four waves stride over the rows
// with the row in registers as above and keep the column sums (dy * xhat, dy) of their rows in registers; the waves are
// folded through LDS in a fixed order and the block writes partial[s][r][{0,1}][c] for ln_bwd_finalize_kernel.
// x, dy and dx_in are read once (the two-kernel form read x and dy twice: 670 MB instead of 469 MB per call at 21760 x 1536).
template <int NV, bool TAIL>
__global__ __launch_bounds__(256) void ln_bwd_fused_kernel(const float* __restrict__ x, const bf16_t* __restrict__ dy, const float* __restrict__ scale,
                                                          long ld_ada, int rows_per, const float* __restrict__ dx_in, float* __restrict__ dx_out,
                                                          float* __restrict__ partial, int R, int C, float eps, int seg_rows) {
    __shared__ f32x4_t red[3][2][NV * 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int s = blockIdx.x, r = blockIdx.y;
    const int t0 = s * seg_rows, t1 = min(rows_per, t0 + seg_rows);
    const float* sc = scale + (long)r * ld_ada;
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4_t s1[NV], ca[NV], cb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        const bool ok = !(TAIL && i == NV - 1) || c < C;
        const f32x4_t s4 = ok ? *(const f32x4_t*)(sc + c) : zero4;
#pragma unroll
        for (int e = 0; e < 4; ++e) s1[i][e] = 1.0f + s4[e];
        ca[i] = zero4; cb[i] = zero4;
    }
    for (int t = t0 + w; t < t1; t += 4) {
        const long row = (long)r * rows_per + t;
        const float* xr = x + row * C;
        const bf16_t* dyr = dy + row * C;
        f32x4_t xv[NV], dv[NV];
#if !CVAR_LNF_LEAN
        f32x4_t din[NV];
#endif
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            const bool ok = !(TAIL && i == NV - 1) || c < C;
            xv[i] = ok ? *(const f32x4_t*)(xr + c) : zero4;
            bf16x4_t dq = {0, 0, 0, 0};
            if (ok) dq = *(const bf16x4_t*)(dyr + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) dv[i][e] = bf16_to_f32((bf16_t)dq[e]);
#if !CVAR_LNF_LEAN
            din[i] = (ok && dx_in) ? *(const f32x4_t*)(dx_in + row * C + c) : zero4;
#endif
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) sum += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
        const float mu = wave_sum(sum) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            const bool ok = !(TAIL && i == NV - 1) || c < C;
#pragma unroll
            for (int e = 0; e < 4; ++e) { xv[i][e] = ok ? xv[i][e] - mu : 0.f; q += xv[i][e] * xv[i][e]; }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[i][e] *= rstd;
                const float g = dv[i][e] * s1[i][e];
                sg += g; sgx += g * xv[i][e];
                ca[i][e] += dv[i][e] * xv[i][e];
                cb[i][e] += dv[i][e];
            }
        const float mg = wave_sum(sg) / (float)C, mgx = wave_sum(sgx) / (float)C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (TAIL && i == NV - 1 && c >= C) continue;
            f32x4_t o;
#if CVAR_LNF_LEAN
            const f32x4_t dn = dx_in ? *(const f32x4_t*)(dx_in + row * C + c) : zero4;
#else
            const f32x4_t dn = din[i];
#endif
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = dn[e] + rstd * (dv[i][e] * s1[i][e] - mg - xv[i][e] * mgx);
            *(f32x4_t*)(dx_out + row * C + c) = o;
        }
    }
    if (w > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) { red[w - 1][0][i * 64 + lane] = ca[i]; red[w - 1][1][i * 64 + lane] = cb[i]; }
    }
    __syncthreads();
    if (w == 0) {
        float* pa = partial + (((long)s * R + r) * 2 + 0) * C;
        float* pb = partial + (((long)s * R + r) * 2 + 1) * C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (TAIL && i == NV - 1 && c >= C) continue;
            f32x4_t a = ca[i], b = cb[i];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const f32x4_t ra = red[k][0][i * 64 + lane], rb = red[k][1][i * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] += ra[e]; b[e] += rb[e]; }
            }
            *(f32x4_t*)(pa + c) = a;
            *(f32x4_t*)(pb + c) = b;
        }
    }
}
static bool ln_bwd_fused_launch(const float* x, const bf16_t* dy, const float* scale, long ld_ada, int rows_per, const float* dx_in, float* dx_out,
                                float* partial, int R, int C, float eps, int nseg, hipStream_t st) {
    if (C % 4 || C > 2048 || ld_ada % 4 || (((uintptr_t)x | (uintptr_t)scale | (uintptr_t)dx_out | (uintptr_t)dx_in | (uintptr_t)partial) & 15) ||
        ((uintptr_t)dy & 7)) return false;
    const int nv = (C + 255) / 256;
    const bool tail = (C % 256) != 0;
    const int seg_rows = (rows_per + nseg - 1) / nseg;
    const dim3 grid(nseg, R), block(256);
#define CVAR_LNF(NVV)                                                                                                                                    \
    case NVV:                                                                                                                                            \
        if (tail) hipLaunchKernelGGL((ln_bwd_fused_kernel<NVV, true>), grid, block, 0, st, x, dy, scale, ld_ada, rows_per, dx_in, dx_out, partial, R, C, eps, seg_rows); \
        else hipLaunchKernelGGL((ln_bwd_fused_kernel<NVV, false>), grid, block, 0, st, x, dy, scale, ld_ada, rows_per, dx_in, dx_out, partial, R, C, eps, seg_rows);     \
        return true;
    switch (nv) { CVAR_LNF(1) CVAR_LNF(2) CVAR_LNF(3) CVAR_LNF(4) CVAR_LNF(5) CVAR_LNF(6) CVAR_LNF(7) CVAR_LNF(8) default: return false; }
#undef CVAR_LNF
}
__global__ void ln_bwd_finalize_kernel(const float* __restrict__ partial, float* __restrict__ dscale, float* __restrict__ dshift, long ldo, int R, int C,
                                       int nseg) {
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= C) return;
    float a = 0.f, b = 0.f;
    for (int s = 0; s < nseg; ++s) { a += partial[(((long)s * R + r) * 2 + 0) * C + c]; b += partial[(((long)s * R + r) * 2 + 1) * C + c]; }
    dscale[(long)r * ldo + c] = a;
    dshift[(long)r * ldo + c] = b;
}

// ws: cvar_train_ws_floats(M, R, C) floats (row stats of the two-kernel form + the segment partials)
#ifndef CVAR_LN_BWD_FUSED
#define CVAR_LN_BWD_FUSED 1
#endif
extern "C" int cvar_ln_modulate_bwd(const float* x, const void* dy, int dtype, const float* scale, int64_t ld_ada, int rows_per,
                                    const float* dx_in, float* dx_out, float* dscale, float* dshift, int64_t ldo,
                                    int M, int C, float eps, float* ws, int64_t ws_floats, void* stream) {
    if (!x || !dy || !scale || !dx_out || !dscale || !dshift || !ws || M <= 0 || C <= 0 || rows_per <= 0 || M % rows_per) return CVAR_EINVAL;
    if (ws_floats < cvar_train_ws_floats(M, M / rows_per, C)) return CVAR_EINVAL;            // ABI 16
    const int R = M / rows_per;
    float* stats = ws;
    float* partial = ws + 2 * (size_t)M;
    dim3 b256(256);
    int nseg = RED_S;
    if (dtype == CVAR_BF16) {
        const int nv = red_segments(R, rows_per, CVAR_RED_TARGET);
        if (CVAR_LN_BWD_FUSED && ln_bwd_fused_launch(x, (const bf16_t*)dy, scale, (long)ld_ada, rows_per, dx_in, dx_out, partial, R, C, eps, nv, as_stream(stream))) {
            nseg = nv;
        } else {
            if (!ln_bwd_row_vec_launch<bf16_t>(x, (const bf16_t*)dy, scale, (long)ld_ada, rows_per, dx_in, dx_out, stats, M, C, eps, as_stream(stream)))
                hipLaunchKernelGGL(ln_bwd_row_kernel<bf16_t>, dim3(cdiv(M, 4)), b256, 0, as_stream(stream), x, (const bf16_t*)dy, scale, (long)ld_ada, rows_per, dx_in, dx_out, stats, M, C, eps);
            if (C % 4 == 0 && ((((uintptr_t)x | (uintptr_t)partial) & 15) == 0) && (((uintptr_t)dy & 7) == 0)) {
                nseg = nv;
                hipLaunchKernelGGL(ln_bwd_col_vec_kernel, dim3(cdiv(C, 512), R, nseg), dim3(128), 0, as_stream(stream), x, (const bf16_t*)dy, stats, partial, R, rows_per, C, nseg);
            } else {
                hipLaunchKernelGGL(ln_bwd_col_kernel<bf16_t>, dim3(cdiv(C, 256), R, RED_S), b256, 0, as_stream(stream), x, (const bf16_t*)dy, stats, partial, R, rows_per, C);
            }
        }
    } else if (dtype == CVAR_F32) {
        if (!ln_bwd_row_vec_launch<float>(x, (const float*)dy, scale, (long)ld_ada, rows_per, dx_in, dx_out, stats, M, C, eps, as_stream(stream)))
            hipLaunchKernelGGL(ln_bwd_row_kernel<float>, dim3(cdiv(M, 4)), b256, 0, as_stream(stream), x, (const float*)dy, scale, (long)ld_ada, rows_per, dx_in, dx_out, stats, M, C, eps);
        hipLaunchKernelGGL(ln_bwd_col_kernel<float>, dim3(cdiv(C, 256), R, RED_S), b256, 0, as_stream(stream), x, (const float*)dy, stats, partial, R, rows_per, C);
    } else return CVAR_EUNSUPPORTED;
    hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3(cdiv(C, 256), R), b256, 0, as_stream(stream), partial, dscale, dshift, (long)ldo, R, C, nseg);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// ---- column sum: out[n] (+)= sum_m A[m, n]   (bias gradients) ---------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ A, long lda, float* __restrict__ partial, long M, int N, int nseg) {
    const int n = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
    if (n >= N) return;
    const long seg = (M + nseg - 1) / nseg;
    const long m0 = s * seg, m1 = min(M, m0 + seg);
    float a = 0.f;
    for (long m = m0; m < m1; ++m) a += Elem<T>::ld(A + m * lda + n);
    partial[(long)s * N + n] = a;
}
// eight bf16 columns per thread (16-byte loads, 16 rows in flight); per-column order over the rows unchanged -> bit-identical to colsum_kernel
__global__ __launch_bounds__(64) void colsum_vec_kernel(const bf16_t* __restrict__ A, long lda, float* __restrict__ partial, long M, int N, int nseg) {
    const int n = (blockIdx.x * 64 + threadIdx.x) * 8, s = blockIdx.y;
    if (n >= N) return;
    const long seg = (M + nseg - 1) / nseg;
    const long m0 = s * seg, m1 = min(M, m0 + seg);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
    for (long m = m0; m < m1; ++m) {
        const bf16x8_t v = *(const bf16x8_t*)(A + m * lda + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += bf16_to_f32((bf16_t)v[e]);
    }
    const f32x4_t lo = {a[0], a[1], a[2], a[3]}, hi = {a[4], a[5], a[6], a[7]};
    *(f32x4_t*)(partial + (long)s * N + n) = lo;
    *(f32x4_t*)(partial + (long)s * N + n + 4) = hi;
}
__global__ void colsum_finalize_kernel(const float* __restrict__ partial, float* __restrict__ out, int N, int nseg, int accumulate) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float a = 0.f;
    for (int s = 0; s < nseg; ++s) a += partial[(long)s * N + n];
    out[n] = accumulate ? out[n] + a : a;
}
// ws: 64 * N floats
// ------------------------------------------------------------------------------------------------
// Gradient of word_embed = nn.Linear(Cvae = 32, C) (control_var.py:74) from the token-major tensors as they lie in memory:
//   dW[c][j] = sum_t dX[row(t)][c] * tok[t][j],   db[c] = sum_t dX[row(t)][c],   t < B * n  (row(t) skips the first `skip` rows of every sample).
// Round 2 did this with B transposes of dX into a [C][tokens] buffer, one transpose of tok, an fp32 GEMM with a 12-tile grid and K = 43 520
// (2.5 ms: no parallelism), and B column-sum launches for the bias - 3.9 ms = 1.3 % of a d24 training step.  Here: grid = (C / 64) x slices over the
// tokens; thread = (tok column j, group of 8 channels); per token 3 loads (tok[t][j] coalesced over j, 2 x 16 B of dX broadcast over j) and 8 fma;
// the slices' partials are summed in slice order by a second launch (fixed order: bit-reproducible).  fp32 throughout.
// ------------------------------------------------------------------------------------------------
constexpr int WEG_CT = 64;          // channels per workgroup
__global__ __launch_bounds__(256) void wordembed_grad_kernel(const float* __restrict__ dx, long ldx, int rows_per_sample, int skip, const float* __restrict__ tok,
                                                             int n, long ntok, int C, int tok_per_slice, float* __restrict__ part) {
    const int j = threadIdx.x & 31, cg = threadIdx.x >> 5;             // 8 channel groups x 8 channels
    const int c0 = blockIdx.x * WEG_CT + cg * 8;
    const long t0 = (long)blockIdx.y * tok_per_slice, t1 = min(ntok, t0 + tok_per_slice);
    float acc[8], accb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { acc[e] = 0.f; accb[e] = 0.f; }
    int b = (int)(t0 / n), r = (int)(t0 - (long)b * n);
    const float* xrow = dx + ((long)b * rows_per_sample + skip + r) * ldx + c0;
    for (long t = t0; t < t1; ++t) {
        const float w = tok[t * 32 + j];
        const f32x4_t x0 = *(const f32x4_t*)xrow, x1 = *(const f32x4_t*)(xrow + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[e] = fmaf(x0[e], w, acc[e]); acc[4 + e] = fmaf(x1[e], w, acc[4 + e]); accb[e] += x0[e]; accb[4 + e] += x1[e]; }
        xrow += ldx;
        if (++r == n) { r = 0; ++b; xrow = dx + ((long)b * rows_per_sample + skip) * ldx + c0; }
    }
    // partials: [slice][C][33] (column 32 = the bias sum, written by the j == 0 lanes)
    float* o = part + ((long)blockIdx.y * C + c0) * 33;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o[e * 33 + j] = acc[e];
        if (j == 0) o[e * 33 + 32] = accb[e];
    }
}
__global__ __launch_bounds__(256) void wordembed_grad_reduce_kernel(const float* __restrict__ part, int nslice, int C, float* __restrict__ dW, float* __restrict__ db) {
    const long total = (long)C * 33;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float v = part[i];
        for (int s2 = 1; s2 < nslice; ++s2) v += part[(long)s2 * total + i];
        const int c = (int)(i / 33), jj = (int)(i % 33);
        if (jj < 32) dW[(long)c * 32 + jj] = v; else db[c] = v;
    }
}
extern "C" int64_t cvar_wordembed_grad_ws_bytes(int64_t ntok, int C) {
    const int64_t nslice = ntok >= 32768 ? 32 : (ntok >= 4096 ? 16 : (ntok >= 256 ? 4 : 1));
    return nslice * (int64_t)C * 33 * (int64_t)sizeof(float);
}
extern "C" int cvar_wordembed_grad(const float* dx, int64_t ldx, int rows_per_sample, int skip, const float* tok, int n_per_sample, int B, int C, int Cvae,
                                   float* dW, float* db, float* ws, void* stream) {
    if (!dx || !tok || !dW || !db || !ws || B <= 0 || n_per_sample <= 0 || C <= 0 || rows_per_sample < skip + n_per_sample) return CVAR_EINVAL;
    if (Cvae != 32 || C % WEG_CT || (ldx & 3) || (((uintptr_t)dx | (uintptr_t)ws) & 15)) return CVAR_EUNSUPPORTED;
    const long ntok = (long)B * n_per_sample;
    const int nslice = ntok >= 32768 ? 32 : (ntok >= 4096 ? 16 : (ntok >= 256 ? 4 : 1));
    const int per = (int)((ntok + nslice - 1) / nslice);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(wordembed_grad_kernel, dim3(C / WEG_CT, nslice), dim3(256), 0, st, dx, (long)ldx, rows_per_sample, skip, tok, n_per_sample, ntok, C, per, ws);
    hipLaunchKernelGGL(wordembed_grad_reduce_kernel, dim3((unsigned)min((long)256, ((long)C * 33 + 255) / 256)), dim3(256), 0, st, (const float*)ws, nslice, C, dW, db);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

extern "C" int cvar_colsum(const void* A, int dtype, int64_t lda, float* out, int64_t M, int N, int accumulate, float* ws, void* stream) {
    if (!A || !out || !ws || M <= 0 || N <= 0) return CVAR_EINVAL;
    const int nseg = (int)min((int64_t)64, max((int64_t)1, M / 64));
    dim3 grid(cdiv(N, 256), nseg), block(256);
    if (dtype == CVAR_BF16 && N % 8 == 0 && lda % 8 == 0 && (((uintptr_t)A | (uintptr_t)ws) & 15) == 0)
        hipLaunchKernelGGL(colsum_vec_kernel, dim3(cdiv(N, 512), nseg), dim3(64), 0, as_stream(stream), (const bf16_t*)A, (long)lda, ws, (long)M, N, nseg);
    else if (dtype == CVAR_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, block, 0, as_stream(stream), (const bf16_t*)A, (long)lda, ws, (long)M, N, nseg);
    else if (dtype == CVAR_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, as_stream(stream), (const float*)A, (long)lda, ws, (long)M, N, nseg);
    else return CVAR_EUNSUPPORTED;
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(N, 256)), block, 0, as_stream(stream), ws, out, N, nseg, accumulate);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// ---- fused cross-entropy forward + backward over the 4096-way vocabulary ------------------------------------------
// loss_tok[m] = logsumexp(logits[m,:]) - logits[m, target[m]];  dlogits[m,v] = (softmax - onehot) * w[m] * gscale
template <typename TO>
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int* __restrict__ target, const float* __restrict__ weight,
                                                float gscale, float* __restrict__ loss_tok, TO* __restrict__ dlogits, int V) {
    __shared__ float red[4];
    const long m = blockIdx.x;
    const float* lr = logits + m * V;
    float mx = -INFINITY;
    for (int v = threadIdx.x; v < V; v += 256) mx = fmaxf(mx, lr[v]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float se = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) se += __expf(lr[v] - mx);
    se = wave_sum(se);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = se;
    __syncthreads();
    se = (red[0] + red[1]) + (red[2] + red[3]);
    const int tg = target[m];
    if (threadIdx.x == 0) loss_tok[m] = (logf(se) + mx) - lr[tg];
    if (dlogits) {
        const float w = (weight ? weight[m] : 1.0f) * gscale;
        const float inv = 1.0f / se;
        for (int v = threadIdx.x; v < V; v += 256) {
            const float pr = __expf(lr[v] - mx) * inv;
            Elem<TO>::st(dlogits + m * V + v, (pr - (v == tg ? 1.0f : 0.0f)) * w);
        }
    }
}
extern "C" int cvar_ce_fwd_bwd(const float* logits, const int32_t* target, const float* weight, float gscale, float* loss_tok,
                               void* dlogits, int out_dtype, int64_t M, int V, void* stream) {
    if (!logits || !target || !loss_tok || M <= 0 || V <= 1) return CVAR_EINVAL;
    dim3 grid((unsigned)M), block(256);
    if (!dlogits || out_dtype == CVAR_F32) hipLaunchKernelGGL(ce_kernel<float>, grid, block, 0, as_stream(stream), logits, target, weight, gscale, loss_tok, (float*)dlogits, V);
    else if (out_dtype == CVAR_BF16) hipLaunchKernelGGL(ce_kernel<bf16_t>, grid, block, 0, as_stream(stream), logits, target, weight, gscale, loss_tok, (bf16_t*)dlogits, V);
    else return CVAR_EUNSUPPORTED;
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// ---- embedding-row scatter (class_emb / cond_embed gradients): dst[idx[i], :] += src[i, :], rows applied in order ---
__global__ void scatter_add_rows_kernel(const float* __restrict__ src, long lds, const int* __restrict__ idx, float* __restrict__ dst, int n, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    for (int i = 0; i < n; ++i) dst[(long)idx[i] * C + c] += src[(long)i * lds + c];
}
extern "C" int cvar_scatter_add_rows(const float* src, int64_t ld_src, const int32_t* idx, float* dst, int n, int C, void* stream) {
    if (!src || !idx || !dst || n <= 0 || C <= 0) return CVAR_EINVAL;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(cdiv(C, 256)), dim3(256), 0, as_stream(stream), src, (long)ld_src, idx, dst, n, C);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// dcond = dsilu * silu'(cond),  silu'(x) = sig(x) * (1 + x * (1 - sig(x)))
__global__ void silu_bwd_kernel(const float* __restrict__ cond, const float* __restrict__ dsilu, float* __restrict__ dcond, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float x = cond[i], sg = 1.0f / (1.0f + __expf(-x));
        dcond[i] = dsilu[i] * (sg * (1.0f + x * (1.0f - sg)));
    }
}
extern "C" int cvar_silu_bwd(const float* cond, const float* dsilu, float* dcond, int64_t n, void* stream) {
    if (!cond || !dsilu || !dcond || n <= 0) return CVAR_EINVAL;
    hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)min((int64_t)2048, (n + 255) / 256)), dim3(256), 0, as_stream(stream), cond, dsilu, dcond, (long)n);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// ---- optimizer ----------------------------------------------------------------------------------------------------
// torch.optim.AdamW step on one tensor: g' = g * gscale (all-reduce mean and clip folded in), p *= 1 - lr*wd,
// m = b1 m + (1-b1) g', v = b2 v + (1-b2) g'^2, p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                             float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, const float* __restrict__ gscale_dev, float gscale) {
    const float gs = gscale * (gscale_dev ? gscale_dev[0] : 1.0f);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gr = g[i] * gs;
        float pv = p[i] * (1.0f - lr * wd);
        const float mv = b1 * m[i] + (1.0f - b1) * gr;
        const float vv = b2 * v[i] + (1.0f - b2) * gr * gr;
        pv -= (lr / bc1) * mv / (sqrtf(vv) / bc2_sqrt + eps);
        p[i] = pv; m[i] = mv; v[i] = vv;
    }
}
extern "C" int cvar_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int step, const float* gscale_dev, float gscale, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return CVAR_EINVAL;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)min((int64_t)4096, (n + 255) / 256)), dim3(256), 0, as_stream(stream), p, g, m, v, (long)n, lr, beta1,
                       beta2, eps, weight_decay, bc1, bc2s, gscale_dev, gscale);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// sum of squares of one tensor into 256 double partials at out[slot*256 ...] (fixed order)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, double* __restrict__ out) {
    __shared__ double red[4];
    double a = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const double v = x[i]; a += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
extern "C" int cvar_sumsq(const float* x, int64_t n, double* partial256, void* stream) {
    if (!x || !partial256 || n <= 0) return CVAR_EINVAL;
    hipLaunchKernelGGL(sumsq_kernel, dim3(256), dim3(256), 0, as_stream(stream), x, (long)n, partial256);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}
// total = sum(partials[0..count)); norm = pre_scale * sqrt(total) (pre_scale = 1/world after a SUM all-reduce); clip_coef = min(1, max_norm / (norm + 1e-6))  (clip_grad_norm_)
__global__ void clip_coef_kernel(const double* __restrict__ partials, long count, float pre_scale, float max_norm, float* __restrict__ out /*[2]: norm, coef*/) {
    __shared__ double red[4];
    double a = 0.0;
    for (long i = threadIdx.x; i < count; i += 256) a += partials[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double nrm = sqrt((red[0] + red[1]) + (red[2] + red[3])) * (double)pre_scale;
        out[0] = (float)nrm;
        out[1] = max_norm > 0.f ? (float)fmin(1.0, (double)max_norm / (nrm + 1e-6)) : 1.0f;
    }
}
extern "C" int cvar_clip_coef(const double* partials, int64_t count, float pre_scale, float max_norm, float* out2, void* stream) {
    if (!partials || !out2 || count <= 0) return CVAR_EINVAL;
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, as_stream(stream), partials, (long)count, pre_scale, max_norm, out2);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// ---- multi-tensor forms: one launch over a device table of tensors instead of one launch per parameter (a d24 model has ~830
// parameters; 2 x 830 tiny launches were 3 % of the training step).  Per tensor the arithmetic and the summation order are
// exactly those of cvar_sumsq / cvar_adamw (blockIdx.y selects the tensor, blockIdx.x plays the single-tensor grid).
struct AdamTensor { float* p; const float* g; float* m; float* v; long n; int group; int pad; bf16_t* w16; };

__global__ __launch_bounds__(256) void sumsq_multi_kernel(const AdamTensor* __restrict__ tab, double* __restrict__ out) {
    __shared__ double red[4];
    const AdamTensor t = tab[blockIdx.y];
    double a = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < t.n; i += (long)gridDim.x * 256) { const double v = t.g[i]; a += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[(long)blockIdx.y * 256 + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
extern "C" int cvar_sumsq_multi(const void* table_dev, int n_tensors, double* partials, void* stream) {
    if (!table_dev || !partials || n_tensors <= 0 || n_tensors > 65535) return CVAR_EINVAL;
    hipLaunchKernelGGL(sumsq_multi_kernel, dim3(256, (unsigned)n_tensors), dim3(256), 0, as_stream(stream), (const AdamTensor*)table_dev, partials);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

struct AdamGroups { float lr[8]; float wd[8]; };
__global__ void adamw_multi_kernel(const AdamTensor* __restrict__ tab, const AdamGroups gr, float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                   const float* __restrict__ gscale_dev, float gscale) {
    const AdamTensor t = tab[blockIdx.y];
    const float lr = gr.lr[t.group], wd = gr.wd[t.group];
    const float gs = gscale * (gscale_dev ? gscale_dev[0] : 1.0f);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < t.n; i += (long)gridDim.x * blockDim.x) {
        const float g = t.g[i] * gs;
        float pv = t.p[i] * (1.0f - lr * wd);
        const float mv = b1 * t.m[i] + (1.0f - b1) * g;
        const float vv = b2 * t.v[i] + (1.0f - b2) * g * g;
        pv -= (lr / bc1) * mv / (sqrtf(vv) / bc2_sqrt + eps);
        t.p[i] = pv; t.m[i] = mv; t.v[i] = vv;
        if (t.w16) t.w16[i] = f32_to_bf16(pv);      // the GEMM-ready bf16 copy of a weight matrix, refreshed while the value is in a register
    }
}
extern "C" int cvar_adamw_multi(const void* table_dev, int n_tensors, const float* lr_by_group_host, const float* wd_by_group_host, int n_groups,
                                float beta1, float beta2, float eps, int step, const float* gscale_dev, float gscale, void* stream) {
    if (!table_dev || !lr_by_group_host || !wd_by_group_host || n_tensors <= 0 || n_tensors > 65535 || n_groups <= 0 || n_groups > 8 || step < 1)
        return CVAR_EINVAL;
    AdamGroups gr;
    for (int i = 0; i < 8; ++i) { gr.lr[i] = i < n_groups ? lr_by_group_host[i] : 0.f; gr.wd[i] = i < n_groups ? wd_by_group_host[i] : 0.f; }
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(128, (unsigned)n_tensors), dim3(256), 0, as_stream(stream), (const AdamTensor*)table_dev, gr, beta1, beta2,
                       eps, bc1, bc2s, gscale_dev, gscale);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}
