// Row-wise / element-wise kernels of the transformer path (HBM-bound; 16-byte vector accesses).
#include "cvar_common.h"

// ------------------------------------------------------------------------------------------------
// adaLN: out = LN(x) * (1 + scale) + shift.   One wave per row, values kept in registers.
// ------------------------------------------------------------------------------------------------
// NV = float4 vectors per lane (C <= NV*256); TAIL = the last vector is only partly populated (C % 256 != 0).
// All loads of a row are issued back to back (no per-vector branches), so a wave has NV 16-byte loads in flight.
template <typename TO, int NV, bool TAIL>
__global__ __launch_bounds__(256) void ln_modulate_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, long ld_ada, int rows_per,
                                                         TO* __restrict__ out, int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (long)row * C;
    const long g = row / rows_per;
    const float* sc = scale + g * ld_ada;
    const float* sh = shift + g * ld_ada;
    f32x4_t v[NV], a[NV], b[NV];
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        const bool ok = !(TAIL && i == NV - 1) || c < C;
        v[i] = ok ? *(const f32x4_t*)(xr + c) : zero4;
        a[i] = ok ? *(const f32x4_t*)(sc + c) : zero4;
        b[i] = ok ? *(const f32x4_t*)(sh + c) : zero4;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        const bool ok = !(TAIL && i == NV - 1) || c < C;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][e] = ok ? v[i][e] - mean : 0.f; q += v[i][e] * v[i][e]; }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    TO* orow = out + (long)row * C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (TAIL && i == NV - 1 && c >= C) continue;
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (v[i][e] * rstd) * (1.0f + a[i][e]) + b[i][e];
        if constexpr (sizeof(TO) == 4) {
            f32x4_t o = {y[0], y[1], y[2], y[3]};
            *(f32x4_t*)(orow + c) = o;
        } else {
            *(bf16x4_t*)(orow + c) = pack_bf16x4(y);
        }
    }
}

template <typename TO, int NV>
static void ln_launch(bool tail, dim3 grid, hipStream_t st, const float* x, const float* scale, const float* shift, long ld_ada,
                      int rows_per, TO* out, int M, int C, float eps) {
    if (tail) hipLaunchKernelGGL((ln_modulate_kernel<TO, NV, true>), grid, dim3(256), 0, st, x, scale, shift, ld_ada, rows_per, out, M, C, eps);
    else hipLaunchKernelGGL((ln_modulate_kernel<TO, NV, false>), grid, dim3(256), 0, st, x, scale, shift, ld_ada, rows_per, out, M, C, eps);
}

template <typename TO>
static int ln_dispatch(const float* x, const float* scale, const float* shift, long ld_ada, int rows_per, TO* out, int M, int C,
                       float eps, hipStream_t st) {
    const int nv = (C + 255) / 256;
    const bool tail = (C % 256) != 0;
    dim3 grid(cdiv(M, 4));
    switch (nv) {
        case 1: ln_launch<TO, 1>(tail, grid, st, x, scale, shift, ld_ada, rows_per, out, M, C, eps); break;
        case 2: ln_launch<TO, 2>(tail, grid, st, x, scale, shift, ld_ada, rows_per, out, M, C, eps); break;
        case 3: ln_launch<TO, 3>(tail, grid, st, x, scale, shift, ld_ada, rows_per, out, M, C, eps); break;
        case 4: ln_launch<TO, 4>(tail, grid, st, x, scale, shift, ld_ada, rows_per, out, M, C, eps); break;
        case 5: ln_launch<TO, 5>(tail, grid, st, x, scale, shift, ld_ada, rows_per, out, M, C, eps); break;
        case 6: ln_launch<TO, 6>(tail, grid, st, x, scale, shift, ld_ada, rows_per, out, M, C, eps); break;
        case 7: ln_launch<TO, 7>(tail, grid, st, x, scale, shift, ld_ada, rows_per, out, M, C, eps); break;
        case 8: ln_launch<TO, 8>(tail, grid, st, x, scale, shift, ld_ada, rows_per, out, M, C, eps); break;
        default: return CVAR_EUNSUPPORTED;
    }
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

extern "C" int cvar_ln_modulate(const float* x, const float* scale, const float* shift, int64_t ld_ada, int rows_per,
                                void* out, int out_dtype, int M, int C, float eps, void* stream) {
    if (!x || !scale || !shift || !out || M <= 0 || rows_per <= 0) return CVAR_EINVAL;
    if (C % 4 || C > 2048 || ld_ada % 4) return CVAR_EUNSUPPORTED;
    if (out_dtype == CVAR_BF16) return ln_dispatch<bf16_t>(x, scale, shift, (long)ld_ada, rows_per, (bf16_t*)out, M, C, eps, as_stream(stream));
    if (out_dtype == CVAR_F32) return ln_dispatch<float>(x, scale, shift, (long)ld_ada, rows_per, (float*)out, M, C, eps, as_stream(stream));
    return CVAR_EUNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
template <typename TO>
__global__ void silu_cast_kernel(const float* __restrict__ x, TO* __restrict__ out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float v = x[i];
        Elem<TO>::st(out + i, v / (1.0f + __expf(-v)));
    }
}

extern "C" int cvar_silu_cast(const float* x, void* out, int out_dtype, int64_t n, void* stream) {
    if (!x || !out || n <= 0) return CVAR_EINVAL;
    dim3 grid((unsigned)min((int64_t)2048, (n + 255) / 256)), block(256);
    if (out_dtype == CVAR_BF16) hipLaunchKernelGGL(silu_cast_kernel<bf16_t>, grid, block, 0, as_stream(stream), x, (bf16_t*)out, (long)n);
    else if (out_dtype == CVAR_F32) hipLaunchKernelGGL(silu_cast_kernel<float>, grid, block, 0, as_stream(stream), x, (float*)out, (long)n);
    else return CVAR_EUNSUPPORTED;
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// ------------------------------------------------------------------------------------------------
// cos-attention pre-pass: q <- normalize(q) * exp(min(scale_mul_h, ln 100)), k <- normalize(k), in place.
// One wave per (sequence, token, head, q|k); lane = channel.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cos_qk_norm_kernel(T* __restrict__ qkv, T* __restrict__ qs, int R, int H, int Lmax, int q_off, int l,
                                                         const float* __restrict__ scale_mul, float* __restrict__ norms, float q_mul) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);     // over R*l*H*2
    const long total = (long)R * l * H * 2;
    if (item >= total) return;
    const int which = (int)(item & 1);
    const int h = (int)((item >> 1) % H);
    const long rt = (item >> 1) / H;
    const int t = (int)(rt % l);
    const long r = rt / l;
    T* p;
    if (qs) p = (which == 0 ? qs + (r * l + t) * (long)(H * 64) : qkv + (r * Lmax + q_off + t) * 2 * (long)(H * 64)) + h * 64 + lane;   // K/V arena + separate queries
    else p = qkv + ((r * Lmax + q_off + t) * 3 + which) * (long)(H * 64) + h * 64 + lane;
    const float v = Elem<T>::ld(p);
    const float nrm = fmaxf(sqrtf(wave_sum(v * v)), 1e-12f);      // F.normalize eps
    float o = v / nrm;
    if (which == 0) o *= __expf(fminf(scale_mul[h], 4.605170185988092f)) * q_mul;
    Elem<T>::st(p, o);
    if (norms && lane == 0) norms[item] = nrm;                     // [R][l][H][q|k], saved for the backward pass
}

extern "C" int cvar_cos_qk_norm(void* qkv, void* q, int dtype, int R, int H, int Lmax, int q_off, int l, const float* scale_mul,
                                float* norms, float q_mul, void* stream) {
    if (!qkv || !scale_mul || R <= 0 || H <= 0 || l <= 0) return CVAR_EINVAL;
    const long total = (long)R * l * H * 2;
    dim3 grid(cdiv(total, 4)), block(256);
    if (dtype == CVAR_BF16) hipLaunchKernelGGL(cos_qk_norm_kernel<bf16_t>, grid, block, 0, as_stream(stream), (bf16_t*)qkv, (bf16_t*)q, R, H, Lmax, q_off, l, scale_mul, norms, q_mul);
    else if (dtype == CVAR_F32) hipLaunchKernelGGL(cos_qk_norm_kernel<float>, grid, block, 0, as_stream(stream), (float*)qkv, (float*)q, R, H, Lmax, q_off, l, scale_mul, norms, q_mul);
    else return CVAR_EUNSUPPORTED;
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// backward of the cos-attention pre-pass, in place on dqkv (gradients w.r.t. q_hat*sm, k_hat -> w.r.t. raw q, k):
//   x_t = x / |x| ;  dx = (g' - x_t (x_t . g')) / |x|  with g' = g * sm for q, g for k;  dsm[token, head] = g . x_t  (q only)
template <typename T>
__global__ __launch_bounds__(256) void cos_qk_norm_bwd_kernel(const T* __restrict__ qkv, T* __restrict__ dqkv, int R, int H, int Lmax, int l,
                                                             const float* __restrict__ scale_mul, const float* __restrict__ norms,
                                                             float* __restrict__ dsm_tok /*[R*l][H]*/) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long total = (long)R * l * H * 2;
    if (item >= total) return;
    const int which = (int)(item & 1);
    const int h = (int)((item >> 1) % H);
    const long rt = (item >> 1) / H;
    const int t = (int)(rt % l);
    const long r = rt / l;
    const long off = ((r * Lmax + t) * 3 + which) * (long)(H * 64) + h * 64 + lane;
    const float sm = __expf(fminf(scale_mul[h], 4.605170185988092f));
    const float xh = Elem<T>::ld(qkv + off);                        // normalized (and, for q, scaled) value kept by the forward
    const float g = Elem<T>::ld(dqkv + off);
    const float nrm = norms[item];
    const float xt = which == 0 ? xh / sm : xh;
    const float gp = which == 0 ? g * sm : g;
    const float dot = wave_sum(xt * gp);
    Elem<T>::st(dqkv + off, (gp - xt * dot) / nrm);
    if (which == 0) {
        const float ds = wave_sum(g * xt);
        // chain rule through sm = exp(min(s, ln 100)): d sm / d s = sm below the clamp, 0 above it
        if (lane == 0) dsm_tok[rt * H + h] = scale_mul[h] < 4.605170185988092f ? ds * sm : 0.f;
    }
}

extern "C" int cvar_cos_qk_norm_bwd(const void* qkv, void* dqkv, int dtype, int R, int H, int Lmax, int l, const float* scale_mul,
                                    const float* norms, float* dsm_tok, void* stream) {
    if (!qkv || !dqkv || !scale_mul || !norms || !dsm_tok || R <= 0 || H <= 0 || l <= 0) return CVAR_EINVAL;
    const long total = (long)R * l * H * 2;
    dim3 grid(cdiv(total, 4)), block(256);
    if (dtype == CVAR_BF16) hipLaunchKernelGGL(cos_qk_norm_bwd_kernel<bf16_t>, grid, block, 0, as_stream(stream), (const bf16_t*)qkv, (bf16_t*)dqkv, R, H, Lmax, l, scale_mul, norms, dsm_tok);
    else if (dtype == CVAR_F32) hipLaunchKernelGGL(cos_qk_norm_bwd_kernel<float>, grid, block, 0, as_stream(stream), (const float*)qkv, (float*)dqkv, R, H, Lmax, l, scale_mul, norms, dsm_tok);
    else return CVAR_EUNSUPPORTED;
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// ------------------------------------------------------------------------------------------------
// word_embed + lvl_pos (fp32): x[rep*nb+b][t][c] = bias[c] + lvl_pos[t][c] + sum_k tok[b][t][k] W[c][k]
// ------------------------------------------------------------------------------------------------
// A thread owns ONE output channel (its 32-float weight row stays in registers) and walks WE_TOK tokens of the block's tile, whose token vectors
// sit in LDS (broadcast reads).  The token-per-block form re-read the whole 196 KB weight matrix from L2 for every 6 KB output row.  Same fma
// chain per output as before: bit-identical.
constexpr int WE_TOK = 64;
__global__ __launch_bounds__(256) void word_embed_kernel(const float* __restrict__ tok, const float* __restrict__ W,
                                                        const float* __restrict__ bias, const float* __restrict__ lvl_pos,
                                                        float* __restrict__ x, int nb, int nrep, int l, int Cvae, int C,
                                                        int x_rows, int x_off, int tpb) {
    __shared__ __attribute__((aligned(16))) float tk[WE_TOK * 64];
    const long ntok = (long)nb * l;
    const long bt0 = (long)blockIdx.x * tpb;            // tpb <= WE_TOK tokens per block
    const int nt = (int)min((long)tpb, ntok - bt0);
    for (int i = threadIdx.x; i < nt * Cvae; i += 256) tk[i] = tok[bt0 * Cvae + i];
    __syncthreads();
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    f32x4_t wv[16];
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
        const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
        wv[k4] = 4 * k4 < Cvae ? *(const f32x4_t*)(W + (long)c * Cvae + 4 * k4) : z;
    }
    const float bc = bias[c];
    for (int i = 0; i < nt; ++i) {
        const long bt = bt0 + i;
        const int t = (int)(bt % l);
        const long b = bt / l;
        const float* tv = tk + i * Cvae;
        float acc = 0.f;
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4) {
            if (4 * k4 < Cvae) {
                const f32x4_t q = *(const f32x4_t*)(tv + 4 * k4);
                acc = fmaf(q[0], wv[k4][0], acc); acc = fmaf(q[1], wv[k4][1], acc);
                acc = fmaf(q[2], wv[k4][2], acc); acc = fmaf(q[3], wv[k4][3], acc);
            }
        }
        const float v = (acc + bc) + lvl_pos[(long)t * C + c];
        for (int rep = 0; rep < nrep; ++rep) x[(((long)rep * nb + b) * x_rows + x_off + t) * C + c] = v;
    }
}

extern "C" int cvar_word_embed(const float* tok, const float* W, const float* bias, const float* lvl_pos, float* x,
                               int nb, int nrep, int l, int Cvae, int C, int x_rows, int x_off, void* stream) {
    if (!tok || !W || !bias || !lvl_pos || !x || nb <= 0 || nrep <= 0 || l <= 0) return CVAR_EINVAL;
    if (x_rows < x_off + l || x_off < 0) return CVAR_EINVAL;
    if (Cvae > 64 || Cvae % 4) return CVAR_EUNSUPPORTED;
    const long ntok = (long)nb * l;
    // tokens per block: 64 amortise the weight rows a thread keeps in registers; few tokens in all (small batches: a B = 1 generation spent 52 us per call in
    // 8 x 6 blocks walking 64 tokens each) are spread over more blocks instead - at least ~512 blocks before the per-block count grows
    int tpb = WE_TOK;
    while (tpb > 4 && ((ntok + tpb - 1) / tpb) * cdiv(C, 256) < 512) tpb >>= 1;
    hipLaunchKernelGGL(word_embed_kernel, dim3((unsigned)((ntok + tpb - 1) / tpb), (unsigned)cdiv(C, 256)), dim3(256), 0, as_stream(stream), tok, W,
                       bias, lvl_pos, x, nb, nrep, l, Cvae, C, x_rows, x_off, tpb);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// first-scale tokens + adaLN condition
__global__ void first_tokens_kernel(const float* __restrict__ class_emb, const float* __restrict__ cond_embed,
                                    const int* __restrict__ labels, const int* __restrict__ types,
                                    const float* __restrict__ pos_start, const float* __restrict__ lvl_pos,
                                    float* __restrict__ x, float* __restrict__ cond, int R, int first_l, int C, int x_rows) {
    const int r = blockIdx.x;
    const float* ce = class_emb + (long)labels[r] * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float cls = ce[c];
        cond[(long)r * C + c] = cls;
        if (first_l == 2) {
            const float ct = cond_embed[(long)types[r] * C + c];
            x[((long)r * x_rows + 0) * C + c] = (ct + pos_start[c]) + lvl_pos[c];
            x[((long)r * x_rows + 1) * C + c] = (cls + pos_start[C + c]) + lvl_pos[C + c];
        } else {
            x[(long)r * x_rows * C + c] = (cls + pos_start[c]) + lvl_pos[c];
        }
    }
}

extern "C" int cvar_first_tokens(const float* class_emb, const float* cond_embed, const int32_t* labels, const int32_t* types,
                                 const float* pos_start, const float* lvl_pos, float* x, float* cond, int R, int first_l,
                                 int C, int x_rows, void* stream) {
    if (!class_emb || !labels || !pos_start || !lvl_pos || !x || !cond || R <= 0) return CVAR_EINVAL;
    if (first_l != 1 && first_l != 2) return CVAR_EUNSUPPORTED;
    if (x_rows < first_l) return CVAR_EINVAL;
    if (first_l == 2 && (!cond_embed || !types)) return CVAR_EINVAL;
    hipLaunchKernelGGL(first_tokens_kernel, dim3(R), dim3(256), 0, as_stream(stream), class_emb, cond_embed, labels, types, pos_start, lvl_pos, x, cond, R, first_l, C, x_rows);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// ------------------------------------------------------------------------------------------------
// VQVAE helpers
// ------------------------------------------------------------------------------------------------
// GroupNorm statistics over NHWC: per-(image, pixel-chunk) per-channel sum / sum of squares of (x - pivot_c), reduced in a fixed
// order (no atomics -> bit-reproducible), then combined over chunks in double by gn_finalize_kernel.  pivot_c = x[b][pixel 0][c]:
// accumulating around a value of the channel's own magnitude removes the cancellation of a raw sum / sum-of-squares pass when
// |mean| >> std (round-1 limit: 2e-3 .. 7e-2 on degenerate groups); the group statistics are reassembled exactly in double.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, float* __restrict__ partial, int HW, int C, int pix_per_block) {
    constexpr int VEC = 16 / sizeof(T);
    extern __shared__ float sred[];          // [VEC][PL][C / VEC][2]
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int ncg = C / VEC;                 // vectors per pixel
    const int PL = 256 / ncg;                // pixels in flight per block iteration
    const int pl = threadIdx.x / ncg, cg = threadIdx.x % ncg;
    if (pl < PL) {
        float s[VEC], q[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { s[e] = 0.f; q[e] = 0.f; }
        const int p0 = chunk * pix_per_block;
        const int p1 = min(p0 + pix_per_block, HW);
        float piv[VEC];
        {
            const T* src = x + (long)b * HW * C + cg * VEC;
            if constexpr (sizeof(T) == 2) {
                const bf16x8_t v = *(const bf16x8_t*)src;
#pragma unroll
                for (int e = 0; e < VEC; ++e) piv[e] = bf16_to_f32((bf16_t)v[e]);
            } else {
                const f32x4_t v = *(const f32x4_t*)src;
#pragma unroll
                for (int e = 0; e < VEC; ++e) piv[e] = v[e];
            }
        }
#pragma unroll 4
        for (int p = p0 + pl; p < p1; p += PL) {
            const T* src = x + ((long)b * HW + p) * C + cg * VEC;
            if constexpr (sizeof(T) == 2) {
                const bf16x8_t v = *(const bf16x8_t*)src;
#pragma unroll
                for (int e = 0; e < VEC; ++e) { const float f = bf16_to_f32((bf16_t)v[e]) - piv[e]; s[e] += f; q[e] += f * f; }
            } else {
                const f32x4_t v = *(const f32x4_t*)src;
#pragma unroll
                for (int e = 0; e < VEC; ++e) { const float f = v[e] - piv[e]; s[e] += f; q[e] += f * f; }
            }
        }
        // [e][pl][cg] pairs: consecutive lanes (cg) write consecutive 8-byte slots (the [pl][c][2] form strode lanes by 64 B: every write a
        // 16-way bank conflict - 88 % of the kernel's LDS cycles were conflict cycles)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float2 sq; sq.x = s[e]; sq.y = q[e];
            *(float2*)(sred + ((long)(e * PL + pl) * ncg + cg) * 2) = sq;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int c = i >> 1, which = i & 1, cg2 = c / VEC, e = c % VEC;
        float a = 0.f;
        for (int k = 0; k < PL; ++k) a += sred[((long)(e * PL + k) * ncg + cg2) * 2 + which];      // same order over k as before: bit-identical
        partial[(((long)b * nchunk + chunk) * C) * 2 + i] = a;
    }
}

// per (b, group): one wave reduces the chunk partials of the group's channels in double (fixed shuffle order ->
// reproducible), then writes a = rstd_g * w_c, d = bias_c - mean_g * rstd_g * w_c for its channels.
template <typename T>
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ partial, int nchunk, const T* __restrict__ x, const float* __restrict__ weight,
                                                        const float* __restrict__ bias, float* __restrict__ coef, int HW, int C, int groups, float eps) {
    const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x;
    const int cpg = C / groups;
    // pass 1: group mean = sum_c (HW * pivot_c + S_c) / n with S_c = sum of (x - pivot_c) over the channel
    double s = 0.0;
    const int items = cpg * nchunk;
    for (int it = lane; it < items; it += 64) {
        const int k = it / cpg, cc = g * cpg + it % cpg;
        s += (double)partial[(((long)b * nchunk + k) * C + cc) * 2];
    }
    for (int cc = g * cpg + lane; cc < (g + 1) * cpg; cc += 64) s += (double)HW * (double)Elem<T>::ld(x + (long)b * HW * C + cc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const double n = (double)HW * cpg;
    const double mean = s / n;
    // pass 2: sum (x - mean)^2 = sum_c [ Q_c + 2 S_c d_c + HW d_c^2 ],  d_c = pivot_c - mean  (exact identity, every term small)
    double q = 0.0;
    for (int it = lane; it < items; it += 64) {
        const int k = it / cpg, cc = g * cpg + it % cpg;
        const double d = (double)Elem<T>::ld(x + (long)b * HW * C + cc) - mean;
        q += (double)partial[(((long)b * nchunk + k) * C + cc) * 2 + 1] + 2.0 * d * (double)partial[(((long)b * nchunk + k) * C + cc) * 2];
    }
    for (int cc = g * cpg + lane; cc < (g + 1) * cpg; cc += 64) {
        const double d = (double)Elem<T>::ld(x + (long)b * HW * C + cc) - mean;
        q += (double)HW * d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    double var = q / n;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int cc = g * cpg + lane; cc < (g + 1) * cpg; cc += 64) {
        const float a = rstd * weight[cc];
        coef[((long)b * C + cc) * 2] = a;
        coef[((long)b * C + cc) * 2 + 1] = bias[cc] - (float)mean * a;
    }
}

// y = silu?(x * a_c + d_c): block = (pixel chunk, image); a thread owns one 16-byte channel group (its 2 x VEC coefficients stay in
// registers) and walks the chunk's pixels PL rows at a time - no index arithmetic in the loop (the flat-index form spent its time in
// 64-bit div / mod: 4.2 TB/s effective over stats + apply), four independent 16-byte loads in flight per thread.
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ coef, T* __restrict__ out,
                                                      int HW, int C, int ppb, int silu) {
    constexpr int VEC = 16 / sizeof(T);
    const int ncg = C / VEC, PL = 256 / ncg;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int cg = threadIdx.x % ncg, pl = threadIdx.x / ncg;
    if (pl >= PL) return;
    float a[VEC], d[VEC];
    {
        const float* cf = coef + ((long)b * C + cg * VEC) * 2;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { a[e] = cf[2 * e]; d[e] = cf[2 * e + 1]; }
    }
    const int p0 = chunk * ppb, p1 = min(HW, p0 + ppb);
    const long base = (long)b * HW * C + cg * VEC;
#pragma unroll 4
    for (int p = p0 + pl; p < p1; p += PL) {
        const long off = base + (long)p * C;
        float y[VEC];
        if constexpr (sizeof(T) == 2) {
            const bf16x8_t v = *(const bf16x8_t*)(x + off);
#pragma unroll
            for (int e = 0; e < VEC; ++e) y[e] = bf16_to_f32((bf16_t)v[e]) * a[e] + d[e];
        } else {
            const f32x4_t v = *(const f32x4_t*)(x + off);
#pragma unroll
            for (int e = 0; e < VEC; ++e) y[e] = v[e] * a[e] + d[e];
        }
        if (silu) {
            // bf16 mode: v_exp_f32 + v_rcp_f32 (about 1 ulp of fp32 each, far inside the bf16 rounding step that follows) - the IEEE division
            // made the pass VALU-bound at 4.2 TB/s; the fp32 parity mode keeps the exact quotient
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                if constexpr (sizeof(T) == 2) y[e] = y[e] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * y[e]));
                else y[e] = y[e] / (1.0f + __expf(-y[e]));
            }
        }
        if constexpr (sizeof(T) == 2) {
            *(bf16x8_t*)(out + off) = pack_bf16x8(y);
        } else {
            f32x4_t o = {y[0], y[1], y[2], y[3]};
            *(f32x4_t*)(out + off) = o;
        }
    }
}

// per (b, group): the conv epilogue's per-tile partials (sum (y - piv_t), sum (y - piv_t)^2, piv_t) combined in double - every tile has its own pivot:
//   sum y = sum_t (n piv_t + S_t),   sum (y - mean)^2 = sum_t [ Q_t + 2 S_t d_t + n d_t^2 ],  d_t = piv_t - mean   (exact identities, every term small)
__global__ __launch_bounds__(64) void gn_finalize_tiles_kernel(const float* __restrict__ part, int ntile, int npix, const float* __restrict__ weight,
                                                              const float* __restrict__ bias, float* __restrict__ coef, int HW, int C, int groups, float eps) {
    const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x;
    const int cpg = C / groups;
    const int items = cpg * ntile;
    const double n = (double)npix;
    double s = 0.0;
    for (int it = lane; it < items; it += 64) {
        const int k = it / cpg, cc = g * cpg + it % cpg;
        const float* q = part + (((long)b * ntile + k) * C + cc) * 3;
        s += n * (double)q[2] + (double)q[0];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const double cnt = (double)HW * cpg;
    const double mean = s / cnt;
    double v = 0.0;
    for (int it = lane; it < items; it += 64) {
        const int k = it / cpg, cc = g * cpg + it % cpg;
        const float* q = part + (((long)b * ntile + k) * C + cc) * 3;
        const double d = (double)q[2] - mean;
        v += (double)q[1] + 2.0 * d * (double)q[0] + n * d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    double var = v / cnt;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int cc = g * cpg + lane; cc < (g + 1) * cpg; cc += 64) {
        const float a = rstd * weight[cc];
        coef[((long)b * C + cc) * 2] = a;
        coef[((long)b * C + cc) * 2 + 1] = bias[cc] - (float)mean * a;
    }
}

static inline int gn_pix_per_block(int HW) { return HW >= 16384 ? 512 : (HW >= 1024 ? 128 : 32); }

extern "C" int64_t cvar_groupnorm_ws_bytes(int B, int HW, int C) {
    const int64_t nchunk = cdiv(HW, gn_pix_per_block(HW));
    return ((int64_t)B * nchunk * C * 2 + (int64_t)B * C * 2) * (int64_t)sizeof(float);
}

template <typename T>
static int groupnorm_typed(const T* x, const float* weight, const float* bias, T* out, int B, int HW, int C, int groups,
                           float eps, int silu, void* ws, hipStream_t st) {
    constexpr int VEC = 16 / sizeof(T);
    if (C % VEC || C % groups || C / VEC > 256) return CVAR_EUNSUPPORTED;
    const int ppb = gn_pix_per_block(HW);
    const int nchunk = cdiv(HW, ppb);
    float* partial = (float*)ws;
    float* coef = partial + (size_t)B * nchunk * C * 2;
    const int PL = 256 / (C / VEC);
    hipLaunchKernelGGL(gn_stats_kernel<T>, dim3(nchunk, B), dim3(256), (size_t)PL * C * 2 * sizeof(float), st, x, partial, HW, C, ppb);
    hipLaunchKernelGGL(gn_finalize_kernel<T>, dim3(groups, B), dim3(64), 0, st, partial, nchunk, x, weight, bias, coef, HW, C, groups, eps);
    hipLaunchKernelGGL(gn_apply_kernel<T>, dim3(nchunk, B), dim3(256), 0, st, x, coef, out, HW, C, ppb, silu);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

extern "C" int cvar_groupnorm_silu_partials(const void* x, int dtype, const float* weight, const float* bias, void* out, int B, int HW, int C, int groups, float eps,
                                            int silu, const float* gn_part, int tiles_per_image, int pixels_per_tile, void* ws, void* stream) {
    if (!x || !weight || !bias || !out || !ws || !gn_part || B <= 0 || HW <= 0 || C <= 0 || groups <= 0 || tiles_per_image <= 0 || pixels_per_tile <= 0) return CVAR_EINVAL;
    if ((long)tiles_per_image * pixels_per_tile != HW) return CVAR_EINVAL;                 // the partials must cover the image exactly
    if (dtype != CVAR_BF16) return CVAR_EUNSUPPORTED;                                       // only the bf16 halo conv emits partials
    constexpr int VEC = 8;
    if (C % VEC || C % groups || C / VEC > 256) return CVAR_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int ppb = gn_pix_per_block(HW);
    const int nchunk = cdiv(HW, ppb);
    float* coef = (float*)ws + (size_t)B * nchunk * C * 2;                                  // same place as in cvar_groupnorm_silu's workspace
    hipLaunchKernelGGL(gn_finalize_tiles_kernel, dim3(groups, B), dim3(64), 0, st, gn_part, tiles_per_image, pixels_per_tile, weight, bias, coef, HW, C, groups, eps);
    hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, dim3(nchunk, B), dim3(256), 0, st, (const bf16_t*)x, coef, (bf16_t*)out, HW, C, ppb, silu);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

extern "C" int cvar_groupnorm_silu(const void* x, int dtype, const float* weight, const float* bias, void* out,
                                   int B, int HW, int C, int groups, float eps, int silu, void* ws, void* stream) {
    if (!x || !weight || !bias || !out || !ws || B <= 0 || HW <= 0 || C <= 0 || groups <= 0) return CVAR_EINVAL;
    if (dtype == CVAR_BF16) return groupnorm_typed<bf16_t>((const bf16_t*)x, weight, bias, (bf16_t*)out, B, HW, C, groups, eps, silu, ws, as_stream(stream));
    if (dtype == CVAR_F32) return groupnorm_typed<float>((const float*)x, weight, bias, (float*)out, B, HW, C, groups, eps, silu, ws, as_stream(stream));
    return CVAR_EUNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 operands ("bf16x3", ABI 20): the middle precision of the VQVAE encoder.  An fp32 value x is carried as hi = bf16(x) and lo = bf16(x - hi)
// (x - hi is exact in fp32; |x - hi - lo| <= 2^-17 |x|), and a product a * w is taken as a_hi w_hi + a_lo w_hi + a_hi w_lo on the bf16 matrix pipe with fp32
// accumulation (the dropped a_lo w_lo term is <= 2^-16 |a w|).  The three products are three K SEGMENTS of one conv / GEMM: an activation row of C channels is
// stored as [hi(C) | lo(C) | hi(C)] bf16 and the weight row as [w_hi | w_hi | w_lo], so the existing bf16 kernels run the sum with 3 x the K extent - no new
// contraction kernel.  These two kernels produce the activation side: from an fp32 tensor as it is, or from GroupNorm (+ SiLU) of it (the apply pass of
// cvar_groupnorm_silu in fp32 arithmetic with the exact quotient, as the fp32 parity mode runs it).  Cpad >= 3 C pads a pixel's row with zeros (the image conv: 3 -> 4
// channels fp32, 12 split channels padded to 16).
__device__ __forceinline__ void split3_store(bf16_t* __restrict__ o, const float* y, int C) {
    float hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const bf16_t h = f32_to_bf16(y[e]); hi[e] = bf16_to_f32(h); lo[e] = y[e] - hi[e]; }
    const bf16x4_t H = pack_bf16x4(hi), L = pack_bf16x4(lo);          // hi is a bf16 value already: packing it is exact
    *(bf16x4_t*)(o) = H;
    *(bf16x4_t*)(o + C) = L;
    *(bf16x4_t*)(o + 2 * C) = H;
}

__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, long ldx, bf16_t* __restrict__ out, long M, int C, int Cpad) {
    const int nq = C / 4, npad = (Cpad - 3 * C) / 4;
    const long total = M * (nq + npad);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / (nq + npad);
        const int q = (int)(i % (nq + npad));
        if (q < nq) {
            const f32x4_t v = *(const f32x4_t*)(x + m * ldx + q * 4);
            const float y[4] = {v[0], v[1], v[2], v[3]};
            split3_store(out + m * Cpad + q * 4, y, C);
        } else {
            const bf16x4_t z = {0, 0, 0, 0};
            *(bf16x4_t*)(out + m * Cpad + 3 * C + (q - nq) * 4) = z;
        }
    }
}

extern "C" int cvar_split3(const float* x, int64_t ldx, void* out, int64_t M, int C, int Cpad, void* stream) {
    if (!x || !out || M <= 0 || C <= 0) return CVAR_EINVAL;
    if (C % 4 || ldx % 4 || Cpad < 3 * C || Cpad % 4 || (((uintptr_t)x | (uintptr_t)out) & 15)) return CVAR_EUNSUPPORTED;
    const long total = M * (long)(Cpad - 2 * C) / 4;
    const int blocks = (int)std::min<long>(cdiv(total, 256L), 256L * 32);
    hipLaunchKernelGGL(split3_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, (long)ldx, (bf16_t*)out, (long)M, C, Cpad);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// gn_apply_kernel<float> with the split store: y = silu?(x a_c + d_c) -> [hi | lo | hi]
__global__ __launch_bounds__(256) void gn_apply_split3_kernel(const float* __restrict__ x, const float* __restrict__ coef, bf16_t* __restrict__ out,
                                                             int HW, int C, int ppb, int silu) {
    const int ncg = C / 4, PL = 256 / ncg;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int cg = threadIdx.x % ncg, pl = threadIdx.x / ncg;
    if (pl >= PL) return;
    float a[4], d[4];
    {
        const float* cf = coef + ((long)b * C + cg * 4) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = cf[2 * e]; d[e] = cf[2 * e + 1]; }
    }
    const int p0 = chunk * ppb, p1 = min(HW, p0 + ppb);
#pragma unroll 4
    for (int p = p0 + pl; p < p1; p += PL) {
        const long row = (long)b * HW + p;
        const f32x4_t v = *(const f32x4_t*)(x + row * C + cg * 4);
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = v[e] * a[e] + d[e];
        if (silu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = y[e] / (1.0f + __expf(-y[e]));
        }
        split3_store(out + row * 3 * C + cg * 4, y, C);
    }
}

extern "C" int cvar_groupnorm_silu_split3(const float* x, const float* weight, const float* bias, void* out, int B, int HW, int C, int groups, float eps, int silu,
                                          void* ws, void* stream) {
    if (!x || !weight || !bias || !out || !ws || B <= 0 || HW <= 0 || C <= 0 || groups <= 0) return CVAR_EINVAL;
    if (C % 4 || C % groups || C / 4 > 256 || (((uintptr_t)x | (uintptr_t)out) & 15)) return CVAR_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int ppb = gn_pix_per_block(HW);
    const int nchunk = cdiv(HW, ppb);
    float* partial = (float*)ws;
    float* coef = partial + (size_t)B * nchunk * C * 2;
    const int PL = 256 / (C / 4);
    hipLaunchKernelGGL(gn_stats_kernel<float>, dim3(nchunk, B), dim3(256), (size_t)PL * C * 2 * sizeof(float), st, x, partial, HW, C, ppb);
    hipLaunchKernelGGL(gn_finalize_kernel<float>, dim3(groups, B), dim3(64), 0, st, partial, nchunk, x, weight, bias, coef, HW, C, groups, eps);
    hipLaunchKernelGGL(gn_apply_split3_kernel, dim3(nchunk, B), dim3(256), 0, st, x, coef, (bf16_t*)out, HW, C, ppb, silu);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// row softmax, one wave per row (cols <= 1024)
template <typename TO>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, TO* __restrict__ p, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* sr = s + row * cols;
    float v[16];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = i * 64 + lane;
        v[i] = c < cols ? sr[c] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = __expf(v[i] - mx); sum += v[i]; }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = i * 64 + lane;
        if (c < cols) Elem<TO>::st(p + row * cols + c, v[i] * inv);
    }
}

extern "C" int cvar_softmax_rows(const float* s, void* p, int out_dtype, int rows, int cols, void* stream) {
    if (!s || !p || rows <= 0 || cols <= 0) return CVAR_EINVAL;
    if (cols > 1024) return CVAR_EUNSUPPORTED;
    dim3 grid(cdiv(rows, 4)), block(256);
    if (out_dtype == CVAR_BF16) hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, grid, block, 0, as_stream(stream), s, (bf16_t*)p, rows, cols);
    else if (out_dtype == CVAR_F32) hipLaunchKernelGGL(softmax_rows_kernel<float>, grid, block, 0, as_stream(stream), s, (float*)p, rows, cols);
    else return CVAR_EUNSUPPORTED;
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// [B][n][c] (row stride ld_in) -> [B][c][ld_out >= n]   64 x 64 tiles, 16-byte global accesses on both sides
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, T* __restrict__ out, int n, int c, long ld_in, long ld_out) {
    constexpr int VEC = 16 / sizeof(T);          // elements per 16-byte access
    constexpr int TS = 64;                       // tile side
    constexpr int TPR = TS / VEC;                // threads per tile row
    constexpr int RPP = 256 / TPR;               // tile rows per pass
    __shared__ T tile[TS][TS + 2 * VEC / VEC + (sizeof(T) == 2 ? 2 : 1)];   // padded rows (odd dword stride)
    const long b = blockIdx.z;
    const int n0 = blockIdx.x * TS, c0 = blockIdx.y * TS;
    const int tr = threadIdx.x / TPR, tv = (threadIdx.x % TPR) * VEC;
    const bool vec_in = ((ld_in % VEC) == 0) && ((((uintptr_t)in) & 15) == 0);
    const bool vec_out = ((ld_out % VEC) == 0) && ((((uintptr_t)out) & 15) == 0);
    // load: tile[r][cc] = in[n0 + r][c0 + cc]
    for (int r = tr; r < TS; r += RPP) {
        const int nn = n0 + r, cc = c0 + tv;
        if (nn < n) {
            const T* src = in + (b * n + nn) * ld_in + cc;
            if (vec_in && cc + VEC <= c) {
                T tmp[VEC];
                *(u32x4_t*)tmp = *(const u32x4_t*)src;
#pragma unroll
                for (int e = 0; e < VEC; ++e) tile[r][tv + e] = tmp[e];
            } else {
                for (int e = 0; e < VEC; ++e) if (cc + e < c) tile[r][tv + e] = src[e];
            }
        }
    }
    __syncthreads();
    // store: out[c0 + r][n0 + cc] = tile[cc][r]
    for (int r = tr; r < TS; r += RPP) {
        const int cidx = c0 + r, nn = n0 + tv;
        if (cidx < c) {
            T* dst = out + (b * c + cidx) * ld_out + nn;
            if (vec_out && nn + VEC <= n) {
                T tmp[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) tmp[e] = tile[tv + e][r];
                *(u32x4_t*)dst = *(const u32x4_t*)tmp;
            } else {
                for (int e = 0; e < VEC; ++e) if (nn + e < n) dst[e] = tile[tv + e][r];
            }
        }
    }
}

extern "C" int cvar_transpose(const void* in, void* out, int dtype, int B, int n, int c, int64_t ld_in, int64_t ld_out, void* stream) {
    if (!in || !out || B <= 0 || n <= 0 || c <= 0 || ld_out < n) return CVAR_EINVAL;
    dim3 grid(cdiv(n, 64), cdiv(c, 64), B), block(256);
    if (dtype == CVAR_BF16) hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, block, 0, as_stream(stream), (const bf16_t*)in, (bf16_t*)out, n, c, (long)ld_in, (long)ld_out);
    else if (dtype == CVAR_F32) hipLaunchKernelGGL(transpose_kernel<float>, grid, block, 0, as_stream(stream), (const float*)in, (float*)out, n, c, (long)ld_in, (long)ld_out);
    else return CVAR_EUNSUPPORTED;
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// out[r] (+)= sum_j A[r][j], j < ncols  (rows contiguous: bias gradients read from the already transposed dY)
template <typename T>
__global__ __launch_bounds__(256) void rowsum_kernel(const T* __restrict__ A, long lda, float* __restrict__ out, int nrows, int ncols, int accumulate) {
    constexpr int VEC = 16 / sizeof(T);
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const T* ar = A + (long)row * lda;
    float acc = 0.f;
    const int nv = ncols / VEC;
    for (int i = lane; i < nv; i += 64) {
        T tmp[VEC];
        *(u32x4_t*)tmp = *(const u32x4_t*)(ar + (long)i * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc += Elem<T>::ld(tmp + e);
    }
    for (int j = nv * VEC + lane; j < ncols; j += 64) acc += Elem<T>::ld(ar + j);
    acc = wave_sum(acc);
    if (lane == 0) out[row] = accumulate ? out[row] + acc : acc;
}
extern "C" int cvar_rowsum(const void* A, int dtype, int64_t lda, float* out, int nrows, int ncols, int accumulate, void* stream) {
    if (!A || !out || nrows <= 0 || ncols <= 0) return CVAR_EINVAL;
    const int es = dtype == CVAR_BF16 ? 2 : 4;
    if ((lda * es) % 16 || ((uintptr_t)A & 15)) return CVAR_EUNSUPPORTED;
    dim3 grid(cdiv(nrows, 4)), block(256);
    if (dtype == CVAR_BF16) hipLaunchKernelGGL(rowsum_kernel<bf16_t>, grid, block, 0, as_stream(stream), (const bf16_t*)A, (long)lda, out, nrows, ncols, accumulate);
    else if (dtype == CVAR_F32) hipLaunchKernelGGL(rowsum_kernel<float>, grid, block, 0, as_stream(stream), (const float*)A, (long)lda, out, nrows, ncols, accumulate);
    else return CVAR_EUNSUPPORTED;
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

template <typename TO>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, TO* __restrict__ out, int C, int HW, int Cpad, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {            // i over B*HW*Cpad
        const int c = (int)(i % Cpad);
        const long bp = i / Cpad;
        const long p = bp % HW, b = bp / HW;
        Elem<TO>::st(out + i, c < C ? in[(b * C + c) * HW + p] : 0.f);
    }
}

extern "C" int cvar_nchw_to_nhwc(const float* in, void* out, int dtype, int B, int C, int HW, int Cpad, void* stream) {
    if (!in || !out || B <= 0 || C <= 0 || HW <= 0 || Cpad < C) return CVAR_EINVAL;
    const long total = (long)B * HW * Cpad;
    dim3 grid((unsigned)min((long)4096, (total + 255) / 256)), block(256);
    if (dtype == CVAR_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, block, 0, as_stream(stream), in, (bf16_t*)out, C, HW, Cpad, total);
    else if (dtype == CVAR_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, block, 0, as_stream(stream), in, (float*)out, C, HW, Cpad, total);
    else return CVAR_EUNSUPPORTED;
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

template <typename TI>
__global__ void nhwc_to_nchw_kernel(const TI* __restrict__ in, long ld_in, float* __restrict__ out, int C, int HW, long total,
                                    float lo, float hi, float mul, float add) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {            // i over B*C*HW (output order)
        const long p = i % HW;
        const long bc = i / HW;
        const int c = (int)(bc % C);
        const long b = bc / C;
        float v = Elem<TI>::ld(in + (b * HW + p) * ld_in + c);
        v = fminf(fmaxf(v, lo), hi);
        out[i] = v * mul + add;
    }
}

extern "C" int cvar_nhwc_to_nchw(const void* in, int dtype, int64_t ld_in, float* out, int B, int C, int HW,
                                 float lo, float hi, float mul, float add, void* stream) {
    if (!in || !out || B <= 0 || C <= 0 || HW <= 0) return CVAR_EINVAL;
    const long total = (long)B * C * HW;
    dim3 grid((unsigned)min((long)4096, (total + 255) / 256)), block(256);
    if (dtype == CVAR_BF16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid, block, 0, as_stream(stream), (const bf16_t*)in, (long)ld_in, out, C, HW, total, lo, hi, mul, add);
    else if (dtype == CVAR_F32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, block, 0, as_stream(stream), (const float*)in, (long)ld_in, out, C, HW, total, lo, hi, mul, add);
    else return CVAR_EUNSUPPORTED;
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

extern "C" int cvar_abi_version(void) { return 20; }
extern "C" const char* cvar_status_str(int status) {
    switch (status) {
        case CVAR_OK: return "ok";
        case CVAR_EINVAL: return "invalid argument";
        case CVAR_EUNSUPPORTED: return "unsupported shape or dtype";
        case CVAR_ELAUNCH: return "kernel launch failed";
        default: return "unknown status";
    }
}
