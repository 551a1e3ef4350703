// Multi-scale residual quantizer kernels (models/quant.py): one workgroup per 16x16x32 feature map, the whole
// map pyramid lives in LDS ([C][S][S] fp32 = 32 KiB per map), so a scale step never touches HBM except for the
// codebook rows (L2 resident, 512 KiB) and the final ids / f_hat / tokens.
//   bicubic up  : separable, operator table (S x pn) from the host (F.interpolate bicubic semantics)
//   phi         : 0.5*h + 0.5*(conv3x3(h)+b), weights repacked [ci][tap][co] and read as wave-uniform scalars
//   area down   : separable, operator table (pn x S) (F.interpolate area == adaptive average pooling)
//   nearest code: d = (|z|^2 + |e|^2) - 2 z.e in fp32, first minimum (torch.argmin), codebook split across lanes
#include "cvar_common.h"

constexpr int MS_C = 32;       // Cvae
constexpr int MS_S = 16;       // largest scale
constexpr int MS_MAP = MS_C * MS_S * MS_S;   // floats per map

// u <- bicubic_up(E[idx]) (or the plain gather when pn == S).  hs / u alias is handled by the caller passing
// distinct buffers: gather -> bufA ([C][pn][pn]); rows -> tmp ([C][S][pn]); cols -> bufA ([C][S][S]).
// NTH = threads of the workgroup (256, or 1024 for the generation-side kernel): the loops stride by it, every output keeps its own fma chain.
template <int NTH = 256>
__device__ void ms_gather_up(const int* __restrict__ idx, const float* __restrict__ E, const float* __restrict__ up,
                             float* bufA, float* tmp, int pn) {
    const int tid = threadIdx.x;
    const int n = pn * pn;
    for (int i = tid; i < n * MS_C; i += NTH) {
        const int t = i / MS_C, c = i % MS_C;
        bufA[c * n + t] = E[(long)idx[t] * MS_C + c];
    }
    __syncthreads();
    if (pn == MS_S) return;
    // rows: tmp[c][i][x] = sum_j up[i][j] * h[c][j][x]
    for (int o = tid; o < MS_C * MS_S * pn; o += NTH) {
        const int x = o % pn, i = (o / pn) % MS_S, c = o / (pn * MS_S);
        float a = 0.f;
        for (int j = 0; j < pn; ++j) a = fmaf(up[i * pn + j], bufA[c * n + j * pn + x], a);
        tmp[o] = a;
    }
    __syncthreads();
    // cols: u[c][i][i2] = sum_j up[i2][j] * tmp[c][i][j]
    for (int o = tid; o < MS_MAP; o += NTH) {
        const int i2 = o % MS_S, ci = o / MS_S;          // ci = c*S + i
        float a = 0.f;
        for (int j = 0; j < pn; ++j) a = fmaf(up[i2 * pn + j], tmp[ci * pn + j], a);
        bufA[o] = a;
    }
    __syncthreads();
}

// h = 0.5*u + 0.5*(conv3x3(u) + b);  fh += h;  (fr -= h if fr != nullptr).  One thread per pixel and block of 32 / (NTH / 256) output channels
// (NTH = 256: all 32; NTH = 1024: 8 - the 9 216-fma chain per pixel is the bulk of a generation-side call, 100 us with 256 threads).
template <int NTH = 256>
__device__ void ms_phi_accumulate(const float* u, const float* __restrict__ w /*[ci][9][co]*/, const float* __restrict__ bias,
                                  float* fh, float* fr) {
    constexpr int NCO = MS_C / (NTH / 256);
    const int pix = threadIdx.x & 255;           // S*S pixels
    const int co0 = (threadIdx.x >> 8) * NCO;
    const int y = pix / MS_S, x = pix % MS_S;
    float acc[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) acc[co] = bias[co0 + co];
    for (int ci = 0; ci < MS_C; ++ci) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            const float v = (yy >= 0 && yy < MS_S && xx >= 0 && xx < MS_S) ? u[ci * 256 + yy * MS_S + xx] : 0.f;
            const float* wr = w + (ci * 9 + tap) * MS_C + co0;
#pragma unroll
            for (int co = 0; co < NCO; ++co) acc[co] = fmaf(v, wr[co], acc[co]);
        }
    }
#pragma unroll
    for (int co = 0; co < NCO; ++co) {
        const float h = u[(co0 + co) * 256 + pix] * 0.5f + acc[co] * 0.5f;
        fh[(co0 + co) * 256 + pix] += h;
        if (fr) fr[(co0 + co) * 256 + pix] -= h;
    }
    __syncthreads();
}

// area-pool src ([C][S][S]) to pn x pn; result token-major: dst[t][c].  tmp: [C][pn][S].
template <int NTH = 256>
__device__ void ms_area_tokens(const float* src, const float* __restrict__ down /*[pn][S]*/, float* tmp, float* dst, int pn,
                               bool dst_global) {
    const int tid = threadIdx.x;
    if (pn == MS_S) {
        for (int o = tid; o < MS_MAP; o += NTH) { const int c = o % MS_C, t = o / MS_C; dst[o] = src[c * 256 + t]; }
        if (!dst_global) __syncthreads();
        return;
    }
    for (int o = tid; o < MS_C * pn * MS_S; o += NTH) {
        const int x = o % MS_S, i = (o / MS_S) % pn, c = o / (MS_S * pn);
        float a = 0.f;
        for (int yy = 0; yy < MS_S; ++yy) a = fmaf(down[i * MS_S + yy], src[c * 256 + yy * MS_S + x], a);
        tmp[o] = a;
    }
    __syncthreads();
    for (int o = tid; o < pn * pn * MS_C; o += NTH) {
        const int c = o % MS_C, t = o / MS_C;
        const int i = t / pn, j = t % pn;
        float a = 0.f;
        for (int xx = 0; xx < MS_S; ++xx) a = fmaf(down[j * MS_S + xx], tmp[(c * pn + i) * MS_S + xx], a);
        dst[o] = a;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
constexpr int MS_NI_THREADS = 1024;       // 16 waves: the phi conv of a map is split over 4 blocks of 8 output channels (4 x shorter chain per thread)
__global__ __launch_bounds__(MS_NI_THREADS) void ms_next_input_kernel(const int* __restrict__ idx, const float* __restrict__ E,
                                                           const float* __restrict__ phi_w, const float* __restrict__ phi_b,
                                                           const float* __restrict__ up, const float* __restrict__ down,
                                                           float* __restrict__ f_hat, float* __restrict__ tok_out,
                                                           int nmaps, int pn, int pn_next) {
    constexpr int NTH = MS_NI_THREADS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bufA = lds;                 // [C][S][S]
    float* fh = lds + MS_MAP;
    float* tmp = lds + 2 * MS_MAP;
    const int map = blockIdx.x % nmaps;
    const long b = blockIdx.x / nmaps;
    float* fg = f_hat + (b * nmaps + map) * (long)MS_MAP;
    for (int o = threadIdx.x; o < MS_MAP; o += NTH) fh[o] = fg[o];
    // the phi weights ([ci][tap][co], 36 KB) into LDS, coalesced, while the gather runs: read through wave-uniform scalar loads they were a chain of 288 s_load
    // round trips per thread (73 us per call at B = 1 with the conv itself at ~5 us of fma)
    float* wl = lds + 3 * MS_MAP;
    for (int o = threadIdx.x * 4; o < MS_C * 9 * MS_C; o += NTH * 4) *(f32x4_t*)(wl + o) = *(const f32x4_t*)(phi_w + o);
    ms_gather_up<NTH>(idx + (b * nmaps + map) * (long)(pn * pn), E, up, bufA, tmp, pn);
    ms_phi_accumulate<NTH>(bufA, wl, phi_b, fh, nullptr);
    for (int o = threadIdx.x; o < MS_MAP; o += NTH) fg[o] = fh[o];
    if (tok_out) {
        float* dst = tok_out + ((b * nmaps + map) * (long)(pn_next * pn_next)) * MS_C;
        ms_area_tokens<NTH>(fh, down, tmp, dst, pn_next, true);
    }
}

extern "C" int cvar_ms_next_input(const int32_t* idx, const float* codebook, const float* phi_w, const float* phi_b,
                                  const float* up_mat, const float* down_mat, float* f_hat, float* tok_out,
                                  int nb, int nmaps, int pn, int pn_next, int S, int Cvae, void* stream) {
    if (!idx || !codebook || !phi_w || !phi_b || !f_hat || nb <= 0 || nmaps <= 0 || pn <= 0 || pn > S) return CVAR_EINVAL;
    if (S != MS_S || Cvae != MS_C) return CVAR_EUNSUPPORTED;
    if (pn != S && !up_mat) return CVAR_EINVAL;
    if (tok_out && (pn_next <= 0 || pn_next > S || (pn_next != S && !down_mat))) return CVAR_EINVAL;
    const size_t lds = (3 * MS_MAP + MS_C * 9 * MS_C) * sizeof(float);        // three maps + the phi weights
    (void)hipFuncSetAttribute((const void*)ms_next_input_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(ms_next_input_kernel, dim3((unsigned)((long)nb * nmaps)), dim3(MS_NI_THREADS), lds, as_stream(stream), idx, codebook, phi_w, phi_b,
                       up_mat, down_mat, f_hat, tok_out, nmaps, pn, pn_next);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// ------------------------------------------------------------------------------------------------
struct MsEncodeParams {
    const float* f; const float* E; int V;
    const float* phi_w; const float* phi_b;
    const float* up; const float* down;
    int* idx_out; float* f_hat_out; float* margin_out;
    int nscale;
    int pn[16], phi_map[16], up_off[16], down_off[16], idx_off[16];
    int Ltot;
};

// NTH = 256: one wave per SIMD (the pyramid fills the LDS), both search forms (matrix-pipe search; sequential search when the margins are asked for).
// NTH = 512: 8 waves, two per SIMD - one wave's compare / select / code-row loads run beside the other's MFMAs -, the phi conv split over two blocks of output
// channels; matrix-pipe search only (cvar_ms_encode picks the instance).  Every distance / conv output is the same fma chain in both, and the merges keep the
// first minimum: identical ids.  (Round 4's lane-owns-a-code VALU search - 2.62 ms per call against 1.93 ms - is gone: experiments/README.md.)
template <int NTH>
__global__ __launch_bounds__(NTH) void ms_encode_kernel(const MsEncodeParams p) {
    constexpr int NW = NTH / 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bufA = lds;
    float* fh = lds + MS_MAP;
    float* tmp = lds + 2 * MS_MAP;
    float* fr = lds + 3 * MS_MAP;
    __shared__ float red_d[NTH == 256 ? 256 : 1], red_d2[NTH == 256 ? 256 : 1];
    __shared__ int red_i[NTH == 256 ? 256 : 1];
    __shared__ int sidx[256];
    __shared__ float zzs[256];
    // per-wave best (distance, index) of a token: [NW][256] each, parked in `tmp` - free between the area pooling and the gather (NW <= 16: 32 KB = one map)
    float (*wave_bd)[256] = (float (*)[256])tmp;
    int (*wave_bi)[256] = (int (*)[256])(tmp + NW * 256);
    const int tid = threadIdx.x;
    const long b = blockIdx.x;
    for (int o = tid; o < MS_MAP; o += NTH) { fr[o] = p.f[b * MS_MAP + o]; fh[o] = 0.f; }
    __syncthreads();
    for (int si = 0; si < p.nscale; ++si) {
        const int pn = p.pn[si], n = pn * pn;
        // z[t][c] = area(f_rest) -> bufA (token-major)
        ms_area_tokens<NTH>(fr, p.down + p.down_off[si], tmp, bufA, pn, false);
        if (!p.margin_out && (p.V % (32 * NW)) == 0) {
            // ---- nearest code on the matrix pipe (round 5).  dot[t][v] = sum_c z[t][c] E[v][c] is a GEMM: v_mfma_f32_32x32x2_f32 is exact fp32 and bitwise an
            // fmaf chain over k (MI355X_MICROARCH.md; the fp32 parity mode of gemm.hip rests on the same property), so with channels 2 s, 2 s + 1 in step s
            // every distance is the SAME ascending-c chain the sequential search computes - at the matrix pipe's 64 flop per clock and SIMD, without a
            // broadcast LDS read per fma.  Wave w owns codes [w V/4, (w+1) V/4) as before; per block of 32 tokens it walks its codes 32 at a time:
            // lane j (both half-waves) holds code j's row (B operand: channel 2 s + hi of code j), the accumulator column j holds the block's 32 dots;
            // a lane keeps the first minimum of ITS column per token row (codes come in increasing order), then one lexicographic (distance, index)
            // reduction over the 32 columns per row and the merge of the four waves in code order: exactly the first-minimum index of the sequential search.
            const int lane = tid & 63, w = tid >> 6, col = lane & 31, hi = lane >> 5;
            if (tid < n) {
                float zz = 0.f;
#pragma unroll
                for (int c = 0; c < MS_C; ++c) zz = fmaf(bufA[tid * MS_C + c], bufA[tid * MS_C + c], zz);
                zzs[tid] = zz;
            }
            __syncthreads();
            const int per_wave = p.V / NW, cblocks = per_wave / 32;
            for (int t0 = 0; t0 < n; t0 += 32) {
                // A operand of step s: z[t0 + col][2 s + hi] (tail tokens repeat the last one: results ignored); the row's |z|^2 per accumulator register
                float za[MS_C / 2], zr[16];
                {
                    const float* zt = bufA + min(t0 + col, n - 1) * MS_C + hi;
#pragma unroll
                    for (int s_ = 0; s_ < MS_C / 2; ++s_) za[s_] = zt[2 * s_];
#pragma unroll
                    for (int r = 0; r < 16; ++r) zr[r] = zzs[min(t0 + (r & 3) + 8 * (r >> 2) + 4 * hi, n - 1)];
                }
                float bd[16]; int bi[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) { bd[r] = INFINITY; bi[r] = 0; }
                f32x4_t e4[MS_C / 4], nx[MS_C / 4];
                {
                    const float* er = p.E + (long)(w * per_wave + col) * MS_C;
#pragma unroll
                    for (int q = 0; q < MS_C / 4; ++q) nx[q] = *(const f32x4_t*)(er + 4 * q);
                }
                for (int cb = 0; cb < cblocks; ++cb) {
                    const int v = w * per_wave + cb * 32 + col;
#pragma unroll
                    for (int q = 0; q < MS_C / 4; ++q) e4[q] = nx[q];
                    if (cb + 1 < cblocks) {
                        const float* er = p.E + (long)(v + 32) * MS_C;
#pragma unroll
                        for (int q = 0; q < MS_C / 4; ++q) nx[q] = *(const f32x4_t*)(er + 4 * q);
                    }
                    f32x16_t acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                    for (int s_ = 0; s_ < MS_C / 2; ++s_) {
                        const float eb = hi ? e4[s_ >> 1][2 * (s_ & 1) + 1] : e4[s_ >> 1][2 * (s_ & 1)];       // E[v][2 s + hi]
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(za[s_], eb, acc, 0, 0, 0);
                    }
                    float ee = 0.f;
#pragma unroll
                    for (int q = 0; q < MS_C / 4; ++q) {
                        ee = fmaf(e4[q][0], e4[q][0], ee); ee = fmaf(e4[q][1], e4[q][1], ee);
                        ee = fmaf(e4[q][2], e4[q][2], ee); ee = fmaf(e4[q][3], e4[q][3], ee);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float d = __fadd_rn(__fadd_rn(zr[r], ee), __fmul_rn(-2.0f, acc[r]));
                        if (d < bd[r]) { bd[r] = d; bi[r] = v; }                  // a column's codes come in increasing order: first minimum
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float d = bd[r]; int i = bi[r];
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) {                            // over the 32 columns of this half-wave
                        const float d2 = __shfl_xor(d, o, 64);
                        const int i2 = __shfl_xor(i, o, 64);
                        if (d2 < d || (d2 == d && i2 < i)) { d = d2; i = i2; }
                    }
                    const int t = t0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (col == 0 && t < n) { wave_bd[w][t] = d; wave_bi[w][t] = i; }
                }
            }
            __syncthreads();
            if (tid < n) {
                float bdm = wave_bd[0][tid]; int bim = wave_bi[0][tid];
#pragma unroll
                for (int q = 1; q < NW; ++q)                                     // waves hold increasing code ranges: strict '<'
                    if (wave_bd[q][tid] < bdm) { bdm = wave_bd[q][tid]; bim = wave_bi[q][tid]; }
                sidx[tid] = bim;
                p.idx_out[b * p.Ltot + p.idx_off[si] + tid] = bim;
            }
            __syncthreads();
        } else if constexpr (NTH == 256) {
        // nearest code: thread = (token, part)
        const int P = n >= 256 ? 1 : 256 / n;
        const int t = tid / P, part = tid % P;
        float bd = INFINITY, bd2 = INFINITY;
        int bi = 0;
        if (t < n) {
            float z[MS_C], zz = 0.f;
#pragma unroll
            for (int c = 0; c < MS_C; ++c) { z[c] = bufA[t * MS_C + c]; }
#pragma unroll
            for (int c = 0; c < MS_C; ++c) zz = fmaf(z[c], z[c], zz);
            const int lo = (int)((long)part * p.V / P), hi = (int)((long)(part + 1) * p.V / P);
            for (int v = lo; v < hi; ++v) {
                const float* e = p.E + (long)v * MS_C;
                float dot = 0.f, ee = 0.f;
#pragma unroll
                for (int c = 0; c < MS_C; c += 4) {
                    const f32x4_t ev = *(const f32x4_t*)(e + c);
                    dot = fmaf(z[c], ev[0], dot); dot = fmaf(z[c + 1], ev[1], dot);
                    dot = fmaf(z[c + 2], ev[2], dot); dot = fmaf(z[c + 3], ev[3], dot);
                    ee = fmaf(ev[0], ev[0], ee); ee = fmaf(ev[1], ev[1], ee);
                    ee = fmaf(ev[2], ev[2], ee); ee = fmaf(ev[3], ev[3], ee);
                }
                const float d = __fadd_rn(__fadd_rn(zz, ee), __fmul_rn(-2.0f, dot));
                if (d < bd) { bd2 = bd; bd = d; bi = v; }
                else if (d < bd2) bd2 = d;
            }
        }
        red_d[tid] = bd; red_d2[tid] = bd2; red_i[tid] = bi;
        __syncthreads();
        if (t < n && part == 0) {
            for (int q = 1; q < P; ++q) {        // parts are in increasing code order -> strict '<' keeps the first minimum
                const float d = red_d[tid + q], d2 = red_d2[tid + q];
                if (d < bd) { bd2 = fminf(bd, d2); bd = d; bi = red_i[tid + q]; }
                else bd2 = fminf(bd2, d);
            }
            sidx[t] = bi;
            p.idx_out[b * p.Ltot + p.idx_off[si] + t] = bi;
            if (p.margin_out) p.margin_out[b * p.Ltot + p.idx_off[si] + t] = bd2 - bd;
        }
        __syncthreads();
        }
        // stages with more than 256 tokens do not occur (S*S == 256)
        ms_gather_up<NTH>(sidx, p.E, p.up + p.up_off[si], bufA, tmp, pn);
        const int k = p.phi_map[si];
        ms_phi_accumulate<NTH>(bufA, p.phi_w + (long)k * MS_C * 9 * MS_C, p.phi_b + k * MS_C, fh, fr);
    }
    if (p.f_hat_out)
        for (int o = tid; o < MS_MAP; o += NTH) p.f_hat_out[b * MS_MAP + o] = fh[o];
}

extern "C" int cvar_ms_encode(const float* f, const float* codebook, int V, const float* phi_w, const float* phi_b,
                              const int* phi_map_host, const int* patch_nums_host, int nscale, const float* up_mats,
                              const float* down_mats, int32_t* idx_out, float* f_hat_out, float* margin_out,
                              int B, int S, int Cvae, void* stream) {
    if (!f || !codebook || !phi_w || !phi_b || !phi_map_host || !patch_nums_host || !up_mats || !down_mats || !idx_out) return CVAR_EINVAL;
    if (B <= 0 || V <= 1 || nscale <= 0 || nscale > 16) return CVAR_EINVAL;
    if (S != MS_S || Cvae != MS_C || patch_nums_host[nscale - 1] != S) return CVAR_EUNSUPPORTED;
    MsEncodeParams p;
    p.f = f; p.E = codebook; p.V = V; p.phi_w = phi_w; p.phi_b = phi_b; p.up = up_mats; p.down = down_mats;
    p.idx_out = idx_out; p.f_hat_out = f_hat_out; p.margin_out = margin_out; p.nscale = nscale;
    int uo = 0, io = 0;
    for (int i = 0; i < 16; ++i) {
        const int pn = i < nscale ? patch_nums_host[i] : 0;
        if (i < nscale && (pn <= 0 || pn > S)) return CVAR_EINVAL;
        p.pn[i] = pn; p.phi_map[i] = i < nscale ? phi_map_host[i] : 0;
        p.up_off[i] = uo; p.down_off[i] = uo; p.idx_off[i] = io;
        uo += S * pn; io += pn * pn;
    }
    p.Ltot = io;
    const size_t lds = 4 * MS_MAP * sizeof(float);
#ifndef MS_ENCODE_W8
#define MS_ENCODE_W8 1
#endif
    if (MS_ENCODE_W8 && !margin_out && V % 256 == 0) {
        // eight waves (two per SIMD): one wave's compare / select / code-row loads run beside the other's MFMAs; 512 codes per wave
        (void)hipFuncSetAttribute((const void*)ms_encode_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(ms_encode_kernel<512>, dim3(B), dim3(512), lds, as_stream(stream), p);
        CVAR_CHECK_LAUNCH();
        return CVAR_OK;
    }
    (void)hipFuncSetAttribute((const void*)ms_encode_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(ms_encode_kernel<256>, dim3(B), dim3(256), lds, as_stream(stream), p);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// ------------------------------------------------------------------------------------------------
// Helpers of the low-resolution reconstruction path embed_to_fhat(all_to_max_scale=False) (quant.py:171-180; idxBl_to_img(same_shape=
// False), vqvae.py:97-104: upstream's visualisation of the pyramid at each scale's own resolution).  Tiny tensors (<= 16x16x32).
//   embed_rows   : out[n][:] = codebook[idx[n]][:]                                       (nn.Embedding lookup, vqvae.py:103)
//   resample_sep : out[b][y][x][c] = sum_i sum_j wy[y][i] * wx[x][j] * in[b][i][j][c]     (F.interpolate as two dense matrices, NHWC fp32)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_rows_kernel(const int32_t* __restrict__ idx, const float* __restrict__ E, float* __restrict__ out,
                                                        long n, int C, int V) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n * C; i += (long)gridDim.x * 256) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        const int v = min(max(idx[r], 0), V - 1);
        out[i] = E[(long)v * C + c];
    }
}

extern "C" int cvar_embed_rows(const int32_t* idx, const float* codebook, int V, float* out, int64_t n, int C, void* stream) {
    if (!idx || !codebook || !out || n <= 0 || C <= 0 || V <= 0) return CVAR_EINVAL;
    hipLaunchKernelGGL(embed_rows_kernel, dim3((unsigned)min((int64_t)1024, (n * C + 255) / 256)), dim3(256), 0, as_stream(stream), idx, codebook, out,
                       (long)n, C, V);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

__global__ __launch_bounds__(256) void resample_sep_kernel(const float* __restrict__ in, const float* __restrict__ wy, const float* __restrict__ wx,
                                                          float* __restrict__ out, int B, int h, int w, int H, int W, int C) {
    const long total = (long)B * H * W * C;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const int c = (int)(o % C);
        const int x = (int)((o / C) % W);
        const int y = (int)((o / ((long)C * W)) % H);
        const long b = o / ((long)C * W * H);
        float acc = 0.f;
        for (int i = 0; i < h; ++i) {
            const float a = wy[y * h + i];
            if (a == 0.f) continue;
            float row = 0.f;
            for (int j = 0; j < w; ++j) {
                const float bq = wx[x * w + j];
                if (bq != 0.f) row += bq * in[((b * h + i) * w + j) * C + c];
            }
            acc += a * row;
        }
        out[o] = acc;
    }
}

extern "C" int cvar_resample_sep(const float* in, const float* wy, const float* wx, float* out, int B, int h, int w, int H, int W, int C, void* stream) {
    if (!in || !wy || !wx || !out || B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || C <= 0) return CVAR_EINVAL;
    const long total = (long)B * H * W * C;
    hipLaunchKernelGGL(resample_sep_kernel, dim3((unsigned)min((long)2048, (total + 255) / 256)), dim3(256), 0, as_stream(stream), in, wy, wx, out, B, h, w,
                       H, W, C);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}
