// Small-M GEMMs of the early scales and of small batches (M = R * l <= a few hundred rows): weight-streaming kernel + row-finishing split-K reduction.
//
// Why a second GEMM kernel.  With M <= 144 rows the LDS-tiled kernels of gemm.hip have one or two row tiles; they fill the chip only by splitting K
// (128x128 tiles x up to 16 slices) and pay a second launch that sums the slices: 9.5 + 5.2 us per GEMM at M <= 64 and 13-16 + 6.5-8 us at M = 100 ... 256
// in a B = 1 generation (profiles/r04_small_batch.txt), against 1-4 us of weight streaming.  Here the problem is cut along N instead:
//   * a workgroup owns 16 NT output columns x 16 MT rows over the WHOLE K range: no partial sums leave the workgroup, the epilogue runs in the same launch;
//   * its 8 waves interleave the 32-deep k-steps (wave w takes steps w, w + 8, ... of a per-workgroup rotation of K), so the workgroup streams 512 contiguous
//     bytes of every weight row per round and different workgroups read different k-blocks of the shared activations at any moment;
//   * weights go global -> registers (buffer_load_dwordx4, 16 B per lane = one MFMA fragment; rows past N are out-of-range offsets = zeros): every weight
//     byte is used by exactly one wave, LDS staging would only add a hop;
//   * the activations (16 MT rows x K, re-read by every workgroup out of L2) are the bulk of a workgroup's bytes.  Fetched as fragments - 16 rows x 16 B per
//     quarter wave, 64 cache lines per instruction - the CU's texture path delivers a quarter of its rate (first version: 8.5 us at M = 4 -> 24.6 us at
//     M = 144 for the qkv call, profiles/r04_small_batch.txt).  They are therefore fetched row-contiguously (half a wave per 512-byte row piece), parked in
//     LDS (two 256-k blocks, XOR-swizzled 16-byte slots, two blocks ahead in registers) and read back as fragments with conflict-free ds_read_b128;
//   * v_mfma_f32_16x16x32_bf16 with the weight fragment as the row operand: a lane ends up with 4 consecutive output columns of one row (vector stores);
//   * the 8 partial accumulators meet in LDS (fixed wave order -> bit-reproducible) and each thread finishes one quad through gemm_epilogue_quad.
// Long-K calls (fc2: K = 4C) would make every workgroup read M x K activations; they keep a K split (blockIdx.z, fp32 partial tiles in the caller's workspace)
// and are finished ROW-WISE by cvar_splitk_rowfin_kernel, which also runs the adaLN of the next op on the finished row (cvar_gemm_desc.ln_out): slices -> sum ->
// bias / gate / residual -> x, LN(x) * (1 + scale) + shift -> bf16 in one launch instead of epilogue kernel + cvar_ln_modulate.
// Math: basic_var.py:43-51,92,119,207-209 (same functions as gemm.hip / ops.hip; accumulation order differs from the tile kernels, fp32 sums).
#include "gemm_params.h"

// cache policy of the weight stream (aux of buffer_load): 0 default, 2 = nt (each weight byte is read by ONE workgroup, once)
#ifndef CVAR_SKINNY_T8_TWO
#define CVAR_SKINNY_T8_TWO 1     // the 64x32 tile compiled for two workgroups per CU (<= 128 registers)
#endif
#ifndef CVAR_SKINNY_NT2_SLICED
#define CVAR_SKINNY_NT2_SLICED 1
#endif
#ifndef CVAR_SKINNY_SLICED_GY
#define CVAR_SKINNY_SLICED_GY 1      // K slices with two row groups (fc2 at M = 100 ... 128): 22.2 vs 22.0 us for the tiles - neutral, off
#endif
#ifndef CVAR_SKINNY_PROJ_SLICED
#define CVAR_SKINNY_PROJ_SLICED 1
#endif
#ifndef CVAR_SKINNY_MOST_SLICES
#define CVAR_SKINNY_MOST_SLICES 1
#endif
#ifndef CVAR_SKINNY_BIG_GY
#define CVAR_SKINNY_BIG_GY 4
#endif
#ifndef CVAR_SKINNY_NT2
#define CVAR_SKINNY_NT2 1
#endif
#ifndef CVAR_SKINNY_W_AUX
#define CVAR_SKINNY_W_AUX 0
#endif

typedef __attribute__((ext_vector_type(4))) int v4i_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bfv8_t;

template <int MT, int NT, bool PARTIAL>
__global__ __launch_bounds__(512, (MT * NT == 8 && CVAR_SKINNY_T8_TWO) ? 4 : 2) void cvar_gemm_skinny_kernel(const GemmParams p) {
    constexpr int T = MT * NT, RM = 16 * MT;
    static_assert(T <= 8, "one output quad per thread");
    constexpr int STAGE = RM * 512;                              // one activation block: RM rows x 256 k (512 B per row)
    constexpr int RED = 8 * T * 64 * 16;
    constexpr int LDS_BYTES = 2 * STAGE > RED ? 2 * STAGE : RED;
    constexpr int WPF = (MT * NT == 8 && CVAR_SKINNY_T8_TWO) ? 4 : 8;      // weight fragments in flight per wave, in blocks (K = 1536: 6 blocks); 64x32 tiles keep 4 to fit 128 registers
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * (16 * NT), m0 = blockIdx.y * RM;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((long)p.M * p.lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)((long)p.N * p.ldw * 2), 0x00020000);

    // k range of this workgroup in steps of 32 (a K slice when the call is split) and in blocks of 8 steps; each workgroup starts its walk over the blocks at
    // its own block (rot): every workgroup reads the SAME activation rows, and with one common order a k-block of all rows sits on a few L2 channels
    const int nks_all = p.K >> 5;
    const int ks_lo = p.split_tiles > 0 ? (int)blockIdx.z * p.split_tiles : 0;
    const int ks_hi = p.split_tiles > 0 ? min(nks_all, ks_lo + p.split_tiles) : nks_all;
    const int S = ks_hi - ks_lo, nb = (S + 7) >> 3;
    const int rot = (int)((blockIdx.x * 11u + blockIdx.y * 5u) % (unsigned)max(nb, 1));
    auto blk = [&](int b) { const int bb = b + rot; return bb >= nb ? bb - nb : bb; };

    // staging: thread -> (row, 16-byte chunk) of the block, row-contiguous (half a wave covers a row's 512 B); LDS slot XOR-swizzled by the row
    unsigned a_off[MT];
    int lds_w[MT];
    const int c = tid & 31;
#pragma unroll
    for (int e = 0; e < MT; ++e) {
        const int r = (tid >> 5) + 16 * e, m = m0 + r;
        a_off[e] = m < p.M ? (unsigned)(((long)m * p.lda + c * 8) * 2) : 0x80000000u;
        lds_w[e] = r * 512 + ((c ^ (r & 15)) << 4);
    }
    bf16x8_t ar[2][MT];
    auto issueA = [&](int slot, int b) {
        const int k0 = (ks_lo + 8 * blk(b)) * 32, rem = ks_hi * 32 - k0;          // elements of the slice from this block on
#pragma unroll
        for (int e = 0; e < MT; ++e)
            ar[slot][e] = __builtin_bit_cast(bf16x8_t, (v4i_t)__builtin_amdgcn_raw_buffer_load_b128(a_rsrc, c * 8 < rem ? a_off[e] : 0x80000000u, (unsigned)k0 * 2u, 0));
    };
    unsigned offW[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + 16 * j + l15;
        offW[j] = n < p.N ? (unsigned)(((long)n * p.ldw + kq * 8) * 2) : 0x80000000u;
    }
    bf16x8_t wf[WPF][NT];
    auto issueW = [&](int slot, int b) {
        const int st = 8 * blk(b) + wave;                                          // this wave's step of the block
#pragma unroll
        for (int j = 0; j < NT; ++j)
            wf[slot][j] = __builtin_bit_cast(bf16x8_t, (v4i_t)__builtin_amdgcn_raw_buffer_load_b128(w_rsrc, st < S ? offW[j] : 0x80000000u, (unsigned)(ks_lo + st) * 64u, CVAR_SKINNY_W_AUX));
    };
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment of MFMA row block i: row 16 i + l15, chunk 4 wave + kq of the block (conflict-free with the swizzle: see the ds_read_b128 lane groups)
    const int fr = l15 * 512 + (((4 * wave + kq) ^ l15) << 4);
    if (nb > 0) {
        issueA(0, 0);
        if (nb > 1) issueA(1, 1);
#pragma unroll
        for (int u = 0; u < WPF; ++u)
            if (u < nb) issueW(u, u);
#pragma unroll
        for (int e = 0; e < MT; ++e) *(bf16x8_t*)(smem + lds_w[e]) = ar[0][e];
        if (nb > 2) issueA(0, 2);
    }
    __syncthreads();
    for (int base = 0; base < nb; base += WPF) {
#pragma unroll
        for (int u = 0; u < WPF; ++u) {
            const int b = base + u;
            if (b < nb) {
                bf16x8_t af[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) af[i] = *(const bf16x8_t*)(smem + (u & 1) * STAGE + fr + i * 8192);
                if (b + 1 < nb) {
#pragma unroll
                    for (int e = 0; e < MT; ++e) *(bf16x8_t*)(smem + ((u + 1) & 1) * STAGE + lds_w[e]) = ar[(u + 1) & 1][e];
                }
                if (b + 3 < nb) issueA((u + 1) & 1, b + 3);
                if (8 * blk(b) + wave < S) {
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int i = 0; i < MT; ++i)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfv8_t, wf[u][j]), __builtin_bit_cast(bfv8_t, af[i]), acc[i][j], 0, 0, 0);
                }
                if (b + WPF < nb) issueW(u, b + WPF);
                __syncthreads();
            }
        }
    }
    // the eight waves' partial tiles -> LDS (over the staging buffers: everybody is past the last block's barrier) -> one quad per thread, summed in wave order
    f32x4_t* sred = (f32x4_t*)smem;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) sred[(wave * T + i * NT + j) * 64 + lane] = acc[i][j];
    __syncthreads();
    if (tid < T * 64) {
        f32x4_t v = sred[tid];
#pragma unroll
        for (int w = 1; w < 8; ++w) {
            const f32x4_t o = sred[w * T * 64 + tid];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += o[e];
        }
        const int tile = tid >> 6, ln = tid & 63;
        const int m = m0 + 16 * (tile / NT) + (ln & 15), n = n0 + 16 * (tile % NT) + 4 * (ln >> 4);
        if (m < p.M && n < p.N) {
            if constexpr (PARTIAL) *(f32x4_t*)((float*)p.C + (long)blockIdx.z * p.split_stride + (long)m * p.N + n) = v;      // K slice: raw fp32 sums
            else gemm_epilogue_quad(p, m, n, v);
        }
    }
}

template <int MT, int NT>
static int skinny_launch_cfg(const GemmParams& p, int slices, hipStream_t st) {
    dim3 grid((unsigned)((p.N + 16 * NT - 1) / (16 * NT)), (unsigned)((p.M + 16 * MT - 1) / (16 * MT)), (unsigned)slices), block(512);
    if (p.split_tiles > 0) hipLaunchKernelGGL((cvar_gemm_skinny_kernel<MT, NT, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((cvar_gemm_skinny_kernel<MT, NT, false>), grid, block, 0, st, p);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// plan: rows per workgroup 16 mt, columns 16 nt, K slices (1 = none).  Returns 0 when the call is not one for this kernel.  Every rule below is an A/B of
// tools/skinny_bench.py on the d24 calls (profiles/r04_small_batch.txt, sections 6 and 11: per call incl. the adaLN where requested, streaming / LDS-tiled):
//   - one row group (M <= 64): always (qkv 9.0-12.5 / 11.6-16.7 us, fc1 9.4-12.9 / 13.1-18.8, proj 8.6-10.4 / 12.0-15.1, fc2 14.0-17.5 / 18.9-21.6);
//   - more row groups: the small square call (proj, N K <= 4 M) up to M = 256; big calls (qkv, fc1) only as 64x32 tiles (128 registers: two workgroups per CU)
//     with at most 432 workgroups - M <= 128, qkv M <= 192; beyond that the three-stage tiles win;
//   - 32 columns per workgroup (halves the activation re-reads, the bulk of a 64-row workgroup's bytes) where the grid keeps >= 128 workgroups;
//   - K slices: the fewest that keep a workgroup's bytes under 320 KB (fc2: 4), the most below 64-row groups when a row-finishing launch follows anyway
//     (fc2 at M <= 32: 8), and two for a call with few column groups whose rows get finished anyway (proj: slices + row finish instead of GEMM + adaLN launch).
int cvar_gemm_skinny_plan(int M, int N, int K, long lda, long ldw, int want_rowfin, int have_ws, int* mt_, int* nt_, int* slices_) {
    if (M <= 0 || M > 256 || (K & 31) || (N & 15) || (lda & 7) || (ldw & 7)) return 0;
    if ((long)M * lda * 2 >= (1L << 31) || (long)N * ldw * 2 >= (1L << 31)) return 0;
    const int mt = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
    const int gy = (M + 16 * mt - 1) / (16 * mt);
    // big calls (qkv, fc1) with more than one row group: only as 64x32 tiles (two workgroups per CU since the tile fits 128 registers) and up to CVAR_SKINNY_BIG_GY groups
    const bool big = (long)N * K > (4L << 20);
    // (measured, us per call, 64x32 streaming / three-stage tiles: M = 100 qkv 15.9 / 18.4, fc1 16.8 / 20.8; M = 144 qkv 16.7 / 18.6, fc1 22.9 / 20.9; M = 200-256 lose:
    //  at most 432 workgroups - two per CU resident, a little tail)
    if (gy > 1 && big && (gy > CVAR_SKINNY_BIG_GY || (N & 31) || mt != 4 || (long)(N / 32) * gy > 432)) return 0;
    const int nks = K >> 5;
    // a call with few column groups whose rows get finished anyway (ln_out): two K slices + the row-finishing launch instead of one slice + cvar_ln_modulate
    const int sl0 = (CVAR_SKINNY_PROJ_SLICED && want_rowfin && have_ws && gy == 1 && N / 16 < 128) ? 2 : 1;
    for (int it = 0; it < 4; ++it) {
        // sliced calls that end in the row-finishing launch anyway: most slices first (shorter K walks per workgroup), otherwise fewest
        // (fc2, 8 / 4 slices: M = 4 14.0 / 16.4 us, M = 16 15.3 / 16.5, M = 36 17.7 / 16.9, M = 64 18.9 / 17.5: only below 64-row groups)
        const int sl = (CVAR_SKINNY_MOST_SLICES && want_rowfin && have_ws && gy == 1 && mt < 4) ? (8 >> it) : (1 << it);
        if (sl < sl0) continue;
        if (sl > 1 && (!have_ws || gy > CVAR_SKINNY_SLICED_GY || nks % sl || (nks / sl) % 8 || nks / sl < 16)) continue;          // a K slice is whole 256-k blocks, at least 512 deep
        if ((double)(16 * mt + 16) * (K / sl) * 2.0 > 320.0 * 1024) continue;
        // 32 columns per workgroup halve the activation re-reads - worth it where the activations are the bulk (64-row groups) and the halved grid still
        // has >= 128 workgroups (qkv, fc1); CVAR_SKINNY_NT2 = 0 keeps 16 columns everywhere
        const int nt = (CVAR_SKINNY_NT2 && mt == 4 && (gy == 1 || big) && (sl == 1 || CVAR_SKINNY_NT2_SLICED) && (N & 31) == 0 && (long)(N / 32) * sl >= 128) ? 2 : 1;
        if (gy > 1 && big && nt != 2) continue;
        if ((double)(16 * mt + 16 * nt) * (K / sl) * 2.0 > 320.0 * 1024) continue;
        *mt_ = mt; *nt_ = nt; *slices_ = sl;
        return 1;
    }
    return 0;
}

int cvar_gemm_skinny_launch(const GemmParams& p, int mt, int nt, int slices, hipStream_t st) {
    if (nt == 1) {
        if (mt == 1) return skinny_launch_cfg<1, 1>(p, slices, st);
        if (mt == 2) return skinny_launch_cfg<2, 1>(p, slices, st);
        return skinny_launch_cfg<4, 1>(p, slices, st);
    }
    if (mt == 1) return skinny_launch_cfg<1, 2>(p, slices, st);
    if (mt == 2) return skinny_launch_cfg<2, 2>(p, slices, st);
    return skinny_launch_cfg<4, 2>(p, slices, st);
}

// ------------------------------------------------------------------------------------------------
// Row-finishing split-K reduction: one wave per output row.  part = nsplit fp32 slices [M][N]; the row is summed in slice order (the order of
// cvar_splitk_epilogue_kernel -> same bits), finished with bias / gate / fp32 residual exactly as gemm_epilogue_quad does, stored, and - the point of
// working row-wise - normalised and modulated for the next op while it is still in registers, with cvar_ln_modulate's lane <-> column map and
// reduction order (ops.hip: bit-identical to running that kernel on the stored row).
// ------------------------------------------------------------------------------------------------
struct RowFinLn { void* out; const float* scale; const float* shift; long ld; int rows_per; float eps; };

template <typename TO, int NV, bool TAIL>
__global__ __launch_bounds__(256) void cvar_splitk_rowfin_kernel(const float* __restrict__ part, int nsplit, const GemmParams p, const RowFinLn ln) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const int N = p.N;
    const long sstride = (long)p.M * N;
    const float* q0 = part + (long)row * N;
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4_t v[NV];
    bool ok[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        ok[i] = !(TAIL && i == NV - 1) || c < N;
        v[i] = ok[i] ? *(const f32x4_t*)(q0 + c) : zero4;
    }
    // epilogue operands: issued before the slice sums so that their latency hides behind them
    const int g = fast_div(row, p.gate_magic, p.gate_shift);
    const float* grow = p.gate + (long)g * p.ldg;
    const float* rrow = (const float*)p.residual + (long)row * p.ldr;
    f32x4_t bq[NV], gq[NV], rq[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        bq[i] = (ok[i] && p.bias) ? *(const f32x4_t*)(p.bias + c) : zero4;
        gq[i] = ok[i] ? *(const f32x4_t*)(grow + c) : zero4;
        rq[i] = ok[i] ? *(const f32x4_t*)(rrow + c) : zero4;
    }
    // slices 1 .. nsplit - 1 in batches of four per vector (batches of eight - all slices of a d24 row in flight at once, 256 registers - measured ~1 us SLOWER per call)
    constexpr int SB = 4;
    for (int s0 = 1; s0 < nsplit; s0 += SB) {
        f32x4_t w[SB][NV];
#pragma unroll
        for (int u = 0; u < SB; ++u)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                w[u][i] = (s0 + u < nsplit && ok[i]) ? *(const f32x4_t*)(q0 + (long)(s0 + u) * sstride + c) : zero4;
            }
#pragma unroll
        for (int u = 0; u < SB; ++u)
            if (s0 + u < nsplit) {
#pragma unroll
                for (int i = 0; i < NV; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[i][e] += w[u][i][e];
            }
    }
    const float gs = p.gate_scale ? p.gate_scale[g] : 1.0f;
    float* crow = (float*)p.C + (long)row * p.ldc;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = v[i][e] * p.alpha;                 // same operation order as gemm_epilogue_quad
            if (p.bias) x += bq[i][e];
            x *= gq[i][e] * gs;
            x += rq[i][e];
            v[i][e] = x;
        }
        if (ok[i]) *(f32x4_t*)(crow + c) = v[i];
        else v[i] = zero4;
    }
    // adaLN of the finished row: cvar_ln_modulate's arithmetic (ops.hip:ln_modulate_kernel)
    const long lg = row / ln.rows_per;
    const float* sc = ln.scale + lg * ln.ld;
    const float* sh = ln.shift + lg * ln.ld;
    f32x4_t a[NV], b[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        a[i] = ok[i] ? *(const f32x4_t*)(sc + c) : zero4;
        b[i] = ok[i] ? *(const f32x4_t*)(sh + c) : zero4;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float)N;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][e] = ok[i] ? v[i][e] - mean : 0.f; q += v[i][e] * v[i][e]; }
    const float rstd = rsqrtf(wave_sum(q) / (float)N + ln.eps);
    TO* orow = (TO*)ln.out + (long)row * N;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (!ok[i]) continue;
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
static void rowfin_launch(bool tail, dim3 grid, hipStream_t st, const float* part, int nsplit, const GemmParams& p, const RowFinLn& ln) {
    if (tail) hipLaunchKernelGGL((cvar_splitk_rowfin_kernel<TO, NV, true>), grid, dim3(256), 0, st, part, nsplit, p, ln);
    else hipLaunchKernelGGL((cvar_splitk_rowfin_kernel<TO, NV, false>), grid, dim3(256), 0, st, part, nsplit, p, ln);
}

template <typename TO>
static int rowfin_dispatch(const float* part, int nsplit, const GemmParams& p, const RowFinLn& ln, hipStream_t st) {
    const int nv = (p.N + 255) / 256;
    const bool tail = (p.N % 256) != 0;
    dim3 grid(cdiv(p.M, 4));
    switch (nv) {
        case 1: rowfin_launch<TO, 1>(tail, grid, st, part, nsplit, p, ln); break;
        case 2: rowfin_launch<TO, 2>(tail, grid, st, part, nsplit, p, ln); break;
        case 3: rowfin_launch<TO, 3>(tail, grid, st, part, nsplit, p, ln); break;
        case 4: rowfin_launch<TO, 4>(tail, grid, st, part, nsplit, p, ln); break;
        case 5: rowfin_launch<TO, 5>(tail, grid, st, part, nsplit, p, ln); break;
        case 6: rowfin_launch<TO, 6>(tail, grid, st, part, nsplit, p, ln); break;
        case 7: rowfin_launch<TO, 7>(tail, grid, st, part, nsplit, p, ln); break;
        case 8: rowfin_launch<TO, 8>(tail, grid, st, part, nsplit, p, ln); break;
        default: return CVAR_EUNSUPPORTED;
    }
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// can this call's split-K reduction be finished row-wise with the adaLN behind it?  (the inference proj / fc2 calls: gate + fp32 residual in place, plain rows)
int cvar_splitk_rowfin_ok(const GemmParams& p, const cvar_gemm_desc* d) {
    return d->ln_out && d->ln_scale && d->ln_shift && d->ln_rows > 0 && p.gate && p.residual && p.res_dtype == CVAR_F32 && p.out_dtype == CVAR_F32 &&
           p.act == CVAR_ACT_NONE && !p.C2 && !p.aux && p.remap_l == 0 && p.split_n == 0 && p.ldc == p.N && p.ldr == p.N && (p.N % 4) == 0 && p.N <= 2048 &&
           (d->ld_ln % 4) == 0 && (p.ldg % 4) == 0 && (((uintptr_t)p.gate | (uintptr_t)p.residual | (uintptr_t)p.C | (uintptr_t)d->ln_scale | (uintptr_t)d->ln_shift |
           (uintptr_t)d->ln_out | (uintptr_t)p.bias) & 15) == 0 && (d->ln_out_dtype == CVAR_BF16 || d->ln_out_dtype == CVAR_F32);
}

int cvar_splitk_rowfin_launch(const float* part, int nsplit, const GemmParams& p, const cvar_gemm_desc* d, hipStream_t st) {
    RowFinLn ln;
    ln.out = d->ln_out; ln.scale = d->ln_scale; ln.shift = d->ln_shift; ln.ld = d->ld_ln; ln.rows_per = d->ln_rows; ln.eps = d->ln_eps;
    if (d->ln_out_dtype == CVAR_BF16) return rowfin_dispatch<bf16_t>(part, nsplit, p, ln, st);
    return rowfin_dispatch<float>(part, nsplit, p, ln, st);
}


