// Weight-gradient GEMM on token-major operands:  C[n][k] = sum_t A[t][n] * B[t][k]   (dW = dY^T X, train_control_var_hpu.py:231 under autograd).
//
// Both operands have the contraction index t as their SLOW index (activations and their gradients are [tokens][channels]), which the MFMA
// operand layout (8 consecutive contraction values per lane) cannot read from a row-major LDS tile with plain loads.  Round 1 transposed both
// tensors in HBM first (6.7 % of a training step).  gfx950's `ds_read_b64_tr_b16` reads a [4 t][16 col] block and hands lane i column i:
// two of them are one bf16x8 fragment, straight from the [t][col] image the DMA wrote.
//   tile 128 (n) x 256 (k) per workgroup, 4 waves x (64 x 128), K step = 32 tokens, ring of three 24 KB stages -> two workgroups per CU;
//   LDS rows are split in 64-byte segments XOR-swizzled by (t & 3): the four rows a transpose-read touches fall in four different bank groups;
//   split over the tokens (grid.z) with fp32 partial tiles in a caller workspace, summed in a fixed order (bit-reproducible).
//   Calls with few tiles (the square proj gradient) take a 256 x 256 tile on eight waves instead (same 64 x 128 wave tiles, ring of four 32 KB stages, one
//   workgroup per CU): a third less L2 -> LDS traffic per flop; the large calls stay on the 128-row tile, whose two independent workgroups per CU cover
//   each other's barriers (measured both ways, profiles/r04_train_gemm_tn.txt).
#include "cvar_common.h"

typedef __attribute__((address_space(3))) void* lptr_t;

struct GemmTnParams {
    const bf16_t* A; const bf16_t* B; float* C;
    long lda, ldb, ldc;          // elements
    int T, Nn, Kk;               // tokens, rows of C (columns of A), columns of C (columns of B)
    int tiles_k;                 // column tiles of C
    int steps_per_split, nsteps; // K steps (32 tokens) per grid.z slice, total
    long split_stride;           // elements between the partial outputs of consecutive slices (0: single slice writes C directly)
    float* colsum; long cs_stride;   // optional: colsum[n] = sum_t A[t][n] (the bias gradient of the layer whose dY is A); per-slice stride (0: single slice)
};

constexpr int TN_BN = 256, TN_BK = 32;
constexpr int TN_B_BYTES = TN_BK * TN_BN * 2;       // 16 KB: [32 t][512 B]
#ifndef CVAR_TN_TILE256
#define CVAR_TN_TILE256 1       // 256 (n) x 256 (k) tile on eight waves where Nn % 256 == 0 (one workgroup per CU, ring of CVAR_TN_STAGES256 x 32 KB)
#endif
#ifndef CVAR_TN_TILE256_MAX
#define CVAR_TN_TILE256_MAX 128
#endif
#ifndef CVAR_TN_STAGES256
#define CVAR_TN_STAGES256 4
#endif

// The transpose-read is issued as inline asm: through the builtin the compiler treats it as an LDS read that may alias the DMA writes and puts
// an `s_waitcnt vmcnt(0)` in front of every group - which waits for the stage issued in this very step and throws the ring's depth away.
// As asm the read is invisible to the wait-count pass, so the lgkmcnt waits are written by hand, tied to the fragment registers.
typedef int v2i_t __attribute__((ext_vector_type(2)));
#define TN_TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
__device__ __forceinline__ bf16x8_t tn_pack(const v2i_t lo, const v2i_t hi) {
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    const v4i_t q = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8_t, q);
}

// BM = 128: 4 waves, two workgroups per CU;  BM = 256: 8 waves (4 x 2), one workgroup per CU.  The wave tile is 64 x 128 in both.
template <int TN_BM, int NST>
__global__ __launch_bounds__(TN_BM * 2, TN_BM == 128 ? 2 : 1) void gemm_tn_bf16_kernel(const GemmTnParams p) {
    constexpr int NW = TN_BM / 32;                       // waves
    constexpr int TN_A_BYTES = TN_BK * TN_BM * 2;        // [32 t][BM x 2 B]
    constexpr int TN_STAGE = TN_A_BYTES + TN_B_BYTES;
    constexpr int A_ROW = TN_BM * 2;                     // bytes per token row of the A image
    constexpr int A_RPP = 1024 / A_ROW, A_LPR = 64 / A_RPP;      // token rows per 1 KiB DMA piece, lanes per row
    constexpr int A_PW = (TN_A_BYTES / 1024) / NW, B_PW = (TN_B_BYTES / 1024) / NW;      // pieces per wave and step
    __shared__ __attribute__((aligned(1024))) char smem[NST * TN_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % (TN_BM / 64), wn = wave / (TN_BM / 64);          // wave tile: rows 64 wm .. +64, columns 128 wn .. +128
    const int tile = blockIdx.x;
    const int tk = tile % p.tiles_k, tn = tile / p.tiles_k;
    const int n0 = tn * TN_BM, k0 = tk * TN_BN;
    const int step0 = blockIdx.z * p.steps_per_split;
    const int nstep = min(p.steps_per_split, p.nsteps - step0);
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + n0), 0, (int)min((long)0x7fffffff, ((long)p.T * p.lda - n0) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + k0), 0, (int)min((long)0x7fffffff, ((long)p.T * p.ldb - k0) * 2), 0x00020000);

    // DMA pieces (1 KiB = 64 lanes x 16 B, lane-linear in LDS).  A: piece q holds token rows 4q..4q+3 (256-B rows), B: rows 2q, 2q+1 (512-B rows).
    // The lane fetches the LOGICAL chunk whose swizzled home is its physical slot: segment' = segment ^ (t & 3).
    unsigned a_off[A_PW], b_off[B_PW];
#pragma unroll
    for (int jj = 0; jj < A_PW; ++jj) {
        const int q = wave + NW * jj;
        const int t = A_RPP * q + lane / A_LPR, pc = lane % A_LPR;
        const int lc = (((pc >> 2) ^ (t & 3)) << 2) | (pc & 3);
        a_off[jj] = (unsigned)((t * p.lda + lc * 8) * 2);
    }
#pragma unroll
    for (int jj = 0; jj < B_PW; ++jj) {
        const int q = wave + NW * jj;
        const int t = 2 * q + (lane >> 5), pc = lane & 31;
        const int lc = (((pc >> 2) ^ (t & 3)) << 2) | (pc & 3);
        b_off[jj] = (unsigned)((t * p.ldb + lc * 8) * 2);
    }
    auto issue = [&](int step, int slot) {            // all six pieces of this wave for K step `step` (token rows 32 step ..)
        char* sb = smem + slot * TN_STAGE;
        const int sa = (int)((long)(step0 + step) * TN_BK * p.lda * 2), sbo = (int)((long)(step0 + step) * TN_BK * p.ldb * 2);
#pragma unroll
        for (int jj = 0; jj < A_PW; ++jj)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lptr_t)(sb + (wave + NW * jj) * 1024), 16, (int)a_off[jj], sa, 0, 0);
#pragma unroll
        for (int jj = 0; jj < B_PW; ++jj)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lptr_t)(sb + TN_A_BYTES + (wave + NW * jj) * 1024), 16, (int)b_off[jj], sbo, 0, 0);
    };

    // transpose-read addressing: lane l of a 16-lane group points at row (l & 15) >> 2 of the [4 t][16 col] block, columns 4 (l & 3) .. +3;
    // groups 0 / 1 of a half-wave are the two 16-column halves of a 32-wide MFMA block, the upper half-wave is the k-group 8 tokens later.
    const int l15 = lane & 15, g = (lane >> 4) & 1, hi = lane >> 5;
    const int jrow = l15 >> 2, m = l15 & 3;
    const int t_lane = 8 * hi + jrow;                                          // + 16 ks + 4 q
    const int inseg = ((2 * g + (m >> 1)) << 4) + (m & 1) * 8;                 // byte offset inside the 64-B segment
    // A: 4 segments per 256-B row, segment of block i of this wave = 2 wm + i;  B: 8 segments per 512-B row, segment = 4 wn + j
    int a_lane[2], b_lane[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) a_lane[i] = t_lane * A_ROW + (((2 * wm + i) ^ jrow) << 6) + inseg;
#pragma unroll
    for (int j = 0; j < 4; ++j) b_lane[j] = TN_A_BYTES + t_lane * 512 + (((4 * wn + j) ^ jrow) << 6) + inseg;

    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;        // LDS byte address of the ring

    f32x16_t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // bias gradient: column sums of A on the matrix pipe.  Only the workgroups of the first column tile (tk == 0) and in them the waves of the first
    // column half (wn == 0) carry it: A fragments x a B operand of ones -> every column of the 32x32 result holds sum_t A[t][n]; 4 extra MFMAs per step
    // for 1 / (2 tiles_k) of the waves instead of a separate pass that re-reads dY (colsum_vec_kernel: 2.3 % of a d24 training step)
    const bool do_cs = p.colsum != nullptr && tk == 0 && wn == 0;
    f32x16_t accb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
    const bf16x8_t ones8 = {(short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80};

    constexpr int PW = A_PW + B_PW;                  // DMA pieces of one wave per step
#pragma unroll
    for (int u = 0; u < NST - 1; ++u)
        if (u < nstep) issue(u, u);
    for (int s = 0; s < nstep; ++s) {
        // stage s was issued NST - 1 steps ago (or in the prologue): wait for everything but the pieces of the NST - 2 newer steps, then publish
        if (NST == 4 && s + 2 < nstep) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
        else if (s + 1 < nstep) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + NST - 1 < nstep) issue(s + NST - 1, (s + NST - 1) % NST);
        const unsigned st = lds0 + (unsigned)((s % NST) * TN_STAGE);
        unsigned aa[2], ba[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) aa[i] = st + (unsigned)a_lane[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) ba[j] = st + (unsigned)b_lane[j];
        v2i_t ar[2][2][2], br[2][4][2];          // [ks][block][q]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (ks == 0) { TN_TR_READ(ar[0][i][0], aa[i], 0); TN_TR_READ(ar[0][i][1], aa[i], 4 * A_ROW); }
                else { TN_TR_READ(ar[1][i][0], aa[i], 16 * A_ROW); TN_TR_READ(ar[1][i][1], aa[i], 20 * A_ROW); }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ks == 0) { TN_TR_READ(br[0][j][0], ba[j], 0); TN_TR_READ(br[0][j][1], ba[j], 4 * 512); }
                else { TN_TR_READ(br[1][j][0], ba[j], 16 * 512); TN_TR_READ(br[1][j][1], ba[j], 20 * 512); }
            }
        }
        // LDS returns in order: at most 12 outstanding = the ks = 0 set is back
        asm volatile("s_waitcnt lgkmcnt(12)"
                     : "+v"(ar[0][0][0]), "+v"(ar[0][0][1]), "+v"(ar[0][1][0]), "+v"(ar[0][1][1]), "+v"(br[0][0][0]), "+v"(br[0][0][1]), "+v"(br[0][1][0]),
                       "+v"(br[0][1][1]), "+v"(br[0][2][0]), "+v"(br[0][2][1]), "+v"(br[0][3][0]), "+v"(br[0][3][1]));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tn_pack(ar[0][i][0], ar[0][i][1]), tn_pack(br[0][j][0], br[0][j][1]), acc[i][j], 0, 0, 0);
        if (do_cs) {
#pragma unroll
            for (int i = 0; i < 2; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tn_pack(ar[0][i][0], ar[0][i][1]), ones8, accb[i], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ar[1][0][0]), "+v"(ar[1][0][1]), "+v"(ar[1][1][0]), "+v"(ar[1][1][1]), "+v"(br[1][0][0]), "+v"(br[1][0][1]), "+v"(br[1][1][0]),
                       "+v"(br[1][1][1]), "+v"(br[1][2][0]), "+v"(br[1][2][1]), "+v"(br[1][3][0]), "+v"(br[1][3][1]));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tn_pack(ar[1][i][0], ar[1][i][1]), tn_pack(br[1][j][0], br[1][j][1]), acc[i][j], 0, 0, 0);
        if (do_cs) {
#pragma unroll
            for (int i = 0; i < 2; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tn_pack(ar[1][i][0], ar[1][i][1]), ones8, accb[i], 0, 0, 0);
        }
    }

    // D[n][k]: lane holds column k = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 hi of the 32x32 block -> 128-byte row segments per store
    float* cbase = p.C + (long)blockIdx.z * p.split_stride;
    const int lrow = lane & 31;
    // Kk may end in the middle of the last column tile (Kk % 128 == 0: d30's C = 1920 = 7.5 tiles): that tile's second wave column holds sums over
    // whatever the B rows continue with - never stored
    if (k0 + 128 * wn < p.Kk)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int k = k0 + 128 * wn + 32 * j + lrow;
                cbase[(long)n * p.ldc + k] = acc[i][j][r];
            }
    if (do_cs && lrow == 0) {                       // column 0 of the block: lanes 0 (hi = 0) and 32 (hi = 1) hold the 32 row sums between them
        float* cs = p.colsum + (long)blockIdx.z * p.cs_stride;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) cs[n0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi] = accb[i][r];
    }
}

// out[i] = sum_s part[s][i] in slice order (fixed -> bit-reproducible); row-major [Nn][Kk] partials -> C with leading dimension ldc
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, long ldc, int Nn, int Kk, int nsplit,
                                                             const float* __restrict__ cs_part, float* __restrict__ cs_out) {
    if (cs_out) {                                   // the slices' column sums, same fixed order; spread over the grid (one block doing all of it was the launch's straggler)
        for (long n = (long)blockIdx.x * 256 + threadIdx.x; n < Nn; n += (long)gridDim.x * 256) {
            float v = cs_part[n];
            for (int s = 1; s < nsplit; ++s) v += cs_part[(long)s * Nn + n];
            cs_out[n] = v;
        }
    }
    const long nvec = (long)Nn * (Kk / 4);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const int n = (int)(i / (Kk / 4)), k = (int)(i % (Kk / 4)) * 4;
        f32x4_t v = *(const f32x4_t*)(part + (long)n * Kk + k);
        for (int s0 = 1; s0 < nsplit; s0 += 8) {             // slice order kept, loads of eight slices in flight
            f32x4_t w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < nsplit) w[u] = *(const f32x4_t*)(part + (long)(s0 + u) * Nn * Kk + (long)n * Kk + k);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < nsplit) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += w[u][e];
                }
        }
        *(f32x4_t*)(C + (long)n * ldc + k) = v;
    }
}

/* C[Nn][Kk] (fp32, leading dimension ldc) = A^T B with A = [T][lda] (Nn columns used), B = [T][ldb] (Kk columns used), both bf16.
 * Nn % 128 == 0, Kk % 128 == 0 (column tiles are 256 wide; a half-filled last one stores its first 128 columns); lda, ldb multiples of 8 (any T: the last K step is zero-filled); ws: caller workspace for the token-split partials (may be NULL: one slice).
 * colsum_a (optional, ABI 14): colsum_a[n] = sum_t A[t][n], the bias gradient of the same layer, from the A fragments the kernel holds anyway. */
extern "C" int cvar_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int T, int Nn, int Kk,
                            float* ws, int64_t ws_bytes, float* colsum_a, void* stream) {
    if (!A || !B || !C || T <= 0 || Nn <= 0 || Kk <= 0) return CVAR_EINVAL;
    if (Nn % 128 || Kk % (TN_BN / 2) || lda % 8 || ldb % 8 || ldc % 4 || lda < Nn || ldb < Kk || ldc < Kk) return CVAR_EUNSUPPORTED;
    if ((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) || (long)T * lda * 2 >= 0x7fffffffL || (long)T * ldb * 2 >= 0x7fffffffL) return CVAR_EUNSUPPORTED;
    GemmTnParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.T = T; p.Nn = Nn; p.Kk = Kk;
    p.tiles_k = (Kk + TN_BN - 1) / TN_BN;
    p.nsteps = (T + TN_BK - 1) / TN_BK;          // token rows past T lie outside the buffer resources: the DMA writes zeros for them
    // 256-row tile: measured on the d24 shapes at T = 43 520 (tools/gemm_tn_ab.py, profiles/r04_train_gemm_tn.txt) it wins where the 128-row tiles are few
    // (proj 1536 x 1536: 260 -> 216 us) and loses 3 % on the large ones (qkv / fc1 / fc2), so it takes the calls with <= CVAR_TN_TILE256_MAX 128-row tiles
    const bool big = CVAR_TN_TILE256 && Nn % 256 == 0 && (Nn / 128) * p.tiles_k <= CVAR_TN_TILE256_MAX;
    const int tiles = (Nn / (big ? 256 : 128)) * p.tiles_k, slots = big ? 256 : 512;
    // slices over the tokens: fill whole rounds of the workgroup slots (two per CU for the 128-row tile, one for the 256-row tile), at least 24 K steps per slice, workspace permitting
    int best = 1;
    if (ws && (((uintptr_t)ws & 15) == 0)) {
        double best_cost = 1e30;
        for (int sp = 1; sp <= 16; ++sp) {
            if (p.nsteps / sp < 24 && sp > 1) break;
            if (sp > 1 && (size_t)sp * ((size_t)Nn * Kk + (colsum_a ? Nn : 0)) * sizeof(float) > (size_t)ws_bytes) break;
            const double cost = (double)((tiles * sp + slots - 1) / slots) / sp + 0.004 * sp;       // rounds per slice + a price for the partial traffic
            if (cost < best_cost) { best_cost = cost; best = sp; }
        }
    }
    p.steps_per_split = (p.nsteps + best - 1) / best;
    const int nsplit = (p.nsteps + p.steps_per_split - 1) / p.steps_per_split;
    hipStream_t st = as_stream(stream);
    auto launch = [&](int nz) {
        if (big) hipLaunchKernelGGL((gemm_tn_bf16_kernel<256, CVAR_TN_STAGES256>), dim3(tiles, 1, nz), dim3(512), 0, st, p);
        else hipLaunchKernelGGL((gemm_tn_bf16_kernel<128, 3>), dim3(tiles, 1, nz), dim3(256), 0, st, p);
    };
    if (nsplit == 1) {
        p.C = C; p.split_stride = 0; p.colsum = colsum_a; p.cs_stride = 0;
        launch(1);
    } else {
        float* cs_part = colsum_a ? ws + (size_t)nsplit * Nn * Kk : nullptr;          // behind the partial tiles
        p.C = ws; p.ldc = Kk; p.split_stride = (long)Nn * Kk; p.colsum = cs_part; p.cs_stride = Nn;
        launch(nsplit);
        const long nvec = (long)Nn * (Kk / 4);
        hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)min((long)2048, (nvec + 255) / 256)), dim3(256), 0, st, ws, C, (long)ldc, Nn, Kk, nsplit, cs_part, colsum_a);
    }
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}
