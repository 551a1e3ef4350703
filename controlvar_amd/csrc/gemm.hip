// MFMA GEMM / implicit-GEMM 3x3 convolution with fused epilogue for gfx950 (MI355X).
//
//   C[row(m), n] = cast( residual + gate * act(alpha * sum_k A[m,k] W[n,k] + bias[n]) )
//
// Design (cdna_hip_programming.md section 5):
//   * block tile BM x BN, K tile = 128 bytes of K per row (64 bf16 / 32 f32), WM x WN waves, each wave owns an
//     (BM/WM) x (BN/WN) sub-tile built from 32x32 MFMA blocks: v_mfma_f32_32x32x16_bf16 (bf16 mode) or the
//     exact-f32 v_mfma_f32_32x32x2_f32 (parity mode);
//   * global -> LDS by global_load_lds_dwordx4 (16 B per lane, no VGPR round trip), double buffered, counted
//     vmcnt so the next tile's DMA stays in flight across the barrier;
//   * LDS rows are 128 B; the 16-B chunk index is XOR-swizzled with (row>>1)&7 on the SOURCE side (LDS image
//     stays lane-linear as the DMA requires) and on the ds_read_b128 side -> conflict-free fragment reads;
//   * out-of-range rows / K tail / conv zero padding read a 16-byte zero chunk instead of branching;
//   * conv mode gathers A rows from NHWC activations (3x3 taps, stride 1 or the (0,1,0,1)-padded stride 2,
//     optional nearest x2 upsample folded into the address);
//   * blockIdx -> tile mapping is XCD-aware (contiguous tile ranges per XCD L2) and grouped along M.
// This file is compiled three times (compile time: every (tile, epilogue) combination is a kernel): as gemm.hip - the C ABI, the
// bf16 GEMM kernels, split-K; from gemm_conv.hip (CVAR_GEMM_CONV_TU) - the bf16 implicit-conv kernels; from gemm_f32.hip
// (CVAR_GEMM_F32_TU) - everything fp32 (parity mode).
#include "cvar_common.h"
#include "gemm_params.h"
#if defined(CVAR_GEMM_CONV_TU)
#define CVAR_TU_CONV 1
#define CVAR_TU_PLAIN 0
#elif defined(CVAR_GEMM_F32_TU)
#define CVAR_TU_CONV 1
#define CVAR_TU_PLAIN 1
#else
#define CVAR_TU_CONV 0
#define CVAR_TU_PLAIN 1
#endif
#include <type_traits>

static __device__ __attribute__((aligned(16))) unsigned int cvar_zero_chunk[4] = {0u, 0u, 0u, 0u};   // one per translation unit
#if defined(CVAR_GEMM_TIMING) && defined(CVAR_GEMM_F32_TU)
#undef CVAR_GEMM_TIMING
#endif
#ifdef CVAR_GEMM_TIMING
__device__ unsigned long long cvar_gemm_dbg[64 * 8 * 8];
__device__ unsigned long long cvar_gemm_dbg_tot[8];
// workgroup timeline (round 4): per workgroup {s_memrealtime at entry, at the first MFMA, at the end of the K loop, at exit, HW_ID | XCC_ID << 32}
#define CVAR_DBG_WG_MAX 16384
__device__ unsigned long long cvar_gemm_dbg_wg[CVAR_DBG_WG_MAX * 5];
extern "C" int cvar_gemm_dbg_wg_read(unsigned long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cvar_gemm_dbg_wg), sizeof(unsigned long long) * 5 * (size_t)(n < CVAR_DBG_WG_MAX ? n : CVAR_DBG_WG_MAX)); }
extern "C" int cvar_gemm_dbg_tot_read(unsigned long long* host, int reset) { int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cvar_gemm_dbg_tot), 64); if (reset) { unsigned long long z[8] = {0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(cvar_gemm_dbg_tot), z, 64); } return rc; }
extern "C" int cvar_gemm_dbg_read(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cvar_gemm_dbg), sizeof(unsigned long long) * 64 * 8 * 8); }
#endif

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef int v4i_t __attribute__((ext_vector_type(4)));

template <typename T, int BM, int BN, int WM, int WN, bool CONV, int NSTAGE = 2, bool FAST = false, bool CUP = false>
__device__ __forceinline__ void cvar_gemm_tile(const GemmParams& p, const int vblock) {
    static_assert(!CUP || (CONV && FAST), "CUP (nearest x2 upsample folded into the conv) is a conv FAST variant");
    constexpr int NW = WM * WN;
    constexpr bool FRAG_PIPE = (BM == 256) && (!CONV || FAST);
#ifndef CVAR_DMA_EARLY
#define CVAR_DMA_EARLY 1
#endif
#ifndef CVAR_DMA_SPAN
#define CVAR_DMA_SPAN 2      // the next tile's DMA pieces are issued inside the first CVAR_DMA_SPAN of the 4 k-steps
#endif
    constexpr int DMA_EARLY = CVAR_DMA_EARLY;
    // the next tile's DMA pieces are issued inside the first SPAN_NUM / SPAN_DEN of the tile's 4 k-steps: one wave per SIMD pays
    // every DMA issue with a matrix-pipe bubble, so they are spread over 2 k-steps; with two waves per SIMD the partner wave fills
    // the bubble and the earliest issue (1 k-step) wins (measured 989 vs 979 vs 964 TFLOP/s in situ for 1 / 2 / 3 k-steps)
#ifdef CVAR_DMA_SPAN8_NUM
    constexpr int SPAN_NUM = NW >= 8 ? CVAR_DMA_SPAN8_NUM : CVAR_DMA_SPAN, SPAN_DEN = NW >= 8 ? CVAR_DMA_SPAN8_DEN : 1;
#else
    constexpr int SPAN_NUM = NW >= 8 ? 1 : CVAR_DMA_SPAN, SPAN_DEN = 1;
#endif
    constexpr int ES = sizeof(T);
    constexpr int KCH = 16 / ES;         // elements per 16-byte chunk
    constexpr int KT = 128 / ES;         // elements of K per tile
    constexpr int A_INSTR = BM / 8, B_INSTR = BN / 8;     // 1 KiB wave-instructions per tile
    constexpr int A_PER_W = A_INSTR / NW, B_PER_W = (B_INSTR + NW - 1) / NW;
    static_assert(A_INSTR % NW == 0, "tile / wave mismatch");
    // W pieces need not divide among the waves (160 columns = 20 pieces over 8 waves): the surplus slots of the last round re-issue
    // pieces that another wave issues as well - same source, same LDS destination, identical bytes - so that every wave still issues NL
    // pieces per K tile and the counted vmcnt stays uniform
    constexpr int B_DUP = B_PER_W * NW - B_INSTR;
    static_assert(B_DUP >= 0 && B_DUP < NW, "tile / wave mismatch");
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int SUB_M = BM / WM, SUB_N = BN / WN;
    constexpr int MI = SUB_M / 32, NJ = SUB_N / 32;
    static_assert(SUB_M % 32 == 0 && SUB_N % 32 == 0, "sub-tile must be 32x32 blocks");
    // accumulator blocks are kept transposed (swapped MFMA operands) when a 32-row staging region per wave fits the pipeline LDS
    constexpr bool TRANS = NW * 32 * (SUB_N + 4) * 4 <= NSTAGE * STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef CVAR_GEMM_TIMING
    unsigned long long dbg_rt_entry = __builtin_amdgcn_s_memrealtime();
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    auto w_piece = [&](int jj) { const int j = wave + jj * NW; return j < B_INSTR ? j : j - B_DUP; };

    // ---- start stagger.  Tiles of one launch take the same time, so all CUs reach their epilogues together: HBM idles during
    // the K loops (operands come out of L2) and is saturated by 256 simultaneous output bursts (plus residual reads) at the
    // tile boundaries - the workgroups then all wait on the same queue, which also keeps them in phase for the next round.
    // The FIRST workgroup of each CU (blockIdx < 256: one per CU and round) sleeps 0..stagger cycles by its index inside the XCD;
    // later workgroups inherit the phase of the one they replace.  Costs <= `stagger` cycles at the tail of the launch.
    if (p.stagger > 0 && blockIdx.x < 256u && gridDim.y == 1 && gridDim.z == 1) {
        const unsigned ph = (blockIdx.x >> 3) & 31u;
        if (ph) {
            const unsigned long long t_end = __builtin_amdgcn_s_memtime() + (unsigned long long)ph * (unsigned)p.stagger / 32u;
            while (__builtin_amdgcn_s_memtime() < t_end) __builtin_amdgcn_s_sleep(8);
        }
    }

    // ---- tile coordinates: bijective XCD remap, then grouped-M ordering
    int m0, n0;
    gemm_tile_origin(p, vblock, BM, BN, m0, n0);
    const long zb = blockIdx.z;

    const char* const zero = (const char*)cvar_zero_chunk;
    const char* Abase = p.A + zb * p.strideA * ES;
    const char* Wbase = p.W + zb * p.strideW * ES;

    // ---- per-lane load descriptors
    const int lr = lane >> 3, slot = lane & 7;
    const char* a_ptr[A_PER_W];      // plain: row base + chunk offset (nullptr if row out of range)
    int a_k0[A_PER_W];               // element offset of this lane's chunk inside a K tile
    int a_b[A_PER_W], a_oy[A_PER_W], a_ox[A_PER_W];   // conv: decoded output pixel (a_b < 0: invalid)
    int a_tap[A_PER_W], a_ci[A_PER_W];                // conv: (tap, channel) of this lane's chunk in the NEXT tile to issue
#pragma unroll
    for (int jj = 0; jj < A_PER_W; ++jj) {
        const int j = wave + jj * NW;
        const int row = j * 8 + lr;
        const int chunk = slot ^ ((row >> 1) & 7);
        a_k0[jj] = chunk * KCH;
        const int m = m0 + row;
        a_tap[jj] = 0; a_ci[jj] = 0;
        if (CONV) {
            a_ptr[jj] = nullptr;
            { const int kfirst = (int)blockIdx.y * p.split_tiles * KT + a_k0[jj]; a_tap[jj] = kfirst / p.Cin; a_ci[jj] = kfirst - a_tap[jj] * p.Cin; }
            if (m < p.M) {
                const int hw = p.Hout * p.Wout;
                const int b = m / hw, rem = m - b * hw;
                a_b[jj] = b; a_oy[jj] = rem / p.Wout; a_ox[jj] = rem - a_oy[jj] * p.Wout;
            } else {
                a_b[jj] = -1; a_oy[jj] = 0; a_ox[jj] = 0;
            }
        } else {
            a_b[jj] = 0; a_oy[jj] = 0; a_ox[jj] = 0;
            a_ptr[jj] = (m < p.M) ? Abase + ((long)m * p.lda + a_k0[jj]) * ES : nullptr;
        }
    }
    // FAST (host guarantees K % KT == 0 and 32-bit row offsets): a piece's address is a wave-uniform tile base that advances
    // by 128 B per K tile (scalar arithmetic) plus a per-lane 32-bit offset that never changes -> no vector address math in the
    // K loop.  Rows past M / N are clamped to the last valid row: they only feed output rows / columns that are never stored.
    unsigned a_off[A_PER_W], w_off[B_PER_W];
    const char* const a_tile = CONV ? Abase : Abase + (long)m0 * p.lda * ES;
    const char* const w_tile = Wbase + (long)n0 * p.ldw * ES;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a_tile, 0, CONV ? (int)p.conv_bytes : 0x7fffffff, 0x00020000);
    // W's range ends with the last row of the matrix: when K is not a multiple of the K tile (conv: K = 9 Cin) the last row's
    // tail would otherwise read whatever follows the weights - the A side is zero there, but 0 x Inf/NaN garbage is NaN
    const long w_bytes = ((long)(p.N - n0 - 1) * p.ldw + p.K) * ES;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w_tile, 0, (int)min(w_bytes, (long)0x7fffffff), 0x00020000);
    // conv FAST (stride 1, no upsample, Cin % 32 == 0, input < 2 GiB): each 32-element half of a K tile lies inside ONE tap, so
    // (tap, channel) of the two halves are wave-uniform scalars advanced per K tile; a lane keeps its pixel's byte offset and a
    // 9-bit mask of in-range taps per piece, and because the swizzled chunk of a lane is the same for all its pieces
    // ((row >> 1) & 7 does not depend on jj when NW is even) the half it reads from is a lane constant:
    //   offset = pixel + (half ? delta1 : delta0),  valid = mask >> (half ? tap1 : tap0) & 1,  invalid -> out-of-range offset -> zeros
    unsigned c_vm[CONV ? A_PER_W : 1];
    const int lane_half = (slot ^ ((((wave * 8 + lr)) >> 1) & 7)) >> 2;
    int cf_tap = 0, cf_ci = 0, lane_D = 0, lane_T = 15, lane_ky = 0, lane_kx = 0;
    static_assert(!(CONV && FAST) || (NW % 2 == 0), "lane-constant half needs an even wave count");
    if (FAST && !CONV) {
#pragma unroll
        for (int jj = 0; jj < A_PER_W; ++jj) {
            const int row = (wave + jj * NW) * 8 + lr;
            a_off[jj] = (unsigned)min(row, p.M - 1 - m0) * (unsigned)(p.lda * ES) + (unsigned)((slot ^ ((row >> 1) & 7)) * 16);
        }
    }
    if (FAST && CONV) {
#pragma unroll
        for (int jj = 0; jj < A_PER_W; ++jj) {
            const int row = (wave + jj * NW) * 8 + lr;
            const int chunk = slot ^ ((row >> 1) & 7);
            unsigned vm = 0;
            const int b = a_b[jj], oy = a_oy[jj], ox = a_ox[jj];
            if (b >= 0) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (p.stride == 2) {                                          // (0,1,0,1)-padded stride 2: taps start AT the pixel
                        const int iy = 2 * oy + t / 3, ix = 2 * ox + t % 3;
                        if (iy < p.Hin && ix < p.Win) vm |= 1u << t;
                    } else {
                        const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;       // tap position on the OUTPUT grid (= input grid unless CUP)
                        if (iy >= 0 && iy < p.Hout && ix >= 0 && ix < p.Wout) vm |= 1u << t;
                    }
                }
            }
            if (CUP) {
                // nearest x2 upsample folded in: tap (ky, kx) of output pixel (oy, ox) reads input pixel
                // ((oy + ky - 1) >> 1, (ox + kx - 1) >> 1) = (oy >> 1) + ((py + ky - 1) >> 1), ... with py = oy & 1: a -1 / 0 / +1
                // step around the base pixel (oy >> 1, ox >> 1); the parities ride in bits 16, 17 of the tap mask (the tap index never exceeds 15)
                vm |= (unsigned)(oy & 1) << 16 | (unsigned)(ox & 1) << 17;
                a_off[jj] = (unsigned)(((max(b, 0) * p.Hin + (oy >> 1)) * p.Win + (ox >> 1)) * p.Cin * ES + (chunk & 3) * 16);
            } else {
                const int sy = p.stride == 2 ? 2 * oy : oy, sx = p.stride == 2 ? 2 * ox : ox;
                a_off[jj] = (unsigned)(((max(b, 0) * p.Hin + sy) * p.Win + sx) * p.Cin * ES + (chunk & 3) * 16);
            }
            c_vm[jj] = vm;
        }
    }
    if (FAST) {
#pragma unroll
        for (int jj = 0; jj < B_PER_W; ++jj) {
            const int row = w_piece(jj) * 8 + lr;
            w_off[jj] = (unsigned)min(row, p.N - 1 - n0) * (unsigned)(p.ldw * ES) + (unsigned)((slot ^ ((row >> 1) & 7)) * 16);
        }
    }
    // conv FAST: lane offset / tap of the NEXT tile to issue, then advance the scalar (tap, channel) state by one K tile
    auto conv_next = [&]() {
        const int t0 = cf_tap, c0 = cf_ci;
        int c1 = c0 + 32, t1 = t0;
        if (c1 >= p.Cin) { c1 -= p.Cin; ++t1; }
        auto delta = [&](int t, int c) {
            const int tt = min(t, 8), ky = (tt * 11) >> 5, kx = tt - ky * 3;
            const int pad = p.stride == 2 ? 0 : 1;
            return (((ky - pad) * p.Win + (kx - pad)) * p.Cin + c) * ES;
        };
        const int d0 = delta(t0, c0), d1 = delta(t1, c1);
        lane_D = lane_half ? d1 : d0;
        lane_T = lane_half ? min(t1, 15) : min(t0, 15);     // taps >= 9 (K tail) index mask bits 9..15, which are always 0
        if (CUP) {
            const int tt = min(lane_T, 8);
            lane_ky = (tt * 11) >> 5;
            lane_kx = tt - lane_ky * 3;
            lane_D = (lane_half ? c1 : c0) * ES;          // channel part only; the pixel step depends on the piece's row parity
        }
        int c2 = c1 + 32, t2 = t1;
        if (c2 >= p.Cin) { c2 -= p.Cin; ++t2; }
        cf_tap = t2; cf_ci = c2;
    };
    const char* w_ptr[B_PER_W];
    int w_k0[B_PER_W];
#pragma unroll
    for (int jj = 0; jj < B_PER_W; ++jj) {
        const int j = w_piece(jj);
        const int row = j * 8 + lr;
        const int chunk = slot ^ ((row >> 1) & 7);
        w_k0[jj] = chunk * KCH;
        const int n = n0 + row;
        w_ptr[jj] = (n < p.N) ? Wbase + ((long)n * p.ldw + w_k0[jj]) * ES : nullptr;
    }

    // one 1-KiB DMA piece of tile kt: idx < A_PER_W -> A operand, else W operand
    // the K loop issues pieces unconditionally (past the last tile it re-fetches the last one into a stage nobody reads), so
    // that its body is one basic block and the scheduler can place DMA issue and address math between MFMAs
    auto issue_one = [&](int kt, int stage, int idx) {
        char* sbase = smem + stage * STAGE;
        if (FAST) {
            // buffer_load_dwordx4 v_off, s[rsrc], s_koff offen lds: resource and K offset are scalar, the lane offset is fixed
            if (idx < A_PER_W) {
                if (CONV) {
                    int step = lane_D;
                    if (CUP) {
                        const int dy = ((int)((c_vm[idx] >> 16) & 1u) + lane_ky - 1) >> 1;
                        const int dx = ((int)((c_vm[idx] >> 17) & 1u) + lane_kx - 1) >> 1;
                        step += (dy * p.Win + dx) * p.Cin * ES;
                    }
                    const int off = ((c_vm[idx] >> lane_T) & 1u) ? (int)(a_off[idx] + (unsigned)step) : (int)0x80000000;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lptr_t)(sbase + (wave + idx * NW) * 1024), 16, off, 0, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lptr_t)(sbase + (wave + idx * NW) * 1024), 16, (int)a_off[idx], kt * 128, 0, 0);
                }
            } else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lptr_t)(sbase + BM * 128 + w_piece(idx - A_PER_W) * 1024), 16,
                                                         (int)w_off[idx - A_PER_W], kt * 128, 0, 0);
            return;
        }
        if (idx < A_PER_W) {
            const int jj = idx;
            const int j = wave + jj * NW;
            const int k = kt * KT + a_k0[jj];
            const char* src = zero;
            if (CONV) {
                // pieces of one lane are issued in increasing kt, so (tap, ci) advance incrementally: no division in the loop
                const int tap = a_tap[jj], ci = a_ci[jj];
                if (a_b[jj] >= 0 && tap < 9) {
                    const int ky = (tap * 11) >> 5, kx = tap - ky * 3;        // tap / 3, tap % 3 for tap < 9
                    int iy, ix;
                    bool ok;
                    if (p.stride == 1) {
                        iy = a_oy[jj] + ky - 1; ix = a_ox[jj] + kx - 1;
                        ok = (iy >= 0) && (iy < p.Hout) && (ix >= 0) && (ix < p.Wout);
                        if (p.up) { iy >>= 1; ix >>= 1; }
                    } else {
                        iy = 2 * a_oy[jj] + ky; ix = 2 * a_ox[jj] + kx;
                        ok = (iy < p.Hin) && (ix < p.Win);
                    }
                    if (ok) src = Abase + ((((long)a_b[jj] * p.Hin + iy) * p.Win + ix) * p.Cin + ci) * ES;
                }
                int nci = ci + p.cv_rem, ntap = tap + p.cv_adv;
                if (nci >= p.Cin) { nci -= p.Cin; ++ntap; }
                a_ci[jj] = nci; a_tap[jj] = ntap;
            } else {
                if (a_ptr[jj] != nullptr && k < p.K) src = a_ptr[jj] + (long)kt * 128;
            }
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sbase + j * 1024), 16, 0, 0);
        } else {
            const int jj = idx - A_PER_W;
            const int j = w_piece(jj);
            const int k = kt * KT + w_k0[jj];
            const char* src = zero;
            if (w_ptr[jj] != nullptr && k < p.K) src = w_ptr[jj] + (long)kt * 128;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sbase + BM * 128 + j * 1024), 16, 0, 0);
        }
    };
    constexpr int NL = A_PER_W + B_PER_W;     // DMA pieces per wave per tile

    f32x16_t acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

#ifndef CVAR_GEMM_M16
#define CVAR_GEMM_M16 1
#endif
#ifndef CVAR_GEMM_XB
#define CVAR_GEMM_XB 1
#endif
#ifndef CVAR_GEMM_128_W8
#define CVAR_GEMM_128_W8 1
#endif
// cost of a round of 128x128 tiles against a round of 256x256 tiles in launch_typed's partial-round rule.  Re-swept with the eight-wave 128x128 tile (generation
// latency at B = 4 ... 64, one box): 0.45 loses 10-14 % from B = 8 on, 0.52 1-2 %, 0.70 / 0.80 within 1 % of 0.61 - kept.
#ifndef CVAR_GEMM_C128
#define CVAR_GEMM_C128 0.61
#endif
// the 256x192 tile in the partial-round rule (launch_typed): on / off, and the cost of one of its rounds against a 256x256 round
#ifndef CVAR_GEMM_T192
#define CVAR_GEMM_T192 1
#endif
#ifndef CVAR_GEMM_C192
#define CVAR_GEMM_C192 0.88
#endif
    // M16: the K loop runs on v_mfma_f32_16x16x32_bf16 (two per 32x32x16's worth of flops, 16 cycles each).  Same fragment bytes out of LDS,
    // but an accumulator register is read and written once per 32 k instead of once per 16: the chip is POWER-limited under this kernel
    // (all-zero operands run the identical instruction stream 30 % faster, profiles/r03_gemm_power.txt) and the narrower tile moves less
    // accumulator state per flop.  W fragment first (as TRANS): lane & 15 = output row, the 4 registers = 4 consecutive output columns.
    constexpr bool M16 = (CVAR_GEMM_M16 != 0) && ES == 2 && FRAG_PIPE && TRANS && !CONV;
    constexpr int MI16 = M16 ? SUB_M / 16 : 1, NJ16 = M16 ? SUB_N / 16 : 1;
    f32x4_t acc4[MI16][NJ16];
#pragma unroll
    for (int i = 0; i < MI16; ++i)
#pragma unroll
        for (int j = 0; j < NJ16; ++j) acc4[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int lrow = lane & 31, hi = lane >> 5, sw = (lane >> 1) & 7;
    const int nk_all = (p.K + KT - 1) / KT;
    const int kt_lo = p.split_tiles > 0 ? (int)blockIdx.y * p.split_tiles : 0;
    const int nk = p.split_tiles > 0 ? min(nk_all, kt_lo + p.split_tiles) : nk_all;

    // Pipeline: NSTAGE LDS stages, PF = NSTAGE-1 tiles in flight.  The DMA pieces of tile kt+PF are spread over the four
    // k-steps of tile kt and issued right behind that k-step's fragment reads, so their issue cost overlaps the MFMAs already
    // queued on the matrix pipe.  One barrier per K tile: passing it means (a) every wave's pieces of tile kt have landed
    // (counted vmcnt before the barrier) and (b) every wave has finished reading tile kt-1, whose stage is the one the
    // pieces issued in this iteration overwrite.
    constexpr int PF = NSTAGE - 1;
    // XB (round 4; the schedule hipBLASLt's MT256x256x64 kernel runs, profiles/r04_gemm_isa_vs_hipblaslt.txt): the K-tile boundary is software-
    // pipelined ACROSS the barrier.  Before, every K tile started with  vmcnt(0) - barrier - ten fragment reads - lgkmcnt(0)  in front of its
    // first MFMA: the matrix pipe idled for the LDS latency of a whole fragment set plus the barrier skew, once per 64 k (s_memtime: 2 380 +
    // 160 cycles per tile against 2 048 of MFMA issue).  Now the barrier of tile kt sits in front of its last two A-fragment rows (Pm): by then a
    // wave has issued every LDS read of tile kt, so passing it means (a) everybody's DMA pieces of tile kt+1 have landed (vmcnt(0) before it) and
    // (b) nobody reads tile kt's stage any more.  Behind it, under the tile's remaining MFMAs, the wave reads the FIRST fragments of tile kt+1 out
    // of the other stage and starts the DMA of tile kt+2 into the stage just released (one piece per two MFMAs, running over into the next
    // tile's first MFMAs), so the MFMA stream continues through the loop edge.  Two LDS stages, two tiles in flight, one barrier per tile.
    constexpr bool XB = M16 && NSTAGE == 2 && (CVAR_GEMM_XB != 0);
    constexpr int RING = XB ? 4 : 3;                              // A-fragment ring: 2 MI16 % RING == 0 keeps the slot numbering across tiles
    constexpr int NM16 = MI16 * NJ16;                             // MFMAs per 32-deep k-step
    constexpr int XB_PM = (2 * MI16 - 2) * NJ16;                  // first MFMA behind the barrier
    static_assert(!XB || (2 * MI16) % RING == 0, "fragment ring must divide the tile");
    // piece t of the next-next tile is issued behind MFMA XB_PM + 1 + 2 t of the tile (positions past the tile's end wrap into the next tile)
    auto xb_pos = [](int t) { return (XB_PM + 1 + 2 * t) % (2 * NM16); };
    auto xb_same_iter = [](int t) { return XB_PM + 1 + 2 * t < 2 * NM16; };
    constexpr int XB_N2 = NL < NJ16 ? NL : NJ16;                  // pieces with xb_same_iter: 2 NM16 - XB_PM = 2 NJ16 MFMAs lie behind the barrier
    bf16x8_t a3[RING], bw[2][M16 ? NJ16 : 1];
    int cur = 0;                         // LDS stage of the K tile being computed
    if constexpr (XB) {
#pragma unroll
        for (int idx = 0; idx < NL; ++idx) issue_one(kt_lo, cur, idx);
#pragma unroll
        for (int t = 0; t < NL; ++t)
            if (xb_same_iter(t)) issue_one(min(kt_lo + 1, nk - 1), cur ^ 1, t);
        // the pieces of the first tile have landed when at most the XB_N2 later ones are outstanding (vmcnt retires in order)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XB_N2) : "memory");
        __builtin_amdgcn_s_barrier();
        {
            const int l15 = lane & 15, kq = lane >> 4, sw16 = (l15 >> 1) & 7;
            const char* A16 = smem + cur * STAGE + (wm * SUB_M + l15) * 128;
            const char* B16 = smem + cur * STAGE + BM * 128 + (wn * SUB_N + l15) * 128;
#pragma unroll
            for (int j = 0; j < NJ16; ++j) bw[0][j] = *(const bf16x8_t*)(B16 + j * 16 * 128 + ((kq ^ sw16) * 16));
            a3[0] = *(const bf16x8_t*)(A16 + ((kq ^ sw16) * 16));
            a3[1] = *(const bf16x8_t*)(A16 + 16 * 128 + ((kq ^ sw16) * 16));
        }
    } else {
#pragma unroll
        for (int t = 0; t < PF; ++t) {
            if (FAST && CONV) conv_next();
#pragma unroll
            for (int idx = 0; idx < NL; ++idx) issue_one(min(kt_lo + t, nk - 1), t, idx);
        }
    }
#ifdef CVAR_GEMM_TIMING
    unsigned long long dbg_comp = 0, dbg_vm = 0, dbg_bar = 0, dbg_last = 0, dbg_ew = 0;
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long dbg_rt_loop = __builtin_amdgcn_s_memrealtime();
#endif
    for (int kt = kt_lo; kt < nk; ++kt) {
        const int ktn = min(kt + PF, nk - 1);
#ifdef CVAR_GEMM_TIMING
        const unsigned long long tq0 = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (!XB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * NL) : "memory");   // uniform: dead pieces keep the count regular
#ifdef CVAR_GEMM_TIMING
        const unsigned long long tq1 = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (!XB) __builtin_amdgcn_s_barrier();
#ifdef CVAR_GEMM_TIMING
        const unsigned long long tq2 = __builtin_amdgcn_s_memtime();
        if (kt > kt_lo) dbg_comp += tq0 - dbg_last;
        dbg_vm += tq1 - tq0; dbg_bar += tq2 - tq1; dbg_last = tq2;
#endif
        const int nxt = (cur + PF) % NSTAGE;
        if (FAST && CONV) conv_next();
        const char* As = smem + cur * STAGE + (wm * SUB_M + lrow) * 128;
        const char* Bs = smem + cur * STAGE + BM * 128 + (wn * SUB_N + lrow) * 128;
        if constexpr (M16) {
            typedef __attribute__((ext_vector_type(8))) __bf16 bfv8;
            const int l15 = lane & 15, kq = lane >> 4, sw16 = (l15 >> 1) & 7;
            const char* A16 = smem + cur * STAGE + (wm * SUB_M + l15) * 128;
            const char* B16 = smem + cur * STAGE + BM * 128 + (wn * SUB_N + l15) * 128;
            // registers: a ring of RING A fragments (one feeds NJ16 MFMAs = 64 cycles, the read two ahead has 128 cycles to arrive) and the W
            // fragments of both 32-deep k-steps (the second set is read during the first step)
            auto rd_a = [&](int s32, int i) { a3[(s32 * MI16 + i) % RING] = *(const bf16x8_t*)(A16 + i * 16 * 128 + (((4 * s32 + kq) ^ sw16) * 16)); };   // ring slot = running fragment number % RING
            auto rd_b = [&](int s32, int j) { bw[s32][j] = *(const bf16x8_t*)(B16 + j * 16 * 128 + (((4 * s32 + kq) ^ sw16) * 16)); };
            // XB: the first fragments of the NEXT tile, out of the other stage, behind this tile's barrier
            const char* A16n = smem + nxt * STAGE + (wm * SUB_M + l15) * 128;
            const char* B16n = smem + nxt * STAGE + BM * 128 + (wn * SUB_N + l15) * 128;
            auto rd_a_next = [&](int i) { a3[i % RING] = *(const bf16x8_t*)(A16n + i * 16 * 128 + ((kq ^ sw16) * 16)); };
            auto rd_b_next = [&](int j) { bw[0][j] = *(const bf16x8_t*)(B16n + j * 16 * 128 + ((kq ^ sw16) * 16)); };
            if constexpr (!XB) {
#pragma unroll
                for (int j = 0; j < NJ16; ++j) rd_b(0, j);
                rd_a(0, 0); rd_a(0, 1);
            }
#pragma unroll
            for (int s32 = 0; s32 < 2; ++s32) {
#pragma unroll
                for (int i = 0; i < MI16; ++i) {
#pragma unroll
                    for (int j = 0; j < NJ16; ++j) {
                        acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfv8, bw[s32][j]), __builtin_bit_cast(bfv8, a3[(s32 * MI16 + i) % RING]), acc4[i][j], 0, 0, 0);
                        const int m = (s32 * MI16 + i) * NJ16 + j;          // MFMA index inside the K tile, 0 .. 2 NM16 - 1
                        if (j == 0) {                                        // A fragment two ahead (the slot freed by fragment i - 1)
                            const int in = i + 2;
                            if (in < MI16) rd_a(s32, in);
                            else if (s32 == 0) rd_a(1, in - MI16);
                            else if (XB) rd_a_next(in - MI16);               // fragments 0, 1 of the next tile (m >= XB_PM: behind the barrier)
                        }
                        if constexpr (NJ16 <= MI16) {
                            if (s32 == 0 && j == 2 && i >= MI16 - NJ16) rd_b(1, i - (MI16 - NJ16));     // next k-step's W fragments, one per A fragment
                        } else {                                                                        // more W than A fragments (64x96 wave tile): up to two per A fragment
                            static_assert(NJ16 <= 2 * MI16 && NJ16 >= 5, "W fragment schedule");
                            if (s32 == 0 && j == 2) rd_b(1, i);
                            if (s32 == 0 && j == 4 && MI16 + i < NJ16) rd_b(1, MI16 + i);
                        }
                        if constexpr (XB) {
                            // W fragments of the next tile's first k-step: one per MFMA behind the barrier (bw[0] is dead since k-step 0 ended)
                            if (m >= XB_PM && j != 0) {
                                const int c = (m - XB_PM) - (m - XB_PM) / NJ16 - 1;          // running count over the j != 0 positions
                                if (c < NJ16) rd_b_next(c);
                            }
#pragma unroll
                            for (int t = 0; t < NL; ++t)
                                if (xb_pos(t) == m) {
                                    // (indices past the last tile are dead re-reads of the last tile)
                                    if (xb_same_iter(t)) issue_one(min(kt + 2, nk - 1), cur, t);     // tile kt+2 into the stage this tile releases
                                    else issue_one(min(kt + 1, nk - 1), nxt, t);                      // the rest of the group started behind the previous barrier
                                }
                            if (m == XB_PM - 1) {
                                // every LDS read of this tile has been issued: wait for them and for this wave's pieces of tile kt+1, then meet
                                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                                __builtin_amdgcn_s_barrier();
                            }
                        } else {
                            // DMA pieces of the next tile behind every (NM16 / NL)-th MFMA of the first quarter (two waves per SIMD) / half of the tile
                            constexpr int SPAN16 = NW >= 8 ? NM16 / 2 : NM16;
#pragma unroll
                            for (int t = 0; t < NL; ++t)
                                if (max(((t + 1) * SPAN16) / NL - 1, 0) == m) issue_one(ktn, nxt, t);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        } else if constexpr (ES == 2 && FRAG_PIPE) {
            // one wave per SIMD (accumulators in AGPRs): nothing else hides the ds_read -> MFMA latency, so the fragments of
            // k-step ks+1 are fetched into a second register set before the MFMAs of k-step ks are issued
            bf16x8_t a[2][MI], b[2][NJ];
            auto read_frag = [&](int buf, int ks1, int r) {      // r < MI: A fragment r, else W fragment r - MI
                const int c = ((2 * ks1 + hi) ^ sw) * 16;
                if (r < MI) a[buf][r] = *(const bf16x8_t*)(As + r * 32 * 128 + c);
                else b[buf][r - MI] = *(const bf16x8_t*)(Bs + (r - MI) * 32 * 128 + c);
            };
#pragma unroll
            for (int r = 0; r < MI + NJ; ++r) read_frag(0, 0, r);
            // The issue order is written out by hand and pinned with sched_barrier(0): per MFMA at most one fragment read
            // (every second MFMA) and one DMA piece with its address math (every fourth), so the matrix pipe never waits
            // behind a bunch of LDS / DMA issues.
            constexpr int NM = MI * NJ, NR = MI + NJ;           // MFMAs / fragment reads per k-step
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int q = 0; q < NM; ++q) {
                    const int i = q / NJ, j = q % NJ;
                    acc[i][j] = TRANS ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks & 1][j], a[ks & 1][i], acc[i][j], 0, 0, 0)
                                      : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1][i], b[ks & 1][j], acc[i][j], 0, 0, 0);
                    // reads of k-step ks+1: read r goes behind MFMA floor(r * NM / NR) (evenly spread, all done before the last MFMA)
                    if (ks < 3) {
#pragma unroll
                        for (int r = 0; r < NR; ++r)
                            if ((r * NM) / NR == q) read_frag((ks + 1) & 1, ks + 1, r);
                    }
                    // DMA pieces of the next tile: piece t goes behind MFMA number ((t+1) * 2NM) / NL - 1 of the tile, i.e. all
                    // inside k-steps 0 and 1 (so they have k-steps 2, 3 to land), evenly spread
                    if (DMA_EARLY == 1) {
#pragma unroll
                        for (int t = 0; t < NL; ++t)
                            if (max(((t + 1) * SPAN_NUM * NM) / (SPAN_DEN * NL) - 1, 0) == ks * NM + q) issue_one(ktn, nxt, t);
                    } else {
#pragma unroll
                        for (int t = 0; t < NL; ++t)
                            if (max(((t + 1) * 4 * NM) / NL - 1, 0) == ks * NM + q) issue_one(ktn, nxt, t);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = ((2 * ks + hi) ^ sw) * 16;
            if constexpr (ES == 2) {
                bf16x8_t a[MI], b[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = *(const bf16x8_t*)(As + i * 32 * 128 + c);
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j] = *(const bf16x8_t*)(Bs + j * 32 * 128 + c);
#pragma unroll
                for (int idx = ks * NL / 4; idx < (ks + 1) * NL / 4; ++idx) issue_one(ktn, nxt, idx);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = TRANS ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0)
                                          : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            } else {
                f32x4_t a[MI], b[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = *(const f32x4_t*)(As + i * 32 * 128 + c);
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j] = *(const f32x4_t*)(Bs + j * 32 * 128 + c);
#pragma unroll
                for (int idx = ks * NL / 4; idx < (ks + 1) * NL / 4; ++idx) issue_one(ktn, nxt, idx);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            acc[i][j] = TRANS ? __builtin_amdgcn_mfma_f32_32x32x2f32(b[j][e], a[i][e], acc[i][j], 0, 0, 0)
                                              : __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
            }
        }
        if constexpr (!XB) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // XB: the next tile's first fragments stay in flight over the loop edge
        cur = (cur + 1) % NSTAGE;
    }
    if constexpr (XB) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // dead pieces of the last tiles must not land in the epilogue's staging rows
#ifdef CVAR_GEMM_TIMING
    const unsigned long long dbg_t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long dbg_rt_loop_end = __builtin_amdgcn_s_memrealtime();
#endif
    __syncthreads();
    // ---- epilogue: accumulators -> LDS (per-wave region, 32 rows at a time) -> row-major 8-wide vectors, so that
    // bias / gate / residual loads and the C stores are 16-byte and coalesced (128-B rows per 8 lanes).
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    constexpr int EROW = SUB_N + 4;                       // fp32 row stride of the staging region (16-B aligned)
    constexpr int LPR = SUB_N / 8;                        // lanes per output row
    constexpr int RPP = 64 / LPR;                         // rows per pass (lanes >= RPP*LPR idle when LPR does not divide 64)
    constexpr int NPASS = (16 + RPP - 1) / RPP;
    // The MFMAs are issued with swapped operands (W fragment first), so a 32x32 accumulator block is held TRANSPOSED: lane & 31
    // is the output row, register r the column (r&3) + 8*(r>>2) + 4*(lane>>5) - four consecutive columns per register quad.
    // Staging a block is then 4 ds_write_b128 per 32 columns instead of 16 ds_write_b32 (LDS stores are the narrow port:
    // 64-85 B/clk); rows of EROW = SUB_N + 4 floats keep the 8-lane store groups on distinct banks.
    constexpr bool FULL32 = TRANS && !M16;                // 16x16 blocks (M16) are staged one block row = 16 output rows at a time
    constexpr int SROWS = FULL32 ? 32 : 16;
    static_assert(NW * SROWS * EROW * 4 <= NSTAGE * STAGE, "epilogue staging must fit the pipeline LDS");
#ifndef CVAR_GEMM_RPF
#define CVAR_GEMM_RPF 1
#endif
#ifndef CVAR_RPF_LD_AUX
#define CVAR_RPF_LD_AUX 2      // cache policy of the residual DMA / of the fp32 stores of the RPF epilogue (gfx940+: 2 = nt): x is read once and written once per call
#endif
#ifndef CVAR_GEMM_ST_NT
#define CVAR_GEMM_ST_NT 1      // the specialised epilogues of the plain GEMM tiles (256-row tiles: M >= 2048) store non-temporally - streaming outputs no longer evict the operand panels from the L2s (profiles/r06_nt_policy_ab.txt)
#endif
#ifndef CVAR_RPF_ST_AUX
#define CVAR_RPF_ST_AUX 2
#endif
    // RPF tile: the eight-wave 256x256 bf16 GEMM (16-row staging, 64 columns per wave = 8 lanes per row, two 8-row passes per half-pass); its launch allocates 32 KB
    // behind the pipeline stages (launch_cfg) - see the RPF epilogue below
    constexpr bool RPF_TILE = (CVAR_GEMM_RPF != 0) && M16 && !CONV && FAST && ES == 2 && BM == 256 && BN == 256 && NW == 8 && SUB_N == 64 && SUB_M == 128 && NSTAGE == 2;
    constexpr int RPF_RING_BASE = (NW * SROWS * EROW * 4 + 1023) / 1024 * 1024;
    static_assert(!RPF_TILE || RPF_RING_BASE + NW * 8192 <= NSTAGE * STAGE, "RPF ring slots 1-2 must fit the released pipeline stages");
    float* stg = (float*)smem + wave * (SROWS * EROW);
    // rows of block i land in the staging region: the whole transposed block at once (4 x b128 per 32 columns), or - tiles whose
    // pipeline LDS is too small for 32 staged rows per wave keep the plain MFMA layout (lane = column) - 16 rows as b32 stores
    auto stage_block = [&](int i, int half) {
        if constexpr (M16) {                  // 16x16 blocks: lane & 15 = row inside the block, lane >> 4 selects 4 of its 16 columns
            const int l15 = lane & 15, cq = lane >> 4;
#pragma unroll
            for (int j = 0; j < NJ16; ++j) *(f32x4_t*)(stg + l15 * EROW + j * 16 + 4 * cq) = acc4[2 * i + half][j];
        } else if (FULL32) {
            if (half != 0) return;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4_t q = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    *(f32x4_t*)(stg + lrow * EROW + j * 32 + 8 * g + 4 * hi) = q;
                }
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8)
                    stg[((r8 & 3) + 8 * (r8 >> 2) + 4 * hi) * EROW + j * 32 + lrow] = acc[i][j][8 * half + r8];
        }
    };
    const int srow_half = FULL32 ? 16 : 0;             // staging row of output row 16*half + rr is rr + srow_half * half
    char* Cb = (char*)p.C;
    const long cz = zb * p.strideC + (long)blockIdx.y * p.split_stride, rz = zb * p.strideR;
    const int erow = lane / LPR, ecol = (lane % LPR) * 8;
    const bool vec_ok = ((p.N & 7) == 0) && ((p.ldc & 7) == 0) && ((p.strideC & 7) == 0) && (((uintptr_t)p.C & 15) == 0) &&
                        (!p.residual || (((p.ldr & 7) == 0) && ((p.strideR & 7) == 0) && (((uintptr_t)p.residual & 15) == 0))) &&
                        (!p.gate || (((p.ldg & 3) == 0) && (((uintptr_t)p.gate & 15) == 0))) &&
                        (!p.bias || (((uintptr_t)p.bias & 15) == 0)) && (!p.C2 || (((uintptr_t)p.C2 & 15) == 0)) &&
                        (p.split_n <= 0 || (((p.split_n & 7) == 0) && ((p.ld_split & 7) == 0) && (((uintptr_t)p.Cs & 15) == 0)));
    // per-lane constants of the row-major phase: the column group never changes, so bias is loaded once
    const int n = n0 + wn * SUB_N + ecol;
    const bool lane_on = (erow < RPP) && (n < p.N);
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = 0.0f;
    if (vec_ok && p.bias && lane_on) {
        const f32x4_t b0 = *(const f32x4_t*)(p.bias + n), b1 = *(const f32x4_t*)(p.bias + n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bias8[e] = b0[e]; bias8[4 + e] = b1[e]; }
    }
    // Pin the bias registers as "arrived" here, on a path every lane takes: otherwise the load stays pending across the
    // predicated passes below and the compiler re-emits s_waitcnt vmcnt(0) in each of them, which also waits for the
    // previous pass's stores (vmcnt counts stores on gfx9) and serialises the whole epilogue on store latency.
    asm volatile("" : "+v"(bias8[0]), "+v"(bias8[1]), "+v"(bias8[2]), "+v"(bias8[3]), "+v"(bias8[4]), "+v"(bias8[5]), "+v"(bias8[6]), "+v"(bias8[7]));
    // Specialised epilogues for the combinations the hot path uses on the large tiles: every flag is a compile-time constant,
    // pointers are hoisted to one per-lane base plus a wave-uniform row offset per pass, so a pass is ~20 vector instructions
    // instead of the generic code's flag tests and 64-bit address rebuilds.
    constexpr bool SPEC = (BM * BN >= 128 * 160) && ES == 2;      // bf16 kernels only: the fp32 parity mode is not a throughput path, and every variant costs compile time
    bool done = false;
    if constexpr (SPEC) {
        auto run = [&](auto OUTBF, auto ACT, auto GATE, auto RES, auto REMAP) __attribute__((always_inline)) {
            constexpr bool out_bf = decltype(OUTBF)::value, gate = decltype(GATE)::value, remap = decltype(REMAP)::value;
            constexpr int act = decltype(ACT)::value, res = decltype(RES)::value;          // res: 0 none, 1 fp32, 2 bf16
            constexpr int OES = out_bf ? 2 : 4, RES_ES = res == 2 ? 2 : 4;
#if CVAR_GEMM_RPF
            // ---- RPF (round 6): the fp32 gate + residual read-modify-write of a FULL 256x256 tile (proj / fc2: x += gate * f, basic_var.py:208-209) with the residual
            // rows prefetched THREE half-passes ahead by LDS DMA.  The register form keeps one half-pass (32 KB per CU) of residual reads in flight - a CU then moves
            // its 512 KB at ~25 B/clk, latency-bound (28 % of proj, 9-10 % of fc2: profiles/r05_gemm_insitu_b512.txt), and deeper register prefetch spills
            // (256 VGPRs, profiles/r03_gemm_rejected_ab.txt).  Here a wave owns a ring of three 4 KB slots (16 rows x 64 fp32 columns each: slot 0 in the 32 KB behind
            // the pipeline stages, slots 1-2 in the released stages behind the staging rows): `buffer_load ... lds` costs no registers, the pieces of half-pass h + 3
            // are issued when half-pass h has consumed its slot, and one counted vmcnt in front of a half-pass (loads, stores and DMA retire in issue order on gfx9;
            // every count below is static because nothing in this path is predicated) waits for exactly its own four pieces.  Stores and gate loads go through
            // buffer instructions on scalar bases + 32-bit offsets.  Same arithmetic in the same order as the register form: bit-identical results.
            // Taken by calls whose output streams (p.nt, set by cvar_gemm for outputs of >= 128 MB: the ring's DMA and the stores then carry the non-temporal bit - x is
            // read once and written once per call) and by tile_cfg 2 (tests, A/B runs); smaller calls, partial tiles and tile_cfg 28 run the register form.
            if constexpr (RPF_TILE && gate && res == 1 && !out_bf && !remap && act == CVAR_ACT_NONE) {
                const bool rpf_ok = vec_ok && m0 + BM <= p.M && n0 + BN <= p.N && !p.C2 && !p.gate_scale && p.tile_cfg != 28 && (p.nt || p.tile_cfg == 2) &&
                                    ((long)(p.M / max(p.gate_rows, 1) + 1) * p.ldg * 4 < 0x7fffffffL) && (long)SUB_M * p.ldr * 4 < 0x7fffffffL && (long)SUB_M * p.ldc * 4 < 0x7fffffffL;
                if (rpf_ok) {
                    const int mw = m0 + wm * SUB_M, nw = n0 + wn * SUB_N;
                    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)p.residual + rz + (long)mw * p.ldr + nw), 0, 0x7fffffff, 0x00020000);
                    const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((float*)p.C + cz + (long)mw * p.ldc + nw), 0, 0x7fffffff, 0x00020000);
                    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.gate + nw), 0, 0x7fffffff, 0x00020000);
                    const int ldr4 = (int)p.ldr * 4, ldc4 = (int)p.ldc * 4, ldg4 = (int)p.ldg * 4;
                    const int dma_lane = (lane >> 4) * ldr4 + (lane & 15) * 16;            // a 1 KB piece = 4 rows x 256 B
                    const int st_lane = erow * ldc4 + ecol * 4;
                    char* const slot0 = smem + NSTAGE * STAGE + wave * 4096;
                    char* const slot12 = smem + RPF_RING_BASE + wave * 8192;
                    auto slot = [&](int h) -> char* { return (h % 3) == 0 ? slot0 : slot12 + ((h % 3) - 1) * 4096; };
                    auto issue_res = [&](int h) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rsrc, (lptr_t)(slot(h) + q * 1024), 16, dma_lane, (h * 16 + q * 4) * ldr4, 0, CVAR_RPF_LD_AUX);
                    };
                    f32x4_t gq2[2][2][2];
                    auto fetch_gate = [&](int h) {
#pragma unroll
                        for (int ps = 0; ps < 2; ++ps) {
                            const int m = mw + h * 16 + ps * 8 + erow;
                            const int goff = fast_div(m, p.gate_magic, p.gate_shift) * ldg4 + ecol * 4;
                            gq2[h & 1][ps][0] = __builtin_bit_cast(f32x4_t, (v4i_t)__builtin_amdgcn_raw_buffer_load_b128(g_rsrc, goff, 0, 0));
                            gq2[h & 1][ps][1] = __builtin_bit_cast(f32x4_t, (v4i_t)__builtin_amdgcn_raw_buffer_load_b128(g_rsrc, goff + 16, 0, 0));
                        }
                    };
                    // issue order: G0 D0 D1 D2 | per half-pass h: wait D_h, stage + combine, G_{h+1}, S_h, D_{h+3}.  G = 4 loads, S = 4 stores, D = 4 pieces.
                    fetch_gate(0);
                    issue_res(0); issue_res(1); issue_res(2);
                    const float* stg_r = stg + erow * EROW + ecol;
                    typedef const __attribute__((address_space(3))) char* lcptr_t;
                    const lcptr_t sl0 = (lcptr_t)(slot0 + erow * 256 + ecol * 4), sl12 = (lcptr_t)(slot12 + erow * 256 + ecol * 4);     // this lane's 32 bytes of a ring row
#pragma clang loop unroll(full)
                    for (int h = 0; h < 2 * MI; ++h) {
                        // operations issued after D_h: h == 0: D1 D2 (8); h == 1: D2 G1 S0 D3 (16); h >= 2: [G S D] x 2 (24), less the D that no longer exist near the end
                        constexpr int NH = 2 * MI;
#ifndef CVAR_RPF_WAIT
#define CVAR_RPF_WAIT 0
#endif
                        // ops issued after D_h, by kind: younger DMA pieces, stores, gate loads (see the issue order above)
                        const int nd = h == 0 ? 8 : (h == 1 ? 4 + (3 < NH ? 4 : 0) : (h + 1 < NH ? 4 : 0) + (h + 2 < NH ? 4 : 0));
                        const int ns = h == 0 ? 0 : (h == 1 ? 4 : 8);
                        const int ng = h == 0 ? 0 : (h == 1 ? 4 : 8);
                        const int after = CVAR_RPF_WAIT == 1 ? nd + ns : (CVAR_RPF_WAIT == 2 ? nd : nd + ns + ng);
                        switch (after) {
                            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                            case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
                            case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
                            case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
                            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                        }
                        if (CVAR_RPF_WAIT == 3) __builtin_amdgcn_s_sleep(4);
                        stage_block(h >> 1, h & 1);
                        float v[2][8];
                        const lcptr_t sl = ((h % 3) == 0 ? sl0 : sl12 + ((h % 3) - 1) * 4096);
#pragma unroll
                        for (int ps = 0; ps < 2; ++ps) {
                            const f32x4_t a0 = *(const f32x4_t*)(stg_r + (ps * 8) * EROW), a1 = *(const f32x4_t*)(stg_r + (ps * 8) * EROW + 4);
                            const f32x4_t r0 = *(const __attribute__((address_space(3))) f32x4_t*)(sl + ps * 2048), r1 = *(const __attribute__((address_space(3))) f32x4_t*)(sl + ps * 2048 + 16);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[ps][e] = a0[e] * p.alpha + bias8[e]; v[ps][4 + e] = a1[e] * p.alpha + bias8[4 + e];
                                v[ps][e] *= gq2[h & 1][ps][0][e]; v[ps][4 + e] *= gq2[h & 1][ps][1][e];
                                v[ps][e] += r0[e]; v[ps][4 + e] += r1[e];
                            }
                        }
                        if (h + 1 < NH) fetch_gate(h + 1);
#pragma unroll
                        for (int ps = 0; ps < 2; ++ps) {
                            const f32x4_t o0 = {v[ps][0], v[ps][1], v[ps][2], v[ps][3]}, o1 = {v[ps][4], v[ps][5], v[ps][6], v[ps][7]};
                            // the row offset rides in the VECTOR offset, soffset stays 0: with an SGPR soffset the compiler assumes a 128-bit buffer store has no
                            // store-data hazard and lets the next VALU instruction overwrite the data registers (GCNHazardRecognizer: "this hazard only exists if the
                            // instruction is not using a register in the soffset field") - on gfx950 a quarter of the lanes then stored the NEW register contents
                            // (last half-pass of every wave tile, nondeterministic; found by the bit-identity test of this path, profiles/r06_rpf_ab.txt)
                            const int st_off = st_lane + (h * 16 + ps * 8) * ldc4;
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i_t, o0), c_rsrc, st_off, 0, CVAR_RPF_ST_AUX);
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i_t, o1), c_rsrc, st_off + 16, 0, CVAR_RPF_ST_AUX);
                        }
                        if (h + 3 < NH) issue_res(h + 3);
                    }
                    return;
                }
            }
#endif
            const int mrow = m0 + wm * SUB_M + erow;
            const float* stg_r = stg + erow * EROW + ecol;
            char* c_lane = Cb + (cz + n) * OES;
            bool in_split = false;
            if constexpr (remap) {               // qkv GEMM of inference: the q columns go to their own buffer, k | v to the arena
                if (p.split_n > 0) {
                    in_split = n < p.split_n;
                    c_lane = in_split ? (char*)p.Cs + (long)n * OES : Cb + (cz + n - p.split_n) * OES;
                }
            }
            const char* r_lane = res ? (const char*)p.residual + (rz + n) * RES_ES : nullptr;
            const float* g_lane = gate ? p.gate + n : nullptr;
            // gate / residual operands are requested one half-pass AHEAD, i.e. before the previous half-pass's stores are issued:
            // vmcnt retires in order, so a load queued behind stores would make its consumer wait for the store latency too
            f32x4_t gq[2][gate ? NPASS : 1][2], rq[2][res == 1 ? NPASS : 1][2];
            bf16x8_t rb[2][res == 2 ? NPASS : 1];
            bf16x8_t xb[2][act == CVAR_ACT_GELU_GRAD ? NPASS : 1];        // gelu' operand (bf16 variants only)
            const char* x_lane = act == CVAR_ACT_GELU_GRAD ? (const char*)p.aux + (cz + n) * 2 : nullptr;
            char* c2_lane = p.C2 ? (char*)p.C2 + (cz + n) * ES : nullptr;      // operand dtype
            auto fetch_operands = [&](int ih) {
                const int i = ih >> 1, half = ih & 1, bsel = ih & 1;
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const int m = mrow + i * 32 + 16 * half + ps * RPP;
                    if (lane_on && m < p.M && ((16 % RPP == 0) || ps * RPP + erow < 16)) {
                        if constexpr (gate) {
                            const int grow_i = fast_div(m, p.gate_magic, p.gate_shift);
                            const float* gp = g_lane + (long)grow_i * p.ldg;
                            gq[bsel][ps][0] = *(const f32x4_t*)gp; gq[bsel][ps][1] = *(const f32x4_t*)(gp + 4);
                            if (p.gate_scale) {
                                const float gs = p.gate_scale[grow_i];
#pragma unroll
                                for (int e = 0; e < 4; ++e) { gq[bsel][ps][0][e] *= gs; gq[bsel][ps][1][e] *= gs; }
                            }
                        }
                        if constexpr (res == 1) {
                            const float* rp = (const float*)(r_lane + (long)m * p.ldr * 4);
                            rq[bsel][ps][0] = *(const f32x4_t*)rp; rq[bsel][ps][1] = *(const f32x4_t*)(rp + 4);
                        }
                        if constexpr (res == 2) rb[bsel][ps] = *(const bf16x8_t*)(r_lane + (long)m * p.ldr * 2);
                        if constexpr (act == CVAR_ACT_GELU_GRAD) xb[bsel][ps] = *(const bf16x8_t*)(x_lane + (long)m * p.ldc * 2);
                    }
                }
            };
            if constexpr (gate || res != 0 || act == CVAR_ACT_GELU_GRAD) fetch_operands(0);
#pragma clang loop unroll(full)
            for (int ih = 0; ih < 2 * MI; ++ih) {
                const int i = ih >> 1, half = ih & 1, bsel = ih & 1;
                if constexpr (gate || res != 0 || act == CVAR_ACT_GELU_GRAD) { if (ih + 1 < 2 * MI) fetch_operands(ih + 1); }
                stage_block(i, half);
                // 16x16-block staging with idle lanes in the row-major phase (LPR does not divide 64, e.g. the 64x96 wave tile): an idle lane never reads the staging
                // region itself, so from ITS point of view every staging store but the last is dead and the compiler predicates them on lane_on - but the
                // other lanes of the wave read those bytes.  A wavefront-scope release fence (the staging stores must be performed before anything behind it) plus a
                // wave barrier keep them - no instruction either; only these instantiations (the documented primitives instead of an empty asm clobber).
                if constexpr (M16 && (64 % LPR) != 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const int roff = i * 32 + 16 * half + ps * RPP;          // wave-uniform, known at compile time
                    const int m = mrow + roff;
                    if (lane_on && m < p.M && ((16 % RPP == 0) || ps * RPP + erow < 16)) {
                        const f32x4_t a0 = *(const f32x4_t*)(stg_r + (ps * RPP + srow_half * half) * EROW);
                        const f32x4_t a1 = *(const f32x4_t*)(stg_r + (ps * RPP + srow_half * half) * EROW + 4);
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] = a0[e] * p.alpha + bias8[e]; v[4 + e] = a1[e] * p.alpha + bias8[4 + e]; }
                        if (c2_lane) {                   // training forward: the branch value the backward needs (fc1 pre-activation, proj / fc2 output)
                            if constexpr (ES == 2) *(bf16x8_t*)(c2_lane + (long)m * p.ldc * ES) = pack_bf16x8(v);
                            else {
                                const f32x4_t q0 = {v[0], v[1], v[2], v[3]}, q1 = {v[4], v[5], v[6], v[7]};
                                *(f32x4_t*)(c2_lane + (long)m * p.ldc * ES) = q0;
                                *(f32x4_t*)(c2_lane + (long)m * p.ldc * ES + 16) = q1;
                            }
                        }
                        if constexpr (act == CVAR_ACT_GELU_TANH) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = (ES == 2) ? gelu_tanh_fast(v[e]) : gelu_tanh_f(v[e]);
                        }
                        if constexpr (act == CVAR_ACT_GELU_GRAD) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] *= gelu_tanh_grad(bf16_to_f32((bf16_t)xb[bsel][ps][e]));
                        }
                        if constexpr (gate) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v[e] *= gq[bsel][ps][0][e]; v[4 + e] *= gq[bsel][ps][1][e]; }
                        }
                        if constexpr (res == 1) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v[e] += rq[bsel][ps][0][e]; v[4 + e] += rq[bsel][ps][1][e]; }
                        } else if constexpr (res == 2) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += bf16_to_f32((bf16_t)rb[bsel][ps][e]);
                        }
                        long orow = m;
                        if constexpr (remap) {
                            const int sq = fast_div(m, p.remap_magic, p.remap_shift);
                            orow = (long)sq * p.remap_L + p.remap_off + (m - sq * p.remap_l);
                        }
                        char* cp = c_lane + orow * p.ldc * OES;
                        if constexpr (remap) {
                            if (in_split) {
                                cp = c_lane + (long)m * p.ld_split * OES;
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] *= p.split_alpha;
                            }
                        }
                        if constexpr (out_bf) {
                            if (CVAR_GEMM_ST_NT != 0 && !CONV && p.nt) __builtin_nontemporal_store(pack_bf16x8(v), (bf16x8_t*)cp);
                            else *(bf16x8_t*)cp = pack_bf16x8(v);
                        } else {
                            const f32x4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                            if (CVAR_GEMM_ST_NT != 0 && !CONV && p.nt) { __builtin_nontemporal_store(o0, (f32x4_t*)cp); __builtin_nontemporal_store(o1, (f32x4_t*)(cp + 16)); }
                            else { *(f32x4_t*)cp = o0; *(f32x4_t*)(cp + 16) = o1; }
                        }
                    }
                }
            }
        };
        using std::integral_constant;
        typedef integral_constant<bool, true> Y; typedef integral_constant<bool, false> NO;
        typedef integral_constant<int, 0> I0; typedef integral_constant<int, 1> I1; typedef integral_constant<int, 2> I2;
        const bool obf = p.out_dtype == CVAR_BF16, rm = p.remap_l > 0;
        if (vec_ok) {
            // each kernel kind only carries the variants its callers use (a variant's registers count against the whole kernel)
            if (!p.gate && !p.residual && p.act == CVAR_ACT_NONE) {
                if (obf && !rm) { run(Y{}, I0{}, NO{}, I0{}, NO{}); done = true; }
                else if (!obf && !rm) { run(NO{}, I0{}, NO{}, I0{}, NO{}); done = true; }
                else if constexpr (!CONV) { if (obf && rm) { run(Y{}, I0{}, NO{}, I0{}, Y{}); done = true; } }
            } else if constexpr (!CONV) {
                if (!p.gate && !p.residual && p.act == CVAR_ACT_GELU_TANH && obf && !rm) {
                    run(Y{}, I1{}, NO{}, I0{}, NO{}); done = true;
                } else if (!p.gate && !p.residual && p.act == CVAR_ACT_GELU_GRAD && obf && ES == 2 && !rm && (((uintptr_t)p.aux & 15) == 0)) {
                    run(Y{}, I2{}, NO{}, I0{}, NO{}); done = true;
                } else if (p.gate && p.residual && p.res_dtype == CVAR_F32 && p.act == CVAR_ACT_NONE && !obf && !rm) {
                    run(NO{}, I0{}, Y{}, I1{}, NO{}); done = true;
                }
            } else {
                if (!p.gate && p.residual && p.res_dtype == CVAR_BF16 && p.act == CVAR_ACT_NONE && obf && !rm) {
                    run(Y{}, I0{}, NO{}, I2{}, NO{}); done = true;
                }
                // (round 6, measured and not kept: an fp32-output + fp32-residual variant for the split-bf16 encoder's conv2 raises the register allocation of EVERY conv tile
                //  of this unit - 256x160 four-wave: 94 -> 142 VGPRs beside 160 AGPRs, two workgroups per CU -> one - so those calls stay on the generic epilogue)
            }
        }
    }
    if (!done) {
    // The staging region is private to the wave and one wave's LDS operations execute in order, so the 2*MI passes need no
    // workgroup barrier between their write and read halves.
#pragma clang loop unroll(full)
    for (int ih = 0; ih < 2 * MI; ++ih) {
        const int i = ih >> 1, half = ih & 1;             // rows 16*half .. 16*half+15 of block i <-> regs 8*half .. 8*half+7
#ifdef CVAR_GEMM_TIMING
        const unsigned long long te0 = __builtin_amdgcn_s_memtime();
#endif
        stage_block(i, half);
        if constexpr (M16 && (64 % LPR) != 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }          // see the specialised loop above
#ifdef CVAR_GEMM_TIMING
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long te1 = __builtin_amdgcn_s_memtime();
        dbg_ew += te1 - te0;
#endif
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int rr = ps * RPP + erow;
            const int m = m0 + wm * SUB_M + i * 32 + 16 * half + rr;
            if (!lane_on || rr >= 16 || m >= p.M) continue;
            float v[8];
            {
                const float* sg = stg;
                const f32x4_t a0 = *(const f32x4_t*)(sg + (rr + srow_half * half) * EROW + ecol);
                const f32x4_t a1 = *(const f32x4_t*)(sg + (rr + srow_half * half) * EROW + ecol + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = a0[e] * p.alpha; v[4 + e] = a1[e] * p.alpha; }
            }
            long orow = m;
            if (p.remap_l > 0) {
                const int sq = fast_div(m, p.remap_magic, p.remap_shift);
                orow = (long)sq * p.remap_L + p.remap_off + (m - sq * p.remap_l);
            }
            const float* grow = p.gate ? p.gate + (long)fast_div(m, p.gate_magic, p.gate_shift) * p.ldg : nullptr;
            void* Cdst = Cb;
            long cbase = cz + orow * p.ldc + n;
            bool scale_split = false;
            if (p.split_n > 0) {
                if (n < p.split_n) { Cdst = p.Cs; cbase = (long)m * p.ld_split + n; scale_split = p.split_alpha != 1.0f; }
                else cbase -= p.split_n;
            }
            if (vec_ok) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bias8[e];
                if (p.C2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) st_any(p.C2, ES == 2 ? CVAR_BF16 : CVAR_F32, cz + (long)m * p.ldc + n + e, v[e]);
                }
                if (scale_split) {                   // the split rides on vector-aligned launches only (cvar_gemm checks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= p.split_alpha;
                }
                if (p.act == CVAR_ACT_GELU_TANH) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (ES == 2) ? gelu_tanh_fast(v[e]) : gelu_tanh_f(v[e]);
                } else if (p.act == CVAR_ACT_GELU_GRAD) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= gelu_tanh_grad(ld_any(p.aux, p.out_dtype, cz + (long)m * p.ldc + n + e));
                }
                if (grow) {
                    const f32x4_t g0 = *(const f32x4_t*)(grow + n), g1 = *(const f32x4_t*)(grow + n + 4);
                    const float gs = p.gate_scale ? p.gate_scale[fast_div(m, p.gate_magic, p.gate_shift)] : 1.0f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] *= g0[e] * gs; v[4 + e] *= g1[e] * gs; }
                }
                if (p.residual) {
                    if (p.res_dtype == CVAR_BF16) {
                        const bf16x8_t rv = *(const bf16x8_t*)((const bf16_t*)p.residual + rz + (long)m * p.ldr + n);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += bf16_to_f32((bf16_t)rv[e]);
                    } else {
                        const float* rp = (const float*)p.residual + rz + (long)m * p.ldr + n;
                        const f32x4_t r0 = *(const f32x4_t*)rp, r1 = *(const f32x4_t*)(rp + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
                    }
                }
                if (p.out_dtype == CVAR_BF16) {
                    *(bf16x8_t*)((bf16_t*)Cdst + cbase) = pack_bf16x8(v);
                } else {
                    float* cp = (float*)Cdst + cbase;
                    const f32x4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                    *(f32x4_t*)cp = o0; *(f32x4_t*)(cp + 4) = o1;
                }
            } else {
                for (int e = 0; e < 8; ++e) {
                    if (n + e >= p.N) break;
                    float x = v[e];
                    if (p.bias) x += p.bias[n + e];
                    if (p.C2) st_any(p.C2, ES == 2 ? CVAR_BF16 : CVAR_F32, cz + (long)m * p.ldc + n + e, x);
                    if (p.act == CVAR_ACT_GELU_TANH) x = gelu_tanh_f(x);
                    else if (p.act == CVAR_ACT_GELU_GRAD) x *= gelu_tanh_grad(ld_any(p.aux, p.out_dtype, cz + (long)m * p.ldc + n + e));
                    if (grow) x *= grow[n + e] * (p.gate_scale ? p.gate_scale[fast_div(m, p.gate_magic, p.gate_shift)] : 1.0f);
                    if (p.residual) x += ld_any(p.residual, p.res_dtype, rz + (long)m * p.ldr + n + e);
                    st_any(Cdst, p.out_dtype, cbase + e, x);
                }
            }
        }
    }
    }   // generic epilogue
#ifdef CVAR_GEMM_TIMING
    if (lane == 0 && wave == 0 && (int)blockIdx.x < CVAR_DBG_WG_MAX && blockIdx.y == 0 && blockIdx.z == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the stores of this wave have been acknowledged
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* o = cvar_gemm_dbg_wg + (size_t)blockIdx.x * 5;
        o[0] = dbg_rt_entry; o[1] = dbg_rt_loop; o[2] = dbg_rt_loop_end; o[3] = __builtin_amdgcn_s_memrealtime(); o[4] = (unsigned long long)hwid | ((unsigned long long)xcc << 32);
    }
    if (lane == 0 && blockIdx.x < 64) {
        const unsigned long long dbg_t2 = __builtin_amdgcn_s_memtime();
        unsigned long long* o = cvar_gemm_dbg + (blockIdx.x * 8 + wave) * 8;
        o[0] = dbg_comp; o[1] = dbg_vm; o[2] = dbg_bar; o[3] = dbg_t1 - dbg_t0; o[4] = dbg_t2 - dbg_t1; o[5] = nk - kt_lo; o[6] = dbg_t0; o[7] = dbg_t2;
    }
    if (lane == 0 && wave == 0) {       // totals over every tile of the launch
        const unsigned long long dbg_t2 = __builtin_amdgcn_s_memtime();
        atomicAdd(&cvar_gemm_dbg_tot[0], dbg_t1 - dbg_t0); atomicAdd(&cvar_gemm_dbg_tot[1], dbg_t2 - dbg_t1); atomicAdd(&cvar_gemm_dbg_tot[2], 1ULL);
        atomicAdd(&cvar_gemm_dbg_tot[6], dbg_ew); atomicAdd(&cvar_gemm_dbg_tot[3], dbg_vm); atomicAdd(&cvar_gemm_dbg_tot[4], dbg_bar); atomicAdd(&cvar_gemm_dbg_tot[5], dbg_comp);
    }
#endif
}

template <typename T, int BM, int BN, int WM, int WN, bool CONV, int NSTAGE = 2, bool FAST = false, bool CUP = false>
__global__ __launch_bounds__(WM * WN * 64) void cvar_gemm_kernel(const GemmParams p) {
    // (round 6, built and measured a loss, no longer here: a workgroup that runs two neighbouring tiles one after the other - profiles/r06_gemm_two_tiles_rejected.txt)
    cvar_gemm_tile<T, BM, BN, WM, WN, CONV, NSTAGE, FAST, CUP>(p, (int)blockIdx.x);
}

// One output quad of a split-K GEMM: the slices' fp32 partials summed in slice order (bit-reproducible), then the complete epilogue of `p`.
__device__ __forceinline__ void splitk_finish_quad(const GemmParams& p, const float* __restrict__ part, int nsplit, int m, int n) {
    f32x4_t v;
    const float* const q0 = part + (long)m * p.N + n;
    const long sstride = (long)p.M * p.N;
    {
        v = *(const f32x4_t*)q0;
        // the loads of the slices are independent: eight in flight at a time (a one-load-one-add loop is a chain of up to 15 memory latencies)
        for (int s0 = 1; s0 < nsplit; s0 += 8) {
            f32x4_t w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < nsplit) w[u] = *(const f32x4_t*)(q0 + (long)(s0 + u) * sstride);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < nsplit) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += w[u][e];
                }
        }
    }
    gemm_epilogue_quad(p, m, n, v);
}

// hipFuncAttributeMaxDynamicSharedMemorySize is sticky per (function, device): set it the first time a kernel is launched on a
// device instead of on every launch (~1200 launches per generation).  `done` is the caller's table for THIS kernel (launch_cfg holds one per kernel it can launch:
// every instantiation has the same function-pointer type, so a table keyed by that type - the form before round 6 - was shared by all of them and only the first
// kernel a process launched ever got the attribute; ROCm does not enforce it, which is why that went unnoticed)
template <typename K>
static void set_max_lds_once(K kfn, size_t lds, unsigned char* done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !done[dev]) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (dev >= 0 && dev < 64) done[dev] = 1;
    }
}

template <typename T, int BM, int BN, int WM, int WN, int NSTAGE = 2, bool CONVFAST = false, bool CUP = false>
static int launch_cfg(const GemmParams& gp, int batch, hipStream_t st) {
    GemmParams p = gp;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    // Group height.  Within a group the A row blocks (GM x BM x K) are re-read once per column tile and the W tile once per group;
    // each XCD's 4 MB L2 holds neither for the long-K / wide-output GEMMs, the refills come out of the Infinity Cache.  Measured on the
    // d24 shapes (profiles/r02_gemm_group_ab.txt): 4 rows instead of 8 is +2.4 % for K = 6144 (fc2) and +3 % for the fp32-output head,
    // neutral or -1 % for the K = 1536 bf16-output GEMMs; 16 loses everywhere.
    if (p.group_m <= 0) p.group_m = (!p.conv && ((long)p.K * (long)sizeof(T) >= 8192 || p.out_dtype == CVAR_F32)) ? 4 : 8;
    if (p.conv) { const int kt_e = 128 / (int)sizeof(T); p.cv_adv = kt_e / p.Cin; p.cv_rem = kt_e % p.Cin; }
    // (+32 KB behind the stages for the eight-wave 256x256 bf16 tile: slot 0 of the RPF epilogue's residual ring; the CU holds one such workgroup either way)
    const size_t lds = NSTAGE * (BM + BN) * 128 + ((sizeof(T) == 2 && BM == 256 && BN == 256 && WM == 2 && WN == 4 && NSTAGE == 2 && !CONVFAST) ? 32768 : 0);
    const int nk_all = (p.K + (128 / (int)sizeof(T)) - 1) / (128 / (int)sizeof(T));
    const int splits = p.split_tiles > 0 ? (nk_all + p.split_tiles - 1) / p.split_tiles : 1;
    const bool plain_fast = !p.conv && p.K % (128 / (int)sizeof(T)) == 0 && (long)(BM - 1) * p.lda * (long)sizeof(T) + (long)p.K * (long)sizeof(T) < (1L << 31) &&
                            (long)(BN - 1) * p.ldw * (long)sizeof(T) + (long)p.K * (long)sizeof(T) < (1L << 31);
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)splits, (unsigned)batch), block(WM * WN * 64);
    // Which kernels a translation unit instantiates: the bf16 conv kernels live in their own unit (gemm_conv.hip), the bf16 GEMM
    // kernels in gemm.hip, everything fp32 in gemm_f32.hip - three compilations that run in parallel.
    constexpr bool WITH_CONV = CVAR_TU_CONV || sizeof(T) == 4, WITH_PLAIN = CVAR_TU_PLAIN || sizeof(T) == 4;
    if constexpr (CONVFAST) {            // this configuration exists for the conv FAST kernel only
        if (!p.conv) return CVAR_EINVAL;
        if constexpr (WITH_CONV) {
            auto kfn = cvar_gemm_kernel<T, BM, BN, WM, WN, true, NSTAGE, true, CUP>;
            static unsigned char lds_set[64] = {0};
            set_max_lds_once(kfn, lds, lds_set);
            hipLaunchKernelGGL(kfn, grid, block, lds, st, p);
            CVAR_CHECK_LAUNCH();
            return CVAR_OK;
        } else return CVAR_EUNSUPPORTED;
    } else if (p.conv) {
        if constexpr (WITH_CONV) {
            auto kfn = cvar_gemm_kernel<T, BM, BN, WM, WN, true, NSTAGE>;
            static unsigned char lds_set[64] = {0};
            set_max_lds_once(kfn, lds, lds_set);
            hipLaunchKernelGGL(kfn, grid, block, lds, st, p);
        } else return CVAR_EUNSUPPORTED;
    } else if constexpr (!WITH_PLAIN) {
        return CVAR_EUNSUPPORTED;
    } else if (plain_fast) {
        auto kfn = cvar_gemm_kernel<T, BM, BN, WM, WN, false, NSTAGE, true>;
        static unsigned char lds_set[64] = {0};
        set_max_lds_once(kfn, lds, lds_set);
        hipLaunchKernelGGL(kfn, grid, block, lds, st, p);
    } else {
        auto kfn = cvar_gemm_kernel<T, BM, BN, WM, WN, false, NSTAGE>;
        static unsigned char lds_set[64] = {0};
        set_max_lds_once(kfn, lds, lds_set);
        hipLaunchKernelGGL(kfn, grid, block, lds, st, p);
    }
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// cvar_gemm_desc::tile_cfg -> the A/B selector of launch_typed: -1 automatic, 0 128x128 tiles only, 1 the 8-wave 256x256 tile, 3 the
// 4-wave 256x256 tile (part of the call, not of the process environment)
static int gemm_cfg_override(const GemmParams& p) {
    return p.tile_cfg == 1 ? 0 : (p.tile_cfg == 2 || p.tile_cfg == 7 || p.tile_cfg == 28) ? 1 : (p.tile_cfg == 3 || p.tile_cfg == 8) ? 3 : p.tile_cfg == 4 ? 4 : -1;
}

template <typename T>
static int launch_typed(const GemmParams& p, int batch, hipStream_t st) {
    // small-M problems (early scales, ada_lin) use a 64-row tile to put more blocks on the chip
    if (p.M <= 64) return launch_cfg<T, 64, 128, 1, 4>(p, batch, st);
    // the 128x128 tile: EIGHT waves (64x32 per wave, two per SIMD) for plain bf16 GEMMs since the second half of round 4 - a wave of the 4-wave form issues 8 DMA
    // pieces per 32 MFMAs with nothing on its SIMD to cover them; tools/gemm_iso.py 8192: 921 / 860 / 910 / 734 -> 1012 / 1118 / 1110 / 891 TFLOP/s (qkv / fc1 / fc2 /
    // proj), bit-identical.  fp32 and the implicit-GEMM convs keep four waves.
    auto launch_128 = [&](const GemmParams& q, int nb, hipStream_t s) -> int {
#if CVAR_TU_PLAIN && !CVAR_TU_CONV
        if constexpr (sizeof(T) == 2) {
            if (CVAR_GEMM_128_W8 && !q.conv && q.tile_cfg != 26) return launch_cfg<T, 128, 128, 2, 4>(q, nb, s);
        }
#endif
        return launch_cfg<T, 128, 128, 2, 2>(q, nb, s);
    };
    // channel counts of the VQVAE (160, 320) are multiples of 160 but not of 128: a 160-wide tile wastes no MFMA work
    // few output channels (the decoder's conv_out: 160 -> 3): a 256x32 tile wastes 10x instead of 42x of the MFMA work of a
    // 128-wide tile; the kernel is then bound by streaming the activations, as it should be
    if constexpr (sizeof(T) == 2) {
        if (p.conv && p.N <= 32 && p.M >= 4096 && p.stride == 1 && !p.up && p.Cin % 32 == 0 && p.split_tiles == 0 && gemm_cfg_override(p) != 0) {
            const long in_bytes = (long)(p.M / (p.Hout * p.Wout)) * p.Hin * p.Win * p.Cin * (long)sizeof(T);
            if (in_bytes < (1L << 31)) {
                GemmParams q = p;
                q.conv_bytes = (unsigned)in_bytes;
                return launch_cfg<T, 256, 32, 4, 1, 2, true>(q, batch, st);
            }
        }
    }
    if (p.N % 160 == 0 && (p.N % 128 != 0 || p.conv) && p.M >= 4096) {
        // stride-1 3x3 convs of the decoder / encoder trunks: 256x160 tile on the scalar-state conv addressing (conv FAST)
        const long in_bytes = p.conv ? (long)(p.M / (p.Hout * p.Wout)) * p.Hin * p.Win * p.Cin * (long)sizeof(T) : 0;
        if constexpr (sizeof(T) == 2) {
            if (p.conv && (p.stride == 1 || (p.stride == 2 && !p.up)) && p.Cin % 32 == 0 && p.split_tiles == 0 && in_bytes < (1L << 31) &&
                (long)159 * p.ldw * 2 + (long)p.K * 2 + 256 < (1L << 31) && gemm_cfg_override(p) != 0 &&
                (!p.up || (p.Hout == 2 * p.Hin && p.Wout == 2 * p.Win))) {
                GemmParams q = p;
                q.conv_bytes = (unsigned)in_bytes;
                // Short-K convs (Cin = 160: 22.5 K tiles): with one wave per SIMD the 13 DMA pieces a wave issues per K tile (~60-100 cycles
                // each) are as long as its 40 MFMAs and nothing hides them; eight waves (two per SIMD, 32x160 each, the 20 W pieces spread
                // with duplicates - see w_piece) let the partner's MFMAs cover the issue: 160->160 at 256^2 734 -> 783 TFLOP/s, with the
                // bf16 residual epilogue 628 -> 745; K >= 2880 is neutral and stays on 4 waves (profiles/r02_conv_8wave_ab.txt).
                // tile_cfg 3 forces the 4-wave tile, 4 the 8-wave tile.  Bit-identical results.
                const int ovc = gemm_cfg_override(p);
                if (!p.up && p.stride == 1 && ovc != 3 && (ovc == 4 || (long)p.K * (long)sizeof(T) <= 4096)) return launch_cfg<T, 256, 160, 8, 1, 2, true>(q, batch, st);
                return p.up ? launch_cfg<T, 256, 160, 4, 1, 2, true, true>(q, batch, st) : launch_cfg<T, 256, 160, 4, 1, 2, true>(q, batch, st);
            }
        }
        return launch_cfg<T, 128, 160, 4, 1>(p, batch, st);
    }
#if CVAR_TU_PLAIN && !CVAR_TU_CONV
    if constexpr (sizeof(T) == 2) {
        if (p.tile_cfg == 27 && !p.conv && p.split_tiles == 0) return launch_cfg<T, 256, 192, 4, 2>(p, batch, st);
    }
#endif
    // large streaming GEMMs: 256x256 tile - halves the operand bytes per flop and doubles the MFMA work per barrier; measured
    // +10..15 % over 128x128 on the d24 shapes.  tile_cfg 1 forces the 128x128 tile, 2 the 8-wave tile everywhere (A/B runs).
    const int ov = gemm_cfg_override(p);
    // N = 1920 / 5760 of d30 (C = 30 * 64) is a multiple of 128 only: the last 256-wide tile is half empty (2-6 % waste), still far
    // better than dropping the whole GEMM to the 128x128 tile
    const bool n_ok = p.N % 256 == 0 || (p.N % 128 == 0 && p.N >= 1536);
    // Two 256x256 variants share the loop and the epilogue code: 8 waves (2x4, two per SIMD, 128x64 per wave) and 4 waves (2x2, one
    // per SIMD, 128x128 per wave, accumulators in AGPRs).  The 4-wave one wins isolated GEMMs by 1-3 % (fewer LDS reads per MFMA),
    // the 8-wave one wins in the model by 1-2 % at every depth (d12 ... d30): twice the waves share the epilogue's loads / stores /
    // GELU and fill each other's DMA-issue bubbles.  Default: 8 waves; tile_cfg 3 selects the 4-wave variant (A/B runs), and the
    // split-K slices of long-K GEMMs (training weight gradients: plain fp32 partial stores) use it as well.
    // (round 2, measured and removed: a 192x128 tile - 4 waves, 80 KB of LDS, TWO workgroups per CU so that one's epilogue overlaps the
    //  other's K loop - ran 3-10 % below the 256x256 tile on every d24 shape, proj included: 749 vs 789 TFLOP/s.  gpurun_out/r2_tilecfg4.txt)
    if (sizeof(T) == 2 && (ov == 3 || (ov != 0 && ov != 1 && p.split_tiles > 0)) && (p.M >= 2048 || (p.split_tiles > 0 && p.M > 1024)) && n_ok)
        return launch_cfg<T, 256, 256, 2, 2>(p, batch, st);
    // Partial rounds: a launch of t tiles takes ceil(t / slots) rounds of the chip.  256x256: 256 slots (one workgroup per CU), cost 1
    // per round; 128x128: 512 slots (two per CU share its matrix cores), a quarter of the work at 0.82x the rate => 0.61 per round
    // (measured on the d24 shapes, tools/gemm_tilecfg.py).  The mid scales of the pyramid with N = C outputs (proj, fc2: 108..432
    // tiles of 256x256) fill the last round badly; the smaller tile wins there (M=4608: proj 464 -> 599, fc2 672 -> 873 TFLOP/s).
    // K order per output is the same for both tiles: bit-identical results.
    if (ov == -1 && sizeof(T) == 2 && p.split_tiles == 0 && batch == 1 && n_ok && p.M >= 2048) {
        const long t256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256), t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
        const double c256 = (double)((t256 + 255) / 256), c128 = CVAR_GEMM_C128 * (double)((t128 + 511) / 512);
#if CVAR_TU_PLAIN && !CVAR_TU_CONV
        // Round 5: a 256x192 tile (8 waves as 4 x 2, 64x96 per wave) for the launches whose 256x256 tiles fill their last round badly - the mid batches (B = 8 ... 32:
        // M = 2 048 ... 12 800 rows per scale).  Per flop it is 9-26 % slower than the 256x256 tile (M = 21 632: 1 172 / 1 148 / 1 022 / 1 038 against 1 285 / 1 313 /
        // 1 384 / 1 269 TFLOP/s for qkv / fc1 / fc2 / proj), i.e. a round of it costs ~0.88 of a 256x256 round instead of 0.75; it wins where it saves a round or fills
        // one that was half empty: M = 8 192 qkv 108 -> 101 us, M = 5 408 fc1 110 -> 98, fc2 113 -> 98, M = 12 800 fc2 247 -> 204 us (tools/tile_probe.py,
        // profiles/r05_gemm_tile_192.txt).  Same K order per output: bit-identical to the other tiles.  tile_cfg 27 forces it (A/B runs, tests).
        if constexpr (sizeof(T) == 2) {
            if (CVAR_GEMM_T192 && !p.conv && p.N % 192 == 0) {
                const long t192 = (long)((p.M + 255) / 256) * (p.N / 192);
                const double c192 = CVAR_GEMM_C192 * (double)((t192 + 255) / 256);
                // (5 % margin against the 256x256 plan only: the 0.61 per round of 128x128 tiles is optimistic where they need a third round - M = 3 200 fc1: 73 us against 61)
                if (c192 < 0.95 * c256 && c192 < c128) return launch_cfg<T, 256, 192, 4, 2>(p, batch, st);
            }
        }
#endif
        if (c128 < 0.97 * c256) {
#if CVAR_TU_PLAIN && !CVAR_TU_CONV
            // one round of at most 256 workgroups in a transformer pass: three LDS stages (see cvar_gemm: a workgroup of this regime is latency-bound)
            if constexpr (sizeof(T) == 2) {
                if (p.tile_cfg == 12 && !p.conv && t128 <= 256 && p.K >= 6 * 64) return launch_cfg<T, 128, 128, 2, 4, 3>(p, batch, st);
            }
#endif
            return launch_128(p, batch, st);
        }
    }
    if (ov != 0 && (p.M >= 2048 || (p.split_tiles > 0 && p.M > 1024)) && n_ok) return launch_cfg<T, 256, 256, 2, 4>(p, batch, st);
#if CVAR_TU_PLAIN && !CVAR_TU_CONV
    // tile_cfg 13 / 14: the 128x128 tile with three / four LDS stages - two / three K tiles in flight per workgroup (one workgroup per CU).  The three-stage
    // instance runs on EIGHT waves (64x32 per wave): with one wave per SIMD nothing hides the 8 DMA issues a wave has per 32 MFMAs - 4 waves / 8 waves, us per call:
    // M = 400 qkv 19.0 / 17.7, fc1 21.5 / 18.8; M = 1024 qkv 35.0 / 31.6 (same slices; bit-identical)
    if constexpr (sizeof(T) == 2) {
        if (p.tile_cfg == 13 && !p.conv) return launch_cfg<T, 128, 128, 2, 4, 3>(p, batch, st);
        if (p.tile_cfg == 14 && !p.conv) return launch_cfg<T, 128, 128, 2, 2, 4>(p, batch, st);
        // transformer passes (tile_cfg 12) that end up here unsliced - 1024 < M < 2048, or N not a 256-tile width - take the third stage as well when
        // all their workgroups fit the chip at once
        if (p.tile_cfg == 12 && !p.conv && batch == 1 && p.split_tiles == 0 && p.K >= 6 * 64 &&
            (long)((p.M + 127) / 128) * ((p.N + 127) / 128) <= 256) return launch_cfg<T, 128, 128, 2, 4, 3>(p, batch, st);
    }
#endif
    return launch_128(p, batch, st);
}

// gemm_skinny.hip: the weight-streaming small-M kernel and the row-finishing split-K reduction (+ adaLN of the next op)
int cvar_gemm_skinny_plan(int M, int N, int K, long lda, long ldw, int want_rowfin, int have_ws, int* mt, int* nt, int* slices);
int cvar_gemm_skinny_launch(const GemmParams& p, int mt, int nt, int slices, hipStream_t st);
int cvar_splitk_rowfin_ok(const GemmParams& p, const cvar_gemm_desc* d);
int cvar_splitk_rowfin_launch(const float* part, int nsplit, const GemmParams& p, const cvar_gemm_desc* d, hipStream_t st);
int cvar_gemm_launch_f32(const GemmParams& p, int batch, hipStream_t st);            // defined in the CVAR_GEMM_F32_TU compilation
int cvar_gemm_launch_conv_bf16(const GemmParams& p, int batch, hipStream_t st);      // defined in the CVAR_GEMM_CONV_TU compilation
#if defined(CVAR_GEMM_F32_TU)
int cvar_gemm_launch_f32(const GemmParams& p, int batch, hipStream_t st) { return launch_typed<float>(p, batch, st); }
#elif defined(CVAR_GEMM_CONV_TU)
int cvar_gemm_launch_conv_bf16(const GemmParams& p, int batch, hipStream_t st) { return launch_typed<bf16_t>(p, batch, st); }
#else

// ---- split-K: the small-M GEMMs of the early scales (and of small batches) have too few tiles to fill 256 CUs and are
// bound by the serial load->MFMA latency chain of one block's K loop.  They are split along K into slices that write fp32
// partial tiles to a caller workspace; this kernel sums the slices in a fixed order and applies the full epilogue.
__global__ __launch_bounds__(256) void cvar_splitk_epilogue_kernel(const float* __restrict__ part, int nsplit, const GemmParams p) {
    const long nvec = (long)p.M * (p.N / 4);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256)
        splitk_finish_quad(p, part, nsplit, (int)(i / (p.N / 4)), (int)(i % (p.N / 4)) * 4);
}


// Does a stride-1 3x3 bf16 conv of this shape emit GroupNorm partials of its output when cvar_gemm_desc.gn_part is set (the wide LDS-halo kernel, chosen
// by shape alone; operands must be 16-byte aligned and dense: ldc == N, ldw == 9 Cin, ldr == N)?  Returns 1 and the partial geometry, else 0.
extern "C" int cvar_conv3x3_gn_partials(int dtype, int stride, int Cin, int Cout, int Hin, int Win, int Hout, int Wout, int* tiles_per_image, int* pixels_per_tile) {
    // (Cin == 8: the image conv of conv_c8.hip - same tile geometry and partial format; it has no upsampled form)
#ifndef CVAR_NO_CONV_C8
    const bool c8 = Cin == 8 && Hin == Hout && Win == Wout;
#else
    const bool c8 = false;
#endif
    const bool ok = dtype == CVAR_BF16 && stride == 1 && Cin > 0 && (Cin % 32 == 0 || c8) && Cout > 0 && Cout % 160 == 0 && Hout > 0 && Wout > 0 && Hout % 16 == 0 && Wout % 16 == 0 &&
                    (long)Hin * Win * Cin * 2 < 0x7fffffffL && (long)Hout * Wout * Cout < 0x7fffffffL;
    if (!ok) return 0;
    if (tiles_per_image) *tiles_per_image = (Hout / 16) * (Wout / 16);
    if (pixels_per_tile) *pixels_per_tile = 256;
    return 1;
}

extern "C" int cvar_gemm(const cvar_gemm_desc* d, void* stream) {
    if (!d || !d->A || !d->W || !d->C) return CVAR_EINVAL;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch < 1) return CVAR_EINVAL;
    if (d->dtype != CVAR_F32 && d->dtype != CVAR_BF16) return CVAR_EUNSUPPORTED;
    const int es = d->dtype == CVAR_BF16 ? 2 : 4, kch = 16 / es;
    if (d->K % kch) return CVAR_EUNSUPPORTED;
    if (((uintptr_t)d->A & 15) || ((uintptr_t)d->W & 15)) return CVAR_EINVAL;
    if ((d->ldw % kch) || (d->strideW % kch) || (d->strideA % kch)) return CVAR_EUNSUPPORTED;
    if (d->conv) {
        if (d->Cin % kch || d->K != 9 * d->Cin) return CVAR_EUNSUPPORTED;
        if (d->stride != 1 && d->stride != 2) return CVAR_EUNSUPPORTED;
        if (d->stride == 2 && d->up) return CVAR_EUNSUPPORTED;
        if (d->stride == 1 && (d->Hout != (d->up ? 2 : 1) * d->Hin || d->Wout != (d->up ? 2 : 1) * d->Win)) return CVAR_EINVAL;
        if (d->stride == 2 && (d->Hout != d->Hin / 2 || d->Wout != d->Win / 2)) return CVAR_EINVAL;
        if (d->M % (d->Hout * d->Wout)) return CVAR_EINVAL;
    } else if (d->lda % kch) {
        return CVAR_EUNSUPPORTED;
    }
    if (d->gn_part && !d->conv) return CVAR_EINVAL;               // GroupNorm partials are a conv epilogue (ABI 18)
    if (d->gate && d->gate_rows <= 0) return CVAR_EINVAL;
    if (d->act == CVAR_ACT_GELU_GRAD && !d->aux) return CVAR_EINVAL;
    if (d->pre_act && d->act == CVAR_ACT_GELU_GRAD) return CVAR_EINVAL;
    if (d->gate_scale && !d->gate) return CVAR_EINVAL;
    if ((d->pre_act || d->aux) && d->remap_l > 0) return CVAR_EUNSUPPORTED;
    if (d->split_n < 0 || d->split_n >= d->N) return CVAR_EINVAL;
    if (d->split_n > 0) {
        if (!d->C_split || d->ld_split < d->split_n) return CVAR_EINVAL;
        // the split rides on the row-remap epilogue (the qkv GEMM is its one user) and moves whole 8-wide vectors
        if (d->remap_l <= 0 || d->conv || d->batch != 1 || d->residual || d->gate || d->act != CVAR_ACT_NONE) return CVAR_EUNSUPPORTED;
        if ((d->split_n & 7) || (d->N & 7) || (d->ld_split & 7) || (d->ldc & 7) || ((uintptr_t)d->C_split & 15) || ((uintptr_t)d->C & 15)) return CVAR_EUNSUPPORTED;
    }
    GemmParams p;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.A = (const char*)d->A; p.lda = d->lda; p.W = (const char*)d->W; p.ldw = d->ldw;
    p.strideA = d->strideA; p.strideW = d->strideW; p.strideC = d->strideC; p.strideR = d->strideR;
    p.conv = d->conv; p.Hin = d->Hin; p.Win = d->Win; p.Cin = d->Cin; p.Hout = d->Hout; p.Wout = d->Wout;
    p.stride = d->stride; p.up = d->up;
    p.alpha = d->alpha; p.bias = d->bias; p.act = d->act;
    p.gate = d->gate; p.ldg = d->ldg; p.gate_rows = d->gate_rows;
    p.residual = d->residual; p.res_dtype = d->res_dtype; p.ldr = d->ldr;
    p.C = d->C; p.out_dtype = d->out_dtype; p.ldc = d->ldc;
    p.C2 = d->pre_act; p.aux = d->aux; p.gate_scale = d->gate_scale; p.in_dtype = d->dtype;
    p.remap_l = d->remap_l; p.remap_L = d->remap_L; p.remap_off = d->remap_off;
    p.Cs = d->C_split; p.split_n = d->split_n; p.ld_split = d->ld_split; p.split_alpha = d->split_alpha == 0.0f ? 1.0f : d->split_alpha;
    p.tiles_m = p.tiles_n = 0;
    p.cv_adv = p.cv_rem = 0; p.conv_bytes = 0;
    make_fast_div(p.remap_l, &p.remap_magic, &p.remap_shift);
    make_fast_div(p.gate ? p.gate_rows : 1, &p.gate_magic, &p.gate_shift);
    p.split_tiles = 0; p.split_stride = 0;
    p.tile_cfg = d->tile_cfg; p.stagger = d->stagger > 0 ? d->stagger : 0; p.group_m = d->group_m > 0 ? d->group_m : 0;
    // round 6: outputs far beyond the L2s (32 MB) and the better part of the Infinity Cache stream - the specialised epilogues of the 256-row tiles store them non-temporally and
    // the fp32 read-modify-write of proj / fc2 takes its LDS-prefetched form; mid-size passes (B = 8: 25-100 MB per tensor) keep the default policy, which still finds them
    // in the caches when the next kernel reads them (B = 8: 133 -> 128.6 images/s with the policy on everything, profiles/r06_nt_policy_ab.txt)
    p.nt = ((long)d->M * d->N * (d->out_dtype == CVAR_F32 ? 4 : 2) >= (128L << 20)) ? 1 : 0;
    // split-K workspace: part of the call (caller-owned, any stream / device), nothing process-wide
    float* const g_splitk_ws = (d->ws && d->ws_bytes > 0 && (((uintptr_t)d->ws & 15) == 0)) ? (float*)d->ws : nullptr;
    const size_t g_splitk_ws_bytes = g_splitk_ws ? (size_t)d->ws_bytes : 0;
    hipStream_t st = as_stream(stream);
    // ABI 17: adaLN of the finished rows for the op that follows (cvar.h).  Row-wise fused with the split-K reduction where this call is sliced
    // (cvar_splitk_rowfin_kernel), a cvar_ln_modulate launch behind the GEMM otherwise - same bits either way.
    if (d->ln_out) {
        if (!d->ln_scale || !d->ln_shift || d->ln_rows <= 0 || (d->N % 4) || d->N > 2048 || (d->ld_ln % 4) || d->out_dtype != CVAR_F32 || d->ldc != d->N ||
            d->remap_l > 0 || d->split_n > 0 || d->conv || d->batch != 1 || (d->ln_out_dtype != CVAR_BF16 && d->ln_out_dtype != CVAR_F32)) return CVAR_EUNSUPPORTED;
    }
    auto ln_after = [&](int rc) -> int {
        if (rc != CVAR_OK || !d->ln_out) return rc;
        return cvar_ln_modulate((const float*)d->C, d->ln_scale, d->ln_shift, d->ld_ln, d->ln_rows, d->ln_out, d->ln_out_dtype, d->M, d->N, d->ln_eps, stream);
    };
    const bool rowfin = g_splitk_ws && cvar_splitk_rowfin_ok(p, d);
    // sums the slices a producer left in the workspace and finishes the call: row-wise with the adaLN, or element-wise (+ the adaLN launch)
    auto finish_slices = [&](int splits) -> int {
        if (rowfin) return cvar_splitk_rowfin_launch(g_splitk_ws, splits, p, d, st);
        const long nvec = (long)d->M * (d->N / 4);
        hipLaunchKernelGGL(cvar_splitk_epilogue_kernel, dim3((unsigned)min((long)2048, (nvec + 255) / 256)), dim3(256), 0, st, g_splitk_ws, splits, p);
        CVAR_CHECK_LAUNCH();
        return ln_after(CVAR_OK);
    };
    // small-M bf16 GEMMs whose caller opted in (tile_cfg 12): the weight-streaming kernel (gemm_skinny.hip) - whole K per workgroup, epilogue in the launch;
    // long K as a few slices + the row-finishing reduction.  Opt-in because the plan, hence the fp32 summation order of an output, depends on M: the
    // transformer's small passes accept that (their split-K already does), the VQVAE's per-image bit-reproducibility across batch sizes must not.
    if (!d->conv && d->batch == 1 && d->dtype == CVAR_BF16 && d->tile_cfg == 12 && !d->pre_act && !d->aux) {
        int mt = 0, nt = 0, sl = 0;
        if (cvar_gemm_skinny_plan(d->M, d->N, d->K, d->lda, d->ldw, rowfin ? 1 : 0, g_splitk_ws ? 1 : 0, &mt, &nt, &sl)) {
            if (sl > 1 && (size_t)sl * d->M * d->N * sizeof(float) <= g_splitk_ws_bytes) {
                GemmParams ps = p;
                ps.C = g_splitk_ws; ps.out_dtype = CVAR_F32; ps.ldc = d->N;
                ps.split_tiles = (d->K >> 5) / sl; ps.split_stride = (long)d->M * d->N;
                const int rc = cvar_gemm_skinny_launch(ps, mt, nt, sl, st);
                if (rc != CVAR_OK) return rc;
                return finish_slices(sl);
            }
            if (sl == 1) return ln_after(cvar_gemm_skinny_launch(p, mt, nt, 1, st));
        }
    }
    // split-K decision: plain (non-conv, unbatched) GEMMs whose tile count leaves most CUs idle
    const int kt_elems = 128 / es;
    const int nk_all = (d->K + kt_elems - 1) / kt_elems;
    const int tm = d->M <= 64 ? (d->M + 63) / 64 : (d->M + 127) / 128, tn = (d->N + 127) / 128;
    // (b) long-K GEMMs with a few hundred output tiles at most - the weight gradients of training (K = tokens of the batch):
    //     256x256 tiles, and the number of K slices s <= 8 that minimises ceil(tiles * s / 256) / s (rounds of the chip per slice)
    int long_k_splits = 0;
    const int t256 = ((d->M + 255) / 256) * ((d->N + 255) / 256);
    if (!d->conv && d->batch == 1 && d->dtype == CVAR_BF16 && d->M > 1024 && d->M % 128 == 0 && d->N % 128 == 0 && d->N >= 1536 && g_splitk_ws &&
        (nk_all >= 128 || (nk_all >= 24 && t256 <= 64 && !d->remap_l))) {
        const int tiles = t256;
        if (tiles < 256) {
            double best = 1e30;
            for (int sp = 1; sp <= 8; ++sp) {
                const double cost = (double)((tiles * sp + 255) / 256) / sp + 0.01 * sp;     // small bias against needless slices
                if (nk_all / sp < 6) break;                                                   // at least 6 K tiles per slice
                if (cost < best && (size_t)sp * d->M * d->N * sizeof(float) <= g_splitk_ws_bytes) { best = cost; long_k_splits = sp; }
            }
            if (long_k_splits < 2) long_k_splits = 0;
        }
    }
    // (Round 4, measured and removed: 64-row tiles UNSPLIT wherever they alone put >= 96 / 128 / 192 workgroups on the chip - e.g. M = 256 x N = 4608 as 144 tiles
    //  of 64x128 in one launch instead of 72 tiles of 128x128 split four ways + the reduction launch - is exactly neutral at B = 1 ... 16:
    //  profiles/r04_small_batch.txt.  The launches of this regime sit on their latency floor either way.)
    // Mid-M passes of the transformer (64 < M <= 1024, tile_cfg 12): the 128x128 tile with THREE LDS stages wherever all its workgroups fit the chip
    // at once (tiles x slices <= 256, one workgroup of 96 KB per CU).  A workgroup of this regime is bound by the latency of its own K loop - one K tile
    // of DMA in flight behind the one being multiplied - not by the matrix pipe: two tiles in flight run M = 400 qkv / fc1 27.7 / 28.6 -> 18.5 / 19.6 us and
    // M = 676 qkv 28.2 -> 19.4 us, while launches of more than 256 workgroups (a second round at one workgroup per CU) lose 15-20 % and keep two stages
    // (tools/skinny_bench.py with ISO_CFG = 13, profiles/r04_small_batch.txt).  Slices are then chosen to fill 256 workgroups instead of 320.
    const bool three = d->tile_cfg == 12 && !d->conv && d->batch == 1 && d->dtype == CVAR_BF16 && d->M > 64 && d->M <= 1024 && (d->N % 4) == 0 && tm * tn <= 256 &&
                       nk_all >= 6 && !d->pre_act && !d->aux;
    int three_splits = 1;
    if (three && g_splitk_ws) {
        three_splits = min(16, max(1, 256 / (tm * tn)));
        while (three_splits > 1 && nk_all / three_splits < 4) --three_splits;           // at least 4 K tiles per slice: the pipeline needs a loop to fill
    }
    if (three) p.tile_cfg = 13;                                                         // launch_typed: three-stage instance of the 128x128 tile
    if (long_k_splits || (three && three_splits > 1) ||
        (!three && !d->conv && d->batch == 1 && d->M <= 1024 && (d->N % 4) == 0 && tm * tn < 128 && nk_all >= 8 && g_splitk_ws)) {
        int splits = long_k_splits ? long_k_splits : three ? three_splits : min(16, max(2, 320 / (tm * tn)));
        int per = (nk_all + splits - 1) / splits;
        if (per < 2) per = 2;
        splits = (nk_all + per - 1) / per;
        const size_t need = (size_t)splits * d->M * d->N * sizeof(float);
        if (splits > 1 && need <= g_splitk_ws_bytes) {
            GemmParams ps = p;
            ps.alpha = 1.0f; ps.bias = nullptr; ps.act = CVAR_ACT_NONE; ps.gate = nullptr; ps.residual = nullptr; ps.C2 = nullptr; ps.aux = nullptr; ps.gate_scale = nullptr;
            ps.C = g_splitk_ws; ps.out_dtype = CVAR_F32; ps.ldc = d->N; ps.remap_l = 0; ps.split_n = 0; ps.Cs = nullptr; ps.strideC = 0;
            ps.split_tiles = per; ps.split_stride = (long)d->M * d->N;
            const int rc = d->dtype == CVAR_BF16 ? launch_typed<bf16_t>(ps, 1, st) : cvar_gemm_launch_f32(ps, 1, st);
            if (rc != CVAR_OK) return rc;
            return finish_slices(splits);
        }
    }
    // the image conv (Cin = 3 padded to 8, 160-multiples of couts: the encoder's conv_in): conv_c8.hip - K = 72 is two half-empty K tiles of gathered 16-byte pieces on the
    // implicit-GEMM tiles (1.9 TB/s of output), and only the LDS-tile kernels emit GroupNorm partials.  Chosen by shape alone (an image's bits do not depend on its batch).
#ifndef CVAR_NO_CONV_C8          // (A/B builds: tools/build_variant.py gemm.hip noc8 -DCVAR_NO_CONV_C8 keeps conv_in on the implicit-GEMM tiles)
    if (d->conv && d->dtype == CVAR_BF16 && d->stride == 1 && !d->up && d->batch == 1 && (d->tile_cfg == 0 || d->tile_cfg == 6) && d->Cin == 8 && d->N % 160 == 0 &&
        d->Hout % 16 == 0 && d->Wout % 16 == 0 && d->act == CVAR_ACT_NONE && !d->gate && d->alpha == 1.0f && !d->residual && !d->pre_act && !d->aux && !d->gate_scale &&
        d->remap_l == 0 && d->split_n == 0 && d->strideC == 0 && d->ldc == d->N && d->ldw == d->K && d->out_dtype == CVAR_BF16 &&
        ((((uintptr_t)d->C | (uintptr_t)d->bias) & 15) == 0) && (long)d->Hout * d->Wout * d->N < 0x7fffffffL)
        return cvar_conv3x3_c8_bf16(d->A, d->W, d->bias, d->C, d->M / (d->Hout * d->Wout), d->Hout, d->Wout, d->N, d->gn_part, st);
#endif
    // stride-1 3x3 convs (plain or behind the nearest x2 upsample) over 32-channel multiples with 160-multiple outputs on 16-multiple images (every ResnetBlock conv of the VQVAE
    // decoder from 16x16 up): the LDS-halo kernel (conv_halo.hip).  tile_cfg 5 keeps them on the implicit-GEMM tiles, 6 forces the halo kernel at any grid size (A/B runs, tests).
    if (d->conv && d->dtype == CVAR_BF16 && d->stride == 1 && d->batch == 1 && (d->tile_cfg == 0 || d->tile_cfg == 6 || d->tile_cfg == 9 || d->tile_cfg == 10) &&
        d->Cin % 32 == 0 && d->Hout % 16 == 0 && d->Wout % 16 == 0 && d->act == CVAR_ACT_NONE && !d->gate && d->alpha == 1.0f &&
        !d->pre_act && !d->aux && !d->gate_scale && d->remap_l == 0 && d->split_n == 0 && d->strideC == 0 && d->ldc == d->N &&
        d->ldw == d->K &&      // the halo kernel addresses packed [Cout][9 Cin] weights: padded weight rows stay on the implicit-GEMM path
        (!d->bias || (((uintptr_t)d->bias & 15) == 0)) && (long)d->Hin * d->Win * d->Cin * 2 < 0x7fffffffL) {
        // wide form: Cout a multiple of 160, bf16 output, optional bf16 residual; two workgroups per CU or the implicit-GEMM tiles win (640->640 at 16x16)
        // (16-byte alignment and < 2^31 elements per output image: the row-major epilogue moves bf16x8 vectors at 32-bit offsets inside an image)
        const bool wide = d->N % 160 == 0 && d->out_dtype == CVAR_BF16 && (((uintptr_t)d->C & 15) == 0) &&
                          (!d->residual || (d->res_dtype == CVAR_BF16 && d->ldr == d->N && (((uintptr_t)d->residual & 15) == 0))) &&
                          (long)d->Hout * d->Wout * d->N < 0x7fffffffL;       // round 4: at any grid size - small batches (B = 1: 36.5 -> 34.9 ms, B = 8: 71.0 -> 69.8 ms per generation, profiles/r04_small_batch.txt) gain too
        // narrow form: Cout <= 32 (conv_out, 160 -> 3), bf16 or fp32 output, no residual - the implicit-GEMM tile spends its time re-fetching the input
        // (decided by the IMAGE size, not by the batch: the two kernels sum K in different orders, and an image's pixels must not depend on the batch it rides in)
        const bool narrow = d->N <= 32 && !d->residual && (d->out_dtype == CVAR_BF16 || d->out_dtype == CVAR_F32) && ((long)d->Hout * d->Wout >= 65536 || d->tile_cfg == 6);
        if (d->gn_part && !wide) return CVAR_EUNSUPPORTED;       // GroupNorm partials come out of the wide halo kernel only (cvar_conv3x3_gn_partials says which calls)
        // round 6: the same kernel with an fp32 output and an fp32 residual - the 3x3 convs of the split-bf16 encoder (fp32 activation stream, 3 x Cin split channels)
        const bool wide32 = d->N % 160 == 0 && d->out_dtype == CVAR_F32 && (((uintptr_t)d->C & 15) == 0) && !d->up &&
                            (!d->residual || (d->res_dtype == CVAR_F32 && d->ldr == d->N && (((uintptr_t)d->residual & 15) == 0))) &&
                            (long)d->Hout * d->Wout * d->N < 0x7fffffffL;
        if (wide || narrow || wide32)
            return cvar_conv3x3_halo_bf16(d->A, d->W, d->bias, d->residual, d->C, d->out_dtype == CVAR_F32, d->M / (d->Hout * d->Wout), d->Hout, d->Wout, d->Cin,
                                          d->N, d->up, d->gn_part, st);
    }
    if (d->gn_part) return CVAR_EUNSUPPORTED;
    if (d->dtype == CVAR_BF16) return d->conv ? cvar_gemm_launch_conv_bf16(p, d->batch, st) : ln_after(launch_typed<bf16_t>(p, d->batch, st));
    return ln_after(cvar_gemm_launch_f32(p, d->batch, st));
}
#endif   // main translation unit
