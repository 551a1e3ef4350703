// 3x3 stride-1 convolution over NHWC bf16 with the input tile (plus halo) resident in LDS.
//
// The implicit-GEMM conv of gemm.hip fetches every input pixel nine times (once per tap) through `buffer_load ... lds`, and at the
// VQVAE's N = 160 tile that DMA issue rate (52 per 1 288 MFMA cycles) and the LDS re-reads of a 32x160 wave tile bound the loop
// (profiles/r02_conv_timing.txt).  Here a workgroup owns a 16x16 pixel tile x 160 output channels:
//   * per 32-channel chunk the 18x18 halo tile is DMA'd ONCE (23 pieces of 1 KiB, rows padded to 20 pixels) and the nine taps read it at shifted addresses;
//   * weights stream as [160 couts][32 channels] tiles per (chunk, tap) (10 pieces) through a ring of three;
//   * 4 waves x (64 pixels x 160 couts) = 20 MFMA 32x32x16 per wave and k-step pair, 7 fragment reads per 10 MFMAs;
//   * 76 KB of LDS and <= 256 registers: TWO workgroups per CU, so one's barriers / epilogue hide behind the other's MFMAs.
// K order: chunk-major, then tap (the implicit-GEMM kernel runs tap-major) - fp32 accumulation order differs, same math.
// vae_modules.py:40-60 (ResnetBlock convs), :28 (Upsample conv after the nearest x2, handled by the caller), NHWC / [Cout][ky][kx][Cin].
#include "cvar_common.h"

typedef __attribute__((address_space(3))) void* lptr_t;

#ifndef CH_EPI_LDS
#define CH_EPI_LDS 1
#endif
// (round-4 experiments - timing ablations, 32-pixel-wide tiles on 8 waves: 1-6 % slower, profiles/r04_conv_ablation.txt - live in the commits listed in
// experiments/README.md, not here)

struct ConvHaloParams {
    const bf16_t* X; const bf16_t* Wt; const float* bias; const bf16_t* res; void* out;
    int B, H, W, Cin, Cout, tiles_x, tiles_y, nchunk;
    int up, Hin, Win;          // up = 1: the conv reads its input through a nearest x2 upsample (Hin = H / 2), vae_modules.py:28
    float* gn_part;            // optional (wide form): GroupNorm partial sums of the OUTPUT, [B][tiles][Cout][3] = (sum (y - piv), sum (y - piv)^2, piv) per tile and channel
};

// Tile width TW = 16 (4 waves, two workgroups per CU).  Halo row stride TW + 4 pixels (TW + 2 used): a multiple of 4, so that the bank slot depends on the column only.

// LDS images: 64-byte rows (one pixel / one cout x 32 channels) whose four 16-B chunks are XOR-swizzled so that every lane group of a
// ds_read_b128 covers all 64 banks once.  The hardware's groups are NOT 16 consecutive lanes: group 0 = lanes {0-3, 12-15, 20-27} etc.
// (MI355X_MICROARCH.md).  Weights: lane = row, swizzle (row >> 2) & 3 - any 16 rows that are distinct mod 16 are conflict-free.  Halo: a
// group holds columns {0-3, 12-15} of one image row and {4-11} of the next, so the slot must depend on the COLUMN only: rows are padded to 20
// pixels (20 mod 4 = 0) and the swizzle is ((column) >> 2) & 3 - then the group's 16 distinct columns (mod 16, for every tap shift) hit 16
// distinct slots.  (With 18-pixel rows and a swizzle of the linear index two lanes of every group collided; fixing it measured +0.3 % -
// the halo reads are 4 of the 14 fragment reads of a step - but it also freed 33 VGPRs of address state.)
// CH_NB = 32-wide cout blocks per tile: 5 (the 160-multiples of the VQVAE) or 1 (conv_out's 3 channels: couts beyond Cout are zero rows of the
// weight tile and are not stored).  TO = output element type (bf16, or fp32 for conv_out).
template <int CH_NB, typename TO, int TW = 16>
__global__ __launch_bounds__(TW * 16, 2) void conv3x3_halo_bf16_kernel(const ConvHaloParams p) {
    constexpr int NW = TW / 4;                        // waves: 64 pixels each
    constexpr int CH_HROW = TW + 4;
    constexpr int CH_HALO_PIECES = (18 * CH_HROW * 64 + 1023) / 1024;        // 23 (TW = 16) / 41 (TW = 32) DMA pieces of 1 KiB
    constexpr int CH_HALO_BYTES = CH_HALO_PIECES * 1024;
    constexpr int H_PER_WAVE = (CH_HALO_PIECES + NW - 1) / NW;               // 6 / 6
    constexpr int CH_W_BYTES = CH_NB * 2048;          // 32 CH_NB rows x 64 B
    constexpr int W_PIECES = CH_NB * 2;               // 1-KiB DMA pieces per weight tile
    constexpr int W_PER_WAVE = (W_PIECES + NW - 1) / NW;
    __shared__ __attribute__((aligned(1024))) char smem[2 * CH_HALO_BYTES + 3 * CH_W_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, hi = lane >> 5;
    int t_ = blockIdx.x;            // (an XCD-contiguous tile order, so that neighbouring tiles share one L2, measured neutral)
    const int tx = t_ % p.tiles_x; t_ /= p.tiles_x;
    const int ty = t_ % p.tiles_y;
    const int b = t_ / p.tiles_y;
    const int ty0 = ty * 16, tx0 = tx * TW;
    const int cout0 = blockIdx.y * (32 * CH_NB);
    const bf16_t* ximg = p.X + (long)b * p.Hin * p.Win * p.Cin;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ximg, 0, p.Hin * p.Win * p.Cin * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Wt + (long)cout0 * 9 * p.Cin), 0, min(32 * CH_NB, p.Cout - cout0) * 9 * p.Cin * 2, 0x00020000);

    // DMA piece q of an image = 16-B slot q of the LDS image: row q >> 2, physical chunk q & 3 <- logical chunk (q & 3) ^ ((row >> 2) & 3)
    unsigned h_off[H_PER_WAVE], w_off[W_PER_WAVE];
#pragma unroll
    for (int jj = 0; jj < H_PER_WAVE; ++jj) {
        const int q = (wave + NW * jj) * 64 + lane;
        const int hp = q >> 2;
        const int hy = hp / CH_HROW, hx = hp - hy * CH_HROW;
        const int lc = (q & 3) ^ ((hx >> 2) & 3);
        const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
        const bool ok = hy < 18 && hx < TW + 2 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        // out of range -> the DMA writes zeros (the conv's zero padding); upsample: output-grid pixel (gy, gx) reads input (gy >> 1, gx >> 1)
        const int sy = p.up ? gy >> 1 : gy, sx = p.up ? gx >> 1 : gx;
        h_off[jj] = ok ? (unsigned)(((sy * p.Win + sx) * p.Cin + lc * 8) * 2) : 0x80000000u;
    }
#pragma unroll
    for (int jj = 0; jj < W_PER_WAVE; ++jj) {
        const int q = (wave + NW * jj) * 64 + lane;
        const int n = q >> 2, lc = (q & 3) ^ ((n >> 2) & 3);
        w_off[jj] = (n < 32 * CH_NB && cout0 + n < p.Cout) ? (unsigned)((n * 9 * p.Cin + lc * 8) * 2) : 0x80000000u;   // rows past Cout: zeros
    }
    auto issue_halo = [&](int c, int jj) -> int {
        const int j = wave + NW * jj;
        if (j >= CH_HALO_PIECES) return 0;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lptr_t)(smem + (c & 1) * CH_HALO_BYTES + j * 1024), 16, (int)h_off[jj], c * 64, 0, 0);
        return 1;
    };
    // weight tiles live in a ring of three: tile (c, t) in slot t % 3 (9 taps per chunk, so the slot does not depend on c)
    auto issue_w = [&](int c, int t) -> int {
        int n = 0;
#pragma unroll
        for (int jj = 0; jj < W_PER_WAVE; ++jj) {
            const int j = wave + NW * jj;
            if (j < W_PIECES) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lptr_t)(smem + 2 * CH_HALO_BYTES + (t % 3) * CH_W_BYTES + j * 1024), 16,
                                                         (int)w_off[jj], (t * p.Cin + c * 32) * 2, 0, 0);
                ++n;
            }
        }
        return n;
    };
    // wait until at most n (wave-uniform, 0..4) of this wave's DMA pieces are outstanding
    auto wait_vm = [&](int n) {
        if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // fragment addressing: pixel of lane = ((64 / TW) wave + (32 / TW) i + lrow / TW, lrow % TW) of the tile -> halo pixel (y + dy) * CH_HROW + (x + dx)
    constexpr int RPI = 32 / TW;                      // image rows of a 32-pixel MFMA row block
    const int px = lrow & (TW - 1), py0 = 2 * RPI * wave + lrow / TW;
    const int hpb = py0 * CH_HROW + px;
    const int swb = (lrow >> 2) & 3;
    const int b_lane = lrow * 64;

    f32x16_t acc[2][CH_NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < CH_NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int jj = 0; jj < H_PER_WAVE; ++jj) issue_halo(0, jj);
    issue_w(0, 0);
    issue_w(0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // Step (c, t): issue one halo piece of chunk c + 1 and the weight tile two steps ahead, compute, then wait only for what was issued in
    // EARLIER steps (vmcnt retires in order): every DMA gets a full step or more to land.  (Measured and rejected: a dedicated halo-DMA wave
    // that waits once per chunk while three waves stream the weights: -3 %.)
    // A fragments (halo reads) of a tap are fetched one step ahead - the halo buffer does not change inside a chunk, and the next chunk's has
    // landed by tap 6 - so a step starts its MFMAs as soon as the first weight fragment is back from LDS (+1-3 %).  af[set][i][ks]
    bf16x8_t af[2][2][2];
    auto read_a = [&](int set, const char* hbuf, int d) {
        const int sw = ((px + d % CH_HROW) >> 2) & 3;          // column of the lane's pixel + the tap's column shift
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hp = hpb + RPI * CH_HROW * i + d;
            const char* ap = hbuf + hp * 64;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[set][i][ks] = *(const bf16x8_t*)(ap + (((2 * ks + hi) ^ sw) << 4));
        }
    };
    read_a(0, smem, 0);
    for (int c = 0; c < p.nchunk; ++c) {
        const char* hb = smem + (c & 1) * CH_HALO_BYTES;
        const char* hb_next = smem + ((c + 1) & 1) * CH_HALO_BYTES;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            int issued = 0;
            if (t < H_PER_WAVE && c + 1 < p.nchunk) issued += issue_halo(c + 1, t);
            {
                const int c2 = t + 2 >= 9 ? c + 1 : c, t2 = t + 2 >= 9 ? t + 2 - 9 : t + 2;
                if (c2 < p.nchunk) issued += issue_w(c2, t2);
            }
            const char* wb = smem + 2 * CH_HALO_BYTES + (t % 3) * CH_W_BYTES + b_lane;
            bf16x8_t w[2][CH_NB];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < CH_NB; ++j) w[ks][j] = *(const bf16x8_t*)(wb + j * 2048 + (((2 * ks + hi) ^ swb) << 4));
            // next tap's A fragments (tap 0 of the next chunk after tap 8; past the last chunk the read is harmless and unused)
            if (t < 8) read_a((t + 1) & 1, hb, ((t + 1) / 3) * CH_HROW + ((t + 1) % 3));
            else read_a(1, hb_next, 0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < CH_NB; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ks][j], af[t & 1][i][ks], acc[i][j], 0, 0, 0);
            wait_vm(issued);
            __builtin_amdgcn_s_barrier();
        }
        // nine taps per chunk: the prefetched set is 1 after tap 8 - move it to set 0 so that every chunk starts alike
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[0][i][ks] = af[1][i][ks];
    }

    // epilogue: lane = pixel, register quad g of block j = couts 32 j + 8 g + 4 hi .. +3 (swapped MFMA operands) -> 8-byte accesses.
    // All residual loads of a row block are issued before its first store: vmcnt retires in order and counts stores, so a load queued
    // behind stores would wait for their latency as well (+13 % on the residual convs).  The bias comes straight from global memory (L2
    // hits): a second static LDS array for it cost 6 % on every shape.
    float gs[8], gq[8], gpiv[8];                 // GroupNorm partials of this lane's (column chunk, pixel residue): wide form with p.gn_part only
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int yy = py0 + RPI * i, xx = px;
        const long gpix = ((long)b * p.H + ty0 + yy) * p.W + tx0 + xx;
        if constexpr (CH_NB == 1) {
            // narrow output (conv_out: 3 channels): element-wise predicated stores of the couts that exist, no residual
            TO* op = (TO*)p.out + gpix * p.Cout + cout0;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = 8 * g + 4 * hi + e;
                    if (cout0 + co < p.Cout) {
                        const float v = acc[i][0][4 * g + e] + (p.bias ? p.bias[cout0 + co] : 0.f);
                        if constexpr (sizeof(TO) == 4) op[co] = v;
                        else op[co] = f32_to_bf16(v);
                    }
                }
        } else if constexpr (CH_NB == 5 && sizeof(TO) == 4) {
            // wide form with an fp32 output and an optional fp32 residual (round 6: the convs of the split-bf16 / fp32-stream encoder, models.VQVAE._hp_conv): straight from the
            // accumulator layout - a lane's four consecutive couts are one 16-byte store; the residual quads of one 32-cout block are loaded just before it is finished
            float* op = (float*)p.out + gpix * p.Cout + cout0 + 4 * hi;
            const float* rp = p.res ? (const float*)p.res + gpix * p.Cout + cout0 + 4 * hi : nullptr;
            const float* bp = p.bias ? p.bias + cout0 + 4 * hi : nullptr;
#pragma unroll
            for (int j = 0; j < CH_NB; ++j) {
                f32x4_t rq[4];
                if (rp) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) rq[g] = *(const f32x4_t*)(rp + 32 * j + 8 * g);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = 32 * j + 8 * g;
                    f32x4_t bq = {0.f, 0.f, 0.f, 0.f};
                    if (bp) bq = *(const f32x4_t*)(bp + co);
                    f32x4_t v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + bq[e];
                    if (rp) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rq[g][e];
                    }
                    *(f32x4_t*)(op + co) = v;
                }
            }
        } else if constexpr (CH_EPI_LDS != 0 && CH_NB == 5) {
            // Row-major epilogue through LDS (round 4).  In the accumulator layout a store instruction writes 16 B to each of 32 pixels (32 cache lines, 2 560
            // partial-line requests per wave and tile) and a residual load reads the same way; that burst also sits in front of the co-resident workgroup's
            // DMA stream, which is why the epilogue (19 % of the 160->160 conv at 256^2, 28 % with the residual: tools ablation, profiles/r04_conv_ablation.txt)
            // did not hide behind the partner's MFMAs.  Here the wave's 32 pixels x 160 couts of this row block pass through its own slice of the (now free)
            // pipeline LDS as [pixel][cout] rows of 336 B: residual in by 16-byte row-contiguous loads, summed in the accumulator layout in fp32 exactly as
            // before (acc + bias + residual, one rounding), out by 16-byte row-contiguous stores - 160 full-line requests per wave and tile.
            constexpr int PS = 336;
            constexpr int SLICE = (2 * CH_HALO_BYTES + 3 * CH_W_BYTES) / NW / 64 * 64;      // LDS of one wave; rows take 32 PS = 10 752 B of it
            static_assert(32 * PS + 512 + 3 * 20 * 64 <= SLICE, "GroupNorm partials need the slack behind the staged rows");
            char* stg = smem + wave * SLICE;
            const float* bp = p.bias ? p.bias + cout0 + 4 * hi : nullptr;
            // chunk q = 64 k + lane of the row block: pixel q / 20, 16-byte part q % 20; offsets relative to the image (32-bit)
            const long img = (long)b * p.H * p.W * p.Cout;
            auto chunk = [&](int k, int& go, int& lo) {
                const int q = k * 64 + lane, pq = q / 20, part = q - pq * 20;
                go = ((ty0 + 2 * RPI * wave + RPI * i + pq / TW) * p.W + tx0 + (pq & (TW - 1))) * p.Cout + cout0 + part * 8;
                lo = pq * PS + part * 16;
            };
            if (p.res) {
                const bf16_t* rimg = p.res + img;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    bf16x8_t rr[5];
                    int lo[5];
#pragma unroll
                    for (int k = 0; k < 5; ++k) { int go; chunk(5 * h + k, go, lo[k]); rr[k] = *(const bf16x8_t*)(rimg + go); }
#pragma unroll
                    for (int k = 0; k < 5; ++k) *(bf16x8_t*)(stg + lo[k]) = rr[k];
                }
            }
            char* mine = stg + lrow * PS + 8 * hi;
#pragma unroll
            for (int j = 0; j < CH_NB; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = 32 * j + 8 * g;
                    f32x4_t bq = {0.f, 0.f, 0.f, 0.f};
                    if (bp) bq = *(const f32x4_t*)(bp + co);
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + bq[e];
                    if (p.res) {
                        const bf16x4_t rq = *(const bf16x4_t*)(mine + co * 2);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += bf16_to_f32((bf16_t)rq[e]);
                    }
                    *(bf16x4_t*)(mine + co * 2) = pack_bf16x4(v);
                }
#pragma unroll
            for (int k = 0; k < 10; ++k) { int go, lo; chunk(k, go, lo); *(bf16x8_t*)((bf16_t*)p.out + img + go) = *(const bf16x8_t*)(stg + lo); }
            // ---- GroupNorm statistics of the tile (round 5; replaces the gn_stats pass of the GroupNorm that reads this conv's output, vae_modules.py:12-13).
            // The staged rows hold the values as STORED (bf16): lane (cc, pg) = (16-byte column chunk, pixel residue mod 3) walks its pixels and accumulates
            // sum (y - piv), sum (y - piv)^2 per channel in fp32 around a pivot common to the whole tile (the channel's value at the tile's first staged pixel:
            // no cancellation when |mean| >> std, as in gn_stats_kernel); the tiles are combined in double by gn_finalize_tiles_kernel.  Fixed order, no atomics.
            if (p.gn_part) {
                char* pv = smem + 32 * PS;                                  // 160 bf16 pivots, behind wave 0's rows
                if (i == 0) {
                    __builtin_amdgcn_wave_barrier();
                    if (wave == 0 && lane < 20) *(bf16x8_t*)(pv + lane * 16) = *(const bf16x8_t*)(stg + lane * 16);       // row 0 of wave 0 (in-order LDS: written above)
                    __syncthreads();
                    if (lane < 60) {
                        const bf16x8_t pq8 = *(const bf16x8_t*)(pv + (lane % 20) * 16);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { gpiv[e] = bf16_to_f32((bf16_t)pq8[e]); gs[e] = 0.f; gq[e] = 0.f; }
                    }
                }
                if (lane < 60) {
                    const int cc = lane % 20, pg = lane / 20;
#pragma unroll
                    for (int it = 0; it < 11; ++it) {
                        const int pp = pg + 3 * it;
                        if (pp < 32) {
                            const bf16x8_t y8 = *(const bf16x8_t*)(stg + pp * PS + cc * 16);
#pragma unroll
                            for (int e = 0; e < 8; ++e) { const float f = bf16_to_f32((bf16_t)y8[e]) - gpiv[e]; gs[e] += f; gq[e] = __builtin_fmaf(f, f, gq[e]); }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();      // the next row block's residual / output rows overwrite the staged rows: reads first
            }
        } else {
            bf16_t* op = (bf16_t*)p.out + gpix * p.Cout + cout0 + 4 * hi;
            const bf16_t* rp = p.res ? p.res + gpix * p.Cout + cout0 + 4 * hi : nullptr;
            const float* bp = p.bias ? p.bias + cout0 + 4 * hi : nullptr;
            bf16x4_t rq[CH_NB][4];
            if (rp) {
#pragma unroll
                for (int j = 0; j < CH_NB; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) rq[j][g] = *(const bf16x4_t*)(rp + 32 * j + 8 * g);
            }
#pragma unroll
            for (int j = 0; j < CH_NB; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = 32 * j + 8 * g;
                    f32x4_t bq = {0.f, 0.f, 0.f, 0.f};
                    if (bp) bq = *(const f32x4_t*)(bp + co);
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + bq[e];
                    if (rp) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += bf16_to_f32((bf16_t)rq[j][g][e]);
                    }
                    *(bf16x4_t*)(op + co) = pack_bf16x4(v);
                }
        }
    }
    if constexpr (CH_NB == 5 && CH_EPI_LDS != 0 && sizeof(TO) == 2) {
        if (p.gn_part) {
            // per-lane partials -> the wave's slack -> 160 threads sum (wave, pixel residue) in a fixed order and write (S, Q, piv) of their channel
            constexpr int PS = 336, SLICE = (2 * CH_HALO_BYTES + 3 * CH_W_BYTES) / NW / 64 * 64, RED = 32 * PS + 512;
            if (lane < 60) {
                float* rd = (float*)(smem + wave * SLICE + RED) + lane * 16;
#pragma unroll
                for (int e = 0; e < 8; e += 4) {
                    *(f32x4_t*)(rd + e) = f32x4_t{gs[e], gs[e + 1], gs[e + 2], gs[e + 3]};
                    *(f32x4_t*)(rd + 8 + e) = f32x4_t{gq[e], gq[e + 1], gq[e + 2], gq[e + 3]};
                }
            }
            __syncthreads();
            if (tid < 160) {
                const int cc = tid >> 3, e = tid & 7;
                float S = 0.f, Q = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w)
#pragma unroll
                    for (int pg = 0; pg < 3; ++pg) {
                        const float* rd = (const float*)(smem + w * SLICE + RED) + (pg * 20 + cc) * 16;
                        S += rd[e]; Q += rd[8 + e];
                    }
                const float piv = bf16_to_f32(*(const bf16_t*)(smem + 32 * PS + tid * 2));
                float* gp = p.gn_part + ((((long)b * p.tiles_y + ty) * p.tiles_x + tx) * p.Cout + cout0 + tid) * 3;
                gp[0] = S; gp[1] = Q; gp[2] = piv;
            }
        }
    }
}

// (H, W: the OUTPUT grid) eligibility is checked by the caller (cvar_gemm): bf16 operands, stride 1, Cin % 32 == 0, H % 16 == 0, W % 16 == 0 and
// either Cout % 160 == 0 with a bf16 output (optional bf16 residual) or an fp32 output (optional FP32 residual; round 6), or Cout <= 32 without residual (bf16 or fp32 output: conv_out)
int cvar_conv3x3_halo_bf16(const void* X, const void* Wt, const float* bias, const void* residual, void* out, int out_f32, int B, int H, int W, int Cin,
                           int Cout, int up, float* gn_part, hipStream_t st) {
    ConvHaloParams p;
    p.X = (const bf16_t*)X; p.Wt = (const bf16_t*)Wt; p.bias = bias; p.res = (const bf16_t*)residual; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.tiles_x = W / 16; p.tiles_y = H / 16; p.nchunk = Cin / 32;
    p.up = up ? 1 : 0; p.Hin = up ? H / 2 : H; p.Win = up ? W / 2 : W;
    p.gn_part = gn_part;
    if (gn_part && !(Cout % 160 == 0 && !out_f32)) return CVAR_EUNSUPPORTED;      // only the wide form emits GroupNorm partials
    const long tiles = (long)B * p.tiles_x * p.tiles_y;
    if (tiles <= 0 || tiles > 0x7fffffffL) return CVAR_EINVAL;
    if (Cout % 160 == 0 && out_f32) {
        // round 6: fp32 output + optional fp32 residual (16-byte aligned, < 2^31 elements per image as below); no GroupNorm partials
        if ((((uintptr_t)out | (uintptr_t)residual) & 15) != 0 || (long)H * W * Cout >= 0x7fffffffL) return CVAR_EUNSUPPORTED;
        hipLaunchKernelGGL((conv3x3_halo_bf16_kernel<5, float>), dim3((unsigned)tiles, Cout / 160), dim3(256), 0, st, p);
    } else if (Cout % 160 == 0 && !out_f32) {
        // the row-major epilogue moves 16-byte vectors at 32-bit element offsets inside an image: callers whose output / residual are only
        // 8-byte aligned or whose images exceed 2^31 elements stay on the implicit-GEMM tiles
        if ((((uintptr_t)out | (uintptr_t)residual) & 15) != 0 || (long)H * W * Cout >= 0x7fffffffL) return CVAR_EUNSUPPORTED;
        hipLaunchKernelGGL((conv3x3_halo_bf16_kernel<5, bf16_t>), dim3((unsigned)tiles, Cout / 160), dim3(256), 0, st, p);
    } else if (Cout <= 32 && !residual) {
        if (out_f32) hipLaunchKernelGGL((conv3x3_halo_bf16_kernel<1, float>), dim3((unsigned)tiles, 1), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3x3_halo_bf16_kernel<1, bf16_t>), dim3((unsigned)tiles, 1), dim3(256), 0, st, p);
    } else {
        return CVAR_EUNSUPPORTED;
    }
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}
