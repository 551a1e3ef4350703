// 3x3 stride-1 convolution of an image with <= 8 input channels (NHWC bf16, channels padded to 8) into 160-multiples of output channels: the VQVAE
// encoder's conv_in (3 -> 160 at 256^2, vae_modules.py:113 `self.conv_in = torch.nn.Conv2d(in_channels, self.ch, kernel_size=3, stride=1, padding=1)`).
//
// Why its own kernel (round 5).  K = 9 taps x 8 channels = 72: on the implicit-GEMM tile (gemm.hip, 128x160) the call is two half-empty K tiles whose DMA gathers
// 16 bytes per (pixel, tap) - 1.43 ms per 128 images for 2.7 GB of output (1.9 TB/s), and because only the LDS-halo kernel emits GroupNorm partials the GroupNorm
// behind it paid a statistics pass over the same 2.7 GB (0.5 ms).  The MFMA work is nothing (50 MFMA 32x32x16 per wave and tile); the kernel is its output stream:
//   * a workgroup owns 16x16 pixels x 160 couts, as conv_halo.hip: the 18x18 halo of 16-byte pixels (9 KB) and the weight block [160][10 taps][8] (28 KB, tap 9 = zeros)
//     are put in LDS once; k-step s of 16 = taps 2 s (lanes 0-31) and 2 s + 1 (lanes 32-63) - a lane's B fragment is ONE 16-byte halo pixel at the tap's shift;
//   * swapped MFMA operands (weights as the row operand): a lane ends up with its pixel's couts - the accumulator layout of conv_halo.hip, so the epilogue is that
//     kernel's row-major epilogue through LDS (16-byte full-line stores) including the GroupNorm partials of the stored output (cvar_gemm_desc.gn_part, ABI 18).
// LDS conflicts: halo rows are 32 pixels apart (512 B = 0 mod 256: the 16 lanes of a ds_read_b128 group - columns {0-3, 12-15} of one image row, {4-11} of the next -
// cover 16 distinct 16-byte slots for every tap shift); weight rows are 176 B = 11 slots apart (odd: any 16 couts distinct mod 16 hit 16 distinct slots).
// K order: tap-major in pairs, fp32 accumulate - differs from the implicit-GEMM order in rounding only.  An image's bits do not depend on its batch (one tile = one image).
#include "cvar_common.h"

struct ConvC8Params {
    const bf16_t* X; const bf16_t* Wt; const float* bias; bf16_t* out;
    int B, H, W, Cout, tiles_x, tiles_y;
    float* gn_part;            // optional: [B][tiles][Cout][3] = (sum (y - piv), sum (y - piv)^2, piv) per tile and channel (conv_halo.hip's format)
};

constexpr int C8_HROW = 32;                                   // halo row stride in pixels
constexpr int C8_HALO_BYTES = 18 * C8_HROW * 16;              // 9 216
constexpr int C8_WROW = 176;                                  // bytes per cout: 10 tap slots + 1 pad slot
constexpr int C8_W_BYTES = 160 * C8_WROW;                     // 28 160
constexpr int C8_PS = 336;                                    // epilogue: staged row of 160 bf16 + pad
constexpr int C8_SLICE = 15360;                               // epilogue LDS of one wave: 32 rows (10 752) + 512 (pivots behind wave 0's rows) + 60 x 64 B of partial sums
static_assert(32 * C8_PS + 512 + 3 * 20 * 64 <= C8_SLICE, "epilogue slice");
static_assert(C8_HALO_BYTES + C8_W_BYTES <= 4 * C8_SLICE, "the pipeline images fit the epilogue's LDS");

__global__ __launch_bounds__(256, 2) void conv3x3_c8_bf16_kernel(const ConvC8Params p) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * C8_SLICE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, hi = lane >> 5;
    int t_ = blockIdx.x;
    const int tx = t_ % p.tiles_x; t_ /= p.tiles_x;
    const int ty = t_ % p.tiles_y;
    const int b = t_ / p.tiles_y;
    const int ty0 = ty * 16, tx0 = tx * 16;
    const int cout0 = blockIdx.y * 160;
    char* const halo = smem;
    char* const wl = smem + C8_HALO_BYTES;

    // ---- halo (zeros outside the image = the conv's padding) and weight block -> LDS
    {
        const bf16_t* ximg = p.X + (long)b * p.H * p.W * 8;
        bf16x8_t hv[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int q = tid + 256 * r, hy = q / 18, hx = q - hy * 18;
            const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
            hv[r] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
            if (q < 324 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) hv[r] = *(const bf16x8_t*)(ximg + ((long)gy * p.W + gx) * 8);
        }
        const bf16_t* wg = p.Wt + (long)cout0 * 72;
        bf16x8_t wv[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const int q = tid + 256 * r, co = q / 10, tap = q - co * 10;
            wv[r] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
            if (q < 1600 && tap < 9) wv[r] = *(const bf16x8_t*)(wg + (co * 9 + tap) * 8);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int q = tid + 256 * r, hy = q / 18, hx = q - hy * 18;
            if (q < 324) *(bf16x8_t*)(halo + (hy * C8_HROW + hx) * 16) = hv[r];
        }
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const int q = tid + 256 * r, co = q / 10, tap = q - co * 10;
            if (q < 1600) *(bf16x8_t*)(wl + co * C8_WROW + tap * 16) = wv[r];
        }
    }
    __syncthreads();

    // ---- 5 k-steps of two taps: acc[i][j] = couts 32 j .. +31 x pixels of row block i (32 pixels = 2 image rows of the wave's 4)
    const int px = lrow & 15, py0 = 4 * wave + (lrow >> 4);
    f32x16_t acc[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int tap = 2 * s + hi;                       // tap 9 (upper half of the last step): zero weights; its pixels are read at tap 8's shift (finite values)
        const int ta = tap < 9 ? tap : 8;
        const int dy = ta / 3, dx = ta - 3 * dy;
        bf16x8_t a[2], w[5];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *(const bf16x8_t*)(halo + ((py0 + 2 * i + dy) * C8_HROW + px + dx) * 16);
#pragma unroll
        for (int j = 0; j < 5; ++j) w[j] = *(const bf16x8_t*)(wl + (32 * j + lrow) * C8_WROW + tap * 16);
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[j], a[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();                                      // every wave is done with the halo / weight images: the epilogue stages its rows over them

    // ---- epilogue (conv_halo.hip's wide form without residual): lane = pixel, register quad g of block j = couts 32 j + 8 g + 4 hi .. +3; the wave's 32 pixels x 160
    // couts of a row block pass through its LDS slice as [pixel][cout] rows and leave by 16-byte row-contiguous stores; GroupNorm partials of the values AS STORED
    float gs[8], gq[8], gpiv[8];
    char* const stg = smem + wave * C8_SLICE;
    const float* bp = p.bias ? p.bias + cout0 + 4 * hi : nullptr;
    const long img = (long)b * p.H * p.W * p.Cout;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        auto chunk = [&](int k, int& go, int& lo) {
            const int q = k * 64 + lane, pq = q / 20, part = q - pq * 20;
            go = ((ty0 + 4 * wave + 2 * i + (pq >> 4)) * p.W + tx0 + (pq & 15)) * p.Cout + cout0 + part * 8;
            lo = pq * C8_PS + part * 16;
        };
        char* mine = stg + lrow * C8_PS + 8 * hi;
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = 32 * j + 8 * g;
                f32x4_t bq = {0.f, 0.f, 0.f, 0.f};
                if (bp) bq = *(const f32x4_t*)(bp + co);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + bq[e];
                *(bf16x4_t*)(mine + co * 2) = pack_bf16x4(v);
            }
#pragma unroll
        for (int k = 0; k < 10; ++k) { int go, lo; chunk(k, go, lo); *(bf16x8_t*)(p.out + img + go) = *(const bf16x8_t*)(stg + lo); }
        if (p.gn_part) {
            char* pv = smem + 32 * C8_PS;                               // 160 bf16 pivots, behind wave 0's rows
            if (i == 0) {
                __builtin_amdgcn_wave_barrier();
                if (wave == 0 && lane < 20) *(bf16x8_t*)(pv + lane * 16) = *(const bf16x8_t*)(stg + lane * 16);       // the tile's first staged pixel
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
                        const bf16x8_t y8 = *(const bf16x8_t*)(stg + pp * C8_PS + cc * 16);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float f = bf16_to_f32((bf16_t)y8[e]) - gpiv[e]; gs[e] += f; gq[e] = __builtin_fmaf(f, f, gq[e]); }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();          // the next row block's rows overwrite the staged rows: reads first
        }
    }
    if (p.gn_part) {
        constexpr int RED = 32 * C8_PS + 512;
        if (lane < 60) {
            float* rd = (float*)(smem + wave * C8_SLICE + RED) + lane * 16;
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
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int pg = 0; pg < 3; ++pg) {
                    const float* rd = (const float*)(smem + w * C8_SLICE + RED) + (pg * 20 + cc) * 16;
                    S += rd[e]; Q += rd[8 + e];
                }
            const float piv = bf16_to_f32(*(const bf16_t*)(smem + 32 * C8_PS + tid * 2));
            float* gp = p.gn_part + ((((long)b * p.tiles_y + ty) * p.tiles_x + tx) * p.Cout + cout0 + tid) * 3;
            gp[0] = S; gp[1] = Q; gp[2] = piv;
        }
    }
}

// eligibility is checked by the caller (cvar_gemm): bf16 NHWC input with exactly 8 (padded) channels, packed [Cout][9][8] weights, Cout % 160 == 0, bf16 output,
// H % 16 == 0, W % 16 == 0, no residual / activation / gate; 16-byte aligned X, Wt, out (and bias); H * W * Cout < 2^31
int cvar_conv3x3_c8_bf16(const void* X, const void* Wt, const float* bias, void* out, int B, int H, int W, int Cout, float* gn_part, hipStream_t st) {
    if (B <= 0 || H <= 0 || W <= 0 || (H & 15) || (W & 15) || Cout <= 0 || Cout % 160) return CVAR_EUNSUPPORTED;
    if ((((uintptr_t)X | (uintptr_t)Wt | (uintptr_t)out | (uintptr_t)bias) & 15) != 0 || (long)H * W * Cout >= 0x7fffffffL) return CVAR_EUNSUPPORTED;
    ConvC8Params p;
    p.X = (const bf16_t*)X; p.Wt = (const bf16_t*)Wt; p.bias = bias; p.out = (bf16_t*)out;
    p.B = B; p.H = H; p.W = W; p.Cout = Cout; p.tiles_x = W / 16; p.tiles_y = H / 16; p.gn_part = gn_part;
    const long tiles = (long)B * p.tiles_x * p.tiles_y;
    if (tiles > 0x7fffffffL) return CVAR_EINVAL;
    hipLaunchKernelGGL(conv3x3_c8_bf16_kernel, dim3((unsigned)tiles, Cout / 160), dim3(256), 0, st, p);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}
