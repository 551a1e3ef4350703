// Parameters and the element-wise epilogue shared by the GEMM kernels (gemm.hip: LDS-tiled MFMA tiles + split-K; gemm_skinny.hip: the
// weight-streaming small-M kernel and the row-finishing split-K reduction).
#pragma once
#include "cvar_common.h"

struct GemmParams {
    int M, N, K;
    const char* A; long lda;
    const char* W; long ldw;
    long strideA, strideW, strideC, strideR;
    int conv, Hin, Win, Cin, Hout, Wout, stride, up;
    float alpha;
    const float* bias;
    int act;
    const float* gate; long ldg; int gate_rows;
    const void* residual; int res_dtype; long ldr;
    void* C; int out_dtype; long ldc;
    void* C2; const void* aux;     // optional: copy of alpha*acc+bias in the OPERAND dtype (before act / gate / residual) / gelu' operand (act GELU_GRAD); ld = ldc
    const float* gate_scale;       // optional per-gate-row multiplier (DropPath keep-scale of training)
    int in_dtype;                  // operand dtype (for the split-K epilogue kernel, which is not templated on it)
    int remap_l, remap_L, remap_off;
    float split_alpha;        // factor on the split columns [0, split_n) (1 = none)
    void* Cs; int split_n; long ld_split;       // column split: columns [0, split_n) -> Cs[m][ld_split] (rows not remapped), the rest -> C at column n - split_n
    int tiles_m, tiles_n;
    int cv_adv, cv_rem;       // conv: a K tile advances (tap, ci) by (KT / Cin, KT % Cin) plus one carry
    unsigned conv_bytes;      // conv FAST: bytes of the NHWC input of one batch slice (buffer range: out-of-range offsets read zeros)
    unsigned remap_magic, gate_magic; int remap_shift, gate_shift;   // exact m / remap_l and m / gate_rows for 0 <= m < 2^31 (fast_div)
    int split_tiles;          // split-K: K tiles per blockIdx.y slice (0 = no split); partials go to C + blockIdx.y * split_stride
    long split_stride;
    int group_m;              // row tiles per scheduling group (see launch_cfg)
    int stagger;              // > 0: the first wave of workgroups (one per CU) starts spread over this many shader cycles (see cvar_gemm_kernel)
    int tile_cfg;             // cvar_gemm_desc::tile_cfg (0 = automatic)
    int nt;                   // 1: the output streams (>= 128 MB): non-temporal stores in the specialised epilogues, RPF for the fp32 read-modify-write (set by cvar_gemm)
};


// floor(m / d) for 0 <= m < 2^31 as mulhi + shift: magic = ceil(2^(31+s) / d), 2^(s-1) < d <= 2^s (shift < 0 encodes d == 1)
__device__ __forceinline__ int fast_div(int m, unsigned magic, int shift) {
    const unsigned q = __umulhi((unsigned)m, magic) >> (shift < 0 ? 0 : shift);
    return shift < 0 ? m : (int)q;
}
static void make_fast_div(long d, unsigned* magic, int* shift) {
    if (d <= 1) { *magic = 0; *shift = -1; return; }
    int sft = 0;
    while ((1L << sft) < d) ++sft;                        // 2^(s-1) < d <= 2^s, s >= 1
    const unsigned long long num = 1ULL << (31 + sft);
    *magic = (unsigned)((num + (unsigned long long)d - 1) / (unsigned long long)d);
    *shift = sft - 1;                                      // mulhi drops 32 bits; 31 + s - 32 remain
}


// tile origin of virtual block id v: bijective XCD remap, then grouped-M ordering (shared by the kernel body and the fused split-K tail)
__device__ __forceinline__ void gemm_tile_origin(const GemmParams& p, int v, int BM, int BN, int& m0_, int& n0_) {
    const int nblk_all = p.tiles_m * p.tiles_n;
    const int xcd = v & 7, q = nblk_all >> 3, r = nblk_all & 7, local = v >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    const int GM = p.group_m;             // row tiles per group: the tiles of GM consecutive rows share each W tile out of L2
    const int group_sz = GM * p.tiles_n;
    const int grp = bid / group_sz, first_m = grp * GM;
    const int gm = min(p.tiles_m - first_m, GM);
    m0_ = (first_m + (bid % group_sz) % gm) * BM;
    n0_ = ((bid % group_sz) / gm) * BN;
}


// The complete epilogue of one output quad (row m, columns n .. n+3) on the fp32 sums v:
//   x = alpha * v + bias -> [pre_act copy] -> activation -> gate -> residual -> store (row remap / column split), cvar.h.
// Every path that finishes a GEMM outside the tile kernels' own epilogue (split-K reduction, skinny kernel) goes through here.
__device__ __forceinline__ void gemm_epilogue_quad(const GemmParams& p, int m, int n, f32x4_t v) {
    long orow = m;
    if (p.remap_l > 0) {
        const int sq = fast_div(m, p.remap_magic, p.remap_shift);
        orow = (long)sq * p.remap_L + p.remap_off + (m - sq * p.remap_l);
    }
    const float* grow = p.gate ? p.gate + (long)fast_div(m, p.gate_magic, p.gate_shift) * p.ldg : nullptr;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e] * p.alpha;
        if (p.bias) x += p.bias[n + e];
        if (p.C2) st_any(p.C2, p.in_dtype, (long)m * p.ldc + n + e, x);
        if (p.act == CVAR_ACT_GELU_TANH) x = gelu_tanh_f(x);
        else if (p.act == CVAR_ACT_GELU_GRAD) x *= gelu_tanh_grad(ld_any(p.aux, p.out_dtype, (long)m * p.ldc + n + e));
        if (grow) x *= grow[n + e] * (p.gate_scale ? p.gate_scale[fast_div(m, p.gate_magic, p.gate_shift)] : 1.0f);
        if (p.residual) x += ld_any(p.residual, p.res_dtype, (long)m * p.ldr + n + e);
        if (p.split_n > 0) {
            if (n < p.split_n) st_any(p.Cs, p.out_dtype, (long)m * p.ld_split + n + e, x * p.split_alpha);
            else st_any(p.C, p.out_dtype, orow * p.ldc + (n - p.split_n) + e, x);
        } else st_any(p.C, p.out_dtype, orow * p.ldc + n + e, x);
    }
}
