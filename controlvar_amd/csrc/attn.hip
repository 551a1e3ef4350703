// Multi-scale KV-cached attention over the qkv arena [R][Lmax][3*H*64], or over a K/V arena [R][Lmax][2*H*64] with the
// queries of the call in their own [R][l][H*64] buffer (inference: Q is dead after its scale, so it is not cached).
//
// attn_rowwise_kernel: exact-order fp32 row-per-lane kernel (VALU).  It is the parity-mode (CVAR_F32)
// implementation and the in-library reference the MFMA flash kernel is A/B-checked against.  One lane owns one
// query row (q, o in registers), K/V tiles of 64 keys are staged in LDS as fp32 and read as wave-wide
// broadcasts; block-wise online softmax (one rescale per 64-key tile).
#include "cvar_common.h"
#include <type_traits>

constexpr int CVAR_ATTN_MAX_LVL = 32;

struct AttnParams {
    const void* qkv;     // K / V rows: [R][Lmax][ldkv], K of head h at column k_col + 64 h, V at v_col + 64 h
    const void* q;       // query rows: [R][q_rows][ldq]; the row of position pos is pos - q_pos0 (packed arena: q == qkv, q_pos0 = 0)
    int ldkv, k_col, v_col, ldq, q_rows, q_pos0;
    void* out;
    int R, H, Lmax, q_off, l;
    float scale;
    int n_lvl;
    int lvl_end[CVAR_ATTN_MAX_LVL];
    int hole_lo[CVAR_ATTN_MAX_LVL], hole_hi[CVAR_ATTN_MAX_LVL];     // per level: keys [lo, hi) are invisible to its queries (empty: lo = hi = INT_MAX)
    float* lse;          // optional [R][H][l] log-sum-exp of the scaled scores (saved for the backward pass)
};

// Visibility of one query: keys [0, kvlen) minus the hole [hlo, hhi).  The prefix is the block-causal level mask of training
// (control_var.py:158-168) or, with half-scale levels, of separate_decoding (:170-180); the hole is the `indep` variant, where the image
// half of a scale does not see the control half of the same scale (:182-191).  Both bounds are non-decreasing in the query position.
struct Vis { int kvlen, hlo, hhi; };
template <typename P>
__device__ __forceinline__ Vis vis_of(const P& p, int pos) {
    Vis v;
    if (p.n_lvl == 0) { v.kvlen = p.q_off + p.l; v.hlo = v.hhi = 0x7fffffff; return v; }
    int k = p.n_lvl - 1;
    for (int i = p.n_lvl - 1; i >= 0; --i)
        if (pos < p.lvl_end[i]) k = i;
    v.kvlen = p.lvl_end[k]; v.hlo = p.hole_lo[k]; v.hhi = p.hole_hi[k];
    return v;
}
__device__ __forceinline__ bool vis_key(const Vis& v, int key) { return key < v.kvlen && !(key >= v.hlo && key < v.hhi); }
// HOLES = false: the launch has no level with a hole (every default model) - the test and its operands compile away
template <bool HOLES>
__device__ __forceinline__ bool vis_key_t(const Vis& v, int key) { return key < v.kvlen && (!HOLES || !(key >= v.hlo && key < v.hhi)); }
__device__ __forceinline__ int vis_full_prefix(const Vis& v) { return min(v.kvlen, v.hlo); }       // keys below this are all visible

// keys below the returned bound are visible to EVERY query at positions [pos_lo, pos_hi]: the minimum of min(end, hole start) over the
// levels the range touches (not monotone in the position: a control half ends after the hole of the image half that follows it)
template <typename P>
__device__ __forceinline__ int range_full_prefix(const P& p, int pos_lo, int pos_hi) {
    if (p.n_lvl == 0) return p.q_off + p.l;
    int best = 0x7fffffff, begin = 0;
    for (int i = 0; i < p.n_lvl; ++i) {
        const int end = p.lvl_end[i];
        const bool last = i == p.n_lvl - 1;
        if (pos_lo < end || last) {                       // level i holds positions [begin, end) (the last level also everything behind it)
            if (pos_hi >= begin || last) best = min(best, min(end, p.hole_lo[i]));
        }
        begin = end;
        if (pos_hi < end) break;
    }
    return best;
}

__device__ __forceinline__ int kv_len_of(const AttnParams& p, int pos) { return vis_of(p, pos).kvlen; }

template <typename T>
__global__ __launch_bounds__(256) void attn_rowwise_kernel(const AttnParams p) {
    constexpr int D = 64, KT = 64;
    constexpr int VEC = 16 / sizeof(T);
    __shared__ __attribute__((aligned(16))) float Ks[KT][D];
    __shared__ __attribute__((aligned(16))) float Vs[KT][D];
    const int tid = threadIdx.x;
    const int h = blockIdx.y;
    const long r = blockIdx.z;
    const int C3 = p.ldkv;
    const T* base = (const T*)p.qkv + r * (long)p.Lmax * C3;
    const int qi = blockIdx.x * 256 + tid;
    const bool valid = qi < p.l;
    const int pos = p.q_off + (valid ? qi : p.l - 1);
    const Vis vis = vis_of(p, pos);
    const int last_q = min(p.l, (int)(blockIdx.x + 1) * 256) - 1;
    const int kv_end = kv_len_of(p, p.q_off + last_q);          // monotone in pos -> block maximum

    float q[D], o[D];
    {
        const T* qp = (const T*)p.q + (r * p.q_rows + (pos - p.q_pos0)) * (long)p.ldq + h * D;
#pragma unroll
        for (int d = 0; d < D; ++d) { q[d] = Elem<T>::ld(qp + d) * p.scale; o[d] = 0.f; }
    }
    float m = -INFINITY, lsum = 0.f;

    for (int kt0 = 0; kt0 < kv_end; kt0 += KT) {
        // cooperative K/V tile load -> fp32 LDS
        for (int v = tid; v < KT * D / VEC; v += 256) {
            const int kk = v / (D / VEC), d0 = (v % (D / VEC)) * VEC;
            const int key = kt0 + kk;
            if (key < kv_end) {
                const T* kp = base + (long)key * C3 + p.k_col + h * D + d0;
                const T* vp = base + (long)key * C3 + p.v_col + h * D + d0;
#pragma unroll
                for (int e = 0; e < VEC; ++e) { Ks[kk][d0 + e] = Elem<T>::ld(kp + e); Vs[kk][d0 + e] = Elem<T>::ld(vp + e); }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) { Ks[kk][d0 + e] = 0.f; Vs[kk][d0 + e] = 0.f; }
            }
        }
        __syncthreads();
        float s[KT];
        float tmax = -INFINITY;
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < D; d += 4) {
                const f32x4_t kv = *(const f32x4_t*)&Ks[kk][d];
                a = fmaf(q[d], kv[0], a); a = fmaf(q[d + 1], kv[1], a);
                a = fmaf(q[d + 2], kv[2], a); a = fmaf(q[d + 3], kv[3], a);
            }
            s[kk] = vis_key(vis, kt0 + kk) ? a : -INFINITY;
            tmax = fmaxf(tmax, s[kk]);
        }
        if (tmax > -INFINITY) {
            const float m_new = fmaxf(m, tmax);
            const float alpha = __expf(m - m_new);              // m = -inf -> 0
            lsum *= alpha;
#pragma unroll
            for (int d = 0; d < D; ++d) o[d] *= alpha;
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {
                float pr = __expf(s[kk] - m_new);
                lsum += pr;
                if constexpr (sizeof(T) == 2) pr = bf16_to_f32(f32_to_bf16(pr));   // P is a bf16 MFMA operand in bf16 mode
#pragma unroll
                for (int d = 0; d < D; d += 4) {
                    const f32x4_t vv = *(const f32x4_t*)&Vs[kk][d];
                    o[d] = fmaf(pr, vv[0], o[d]); o[d + 1] = fmaf(pr, vv[1], o[d + 1]);
                    o[d + 2] = fmaf(pr, vv[2], o[d + 2]); o[d + 3] = fmaf(pr, vv[3], o[d + 3]);
                }
            }
            m = m_new;
        }
        __syncthreads();
    }
    if (valid) {
        const float inv = 1.0f / lsum;
        T* op = (T*)p.out + (r * p.l + qi) * (long)(p.H * D) + h * D;
#pragma unroll
        for (int d = 0; d < D; ++d) Elem<T>::st(op + d, o[d] * inv);
        if (p.lse) p.lse[(r * p.H + h) * (long)p.l + qi] = m + logf(lsum);
    }
}

// (a template takes its launch bounds from the FIRST declaration: without them here the kernels are compiled for 1024-thread groups, 128 VGPRs)
template <bool HOLES, bool QPRE> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_mfma_bf16_kernel(const AttnParams p);
template <bool HOLES> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_mfma_bf16_v1_kernel(const AttnParams p);
template <bool HOLES> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_mfma_bf16_q64_kernel(const AttnParams p);
#ifndef CVAR_ATTN_Q64
#define CVAR_ATTN_Q64 1
#endif
#ifndef CVAR_ATTN_Q64_MINL
#define CVAR_ATTN_Q64_MINL 192      // l = 200 (one workgroup of 256 instead of two of 128): 0.315 -> 0.308 ms per call at B = 128
#endif
// (round-4 experiments - the wave-internally pipelined form, K / V tiles by LDS-DMA, the timing ablations - are not in this file any more:
// experiments/README.md lists the commits that hold them and the profiles/ records with their numbers)

// impl: 0 = auto (MFMA flash kernel for bf16, row-wise exact kernel for fp32), 1 = row-wise
template <typename P>
static int fill_levels(P& p, const int* lvl_end_host, int n_lvl, const int* hole_host) {
    if (n_lvl < 0 || n_lvl > CVAR_ATTN_MAX_LVL || (n_lvl > 0 && !lvl_end_host)) return CVAR_EINVAL;
    p.n_lvl = n_lvl;
    int prev = 0;
    for (int i = 0; i < CVAR_ATTN_MAX_LVL; ++i) {
        p.lvl_end[i] = i < n_lvl ? lvl_end_host[i] : 0;
        p.hole_lo[i] = p.hole_hi[i] = 0x7fffffff;
        if (i < n_lvl) {
            if (p.lvl_end[i] <= prev) return CVAR_EINVAL;                   // strictly increasing level ends
            if (hole_host && hole_host[2 * i + 1] > hole_host[2 * i]) {
                // a hole lies inside the keys its level sees and in front of the level's own tokens (they always see themselves)
                if (hole_host[2 * i] < 0 || hole_host[2 * i + 1] > prev) return CVAR_EINVAL;
                p.hole_lo[i] = hole_host[2 * i]; p.hole_hi[i] = hole_host[2 * i + 1];
            }
            prev = p.lvl_end[i];
        }
    }
    return CVAR_OK;
}

static int cvar_attention_impl(const void* qkv, const void* q, int dtype, int R, int H, int Lmax, int q_off, int l, float scale,
                               const int* lvl_end_host, int n_lvl, const int* hole_host, void* out, float* lse, void* stream, int impl) {
    if (!qkv || !out || R <= 0 || H <= 0 || l <= 0 || q_off < 0 || q_off + l > Lmax) return CVAR_EINVAL;
    AttnParams p;
    const int C = H * 64;
    if (q) {            // K/V arena [R][Lmax][2C] + this call's queries [R][l][C]
        p.q = q; p.ldq = C; p.q_rows = l; p.q_pos0 = q_off; p.ldkv = 2 * C; p.k_col = 0; p.v_col = C;
    } else {            // packed [R][Lmax][3C] (q | k | v thirds)
        p.q = qkv; p.ldq = 3 * C; p.q_rows = Lmax; p.q_pos0 = 0; p.ldkv = 3 * C; p.k_col = C; p.v_col = 2 * C;
    }
    p.qkv = qkv; p.out = out; p.R = R; p.H = H; p.Lmax = Lmax; p.q_off = q_off; p.l = l; p.scale = scale; p.lse = lse;
    { const int rc = fill_levels(p, lvl_end_host, n_lvl, hole_host); if (rc != CVAR_OK) return rc; }
    const bool qpre = (impl & 4) != 0;                     // the query rows carry scale * log2(e) (cvar_attention_prescaled)
    impl &= 3;
    if (qpre && (dtype != CVAR_BF16 || impl == 1 || !q)) return CVAR_EUNSUPPORTED;
    if (dtype == CVAR_BF16 && impl != 1) {
        bool holes = false;
        for (int i = 0; i < p.n_lvl; ++i) holes = holes || p.hole_lo[i] < p.hole_hi[i];
        if (impl == 2) {                                   // round-2 kernel (A/B reference)
            if (qpre) return CVAR_EUNSUPPORTED;
            if (holes) hipLaunchKernelGGL(attn_mfma_bf16_v1_kernel<true>, dim3(cdiv(l, 128), H, R), dim3(256), 0, as_stream(stream), p);
            else hipLaunchKernelGGL(attn_mfma_bf16_v1_kernel<false>, dim3(cdiv(l, 128), H, R), dim3(256), 0, as_stream(stream), p);
            CVAR_CHECK_LAUNCH();
            return CVAR_OK;
        }
        const long nblk = (long)cdiv(l, 128) * H * R;
        if (nblk > 0x7fffffffL) return CVAR_EUNSUPPORTED;
        const dim3 grid((unsigned)nblk), block(256);
        // prescaled queries, long scales: 64 queries per wave (K / V fragments, tiles and barriers shared by two query groups)
        // (only where the last 256-query workgroup is nearly full: l = 512 runs 1.371 -> 1.285 ms per call at B = 128, l = 338 - 82 queries in its second workgroup -
        //  0.685 -> 0.82 ms and stays on the 128-query kernel; profiles/r04_attn_ablation.txt)
        // and only where the halved workgroup count still fills the chip twice over (a B = 1 generation has 48 (row, head) pairs)
        if (CVAR_ATTN_Q64 && qpre && l >= CVAR_ATTN_Q64_MINL && cdiv(l, 256) * 256 - l < 64 && (long)cdiv(l, 256) * H * R >= 512) {
            const dim3 grid64((unsigned)((long)cdiv(l, 256) * H * R));
            if (holes) hipLaunchKernelGGL((attn_mfma_bf16_q64_kernel<true>), grid64, block, 0, as_stream(stream), p);
            else hipLaunchKernelGGL((attn_mfma_bf16_q64_kernel<false>), grid64, block, 0, as_stream(stream), p);
            CVAR_CHECK_LAUNCH();
            return CVAR_OK;
        }
        if (holes) { if (qpre) hipLaunchKernelGGL((attn_mfma_bf16_kernel<true, true>), grid, block, 0, as_stream(stream), p);
                     else hipLaunchKernelGGL((attn_mfma_bf16_kernel<true, false>), grid, block, 0, as_stream(stream), p); }
        else { if (qpre) hipLaunchKernelGGL((attn_mfma_bf16_kernel<false, true>), grid, block, 0, as_stream(stream), p);
               else hipLaunchKernelGGL((attn_mfma_bf16_kernel<false, false>), grid, block, 0, as_stream(stream), p); }
        CVAR_CHECK_LAUNCH();
        return CVAR_OK;
    }
    dim3 grid(cdiv(l, 256), H, R), block(256);
    if (dtype == CVAR_BF16) hipLaunchKernelGGL(attn_rowwise_kernel<bf16_t>, grid, block, 0, as_stream(stream), p);
    else if (dtype == CVAR_F32) hipLaunchKernelGGL(attn_rowwise_kernel<float>, grid, block, 0, as_stream(stream), p);
    else return CVAR_EUNSUPPORTED;
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// ================================================================================================
// attn_mfma_bf16_v1_kernel (the round-2 kernel, kept as the A/B reference of the round-3 one: cvar_attention_impl impl 2): flash-style attention on the matrix cores (bf16 in, fp32 accumulate), head_dim 64.
//
// Per workgroup: 4 waves x 32 queries; KV tiles of 64 keys.  Per wave and tile (16 MFMA 32x32x16):
//   S^T = K . Q^T   (swapped product: each lane then holds 32 scores of ONE query -> the row max / row sum are
//                    31 in-lane ops + one cross-half shuffle, no LDS)
//   P   = exp(S*scale - m)  in fp32, packed to bf16 straight into the B-operand layout of the next product
//   O^T += V^T . P^T        (V is transposed on its way into LDS: two keys packed per ds_write_b32,
//                            rows of 68 elements -> conflict-free b64 fragment reads)
// K tile: row-major 128-B rows with the 16-B chunk index XOR-swizzled by (key>>1)&7 (conflict-free ds_read_b128).
// Global -> register prefetch of tile i+1 is issued before the MFMA work on tile i and written to LDS after it.
// ================================================================================================
constexpr int FA_VT_STRIDE = 68;      // elements per V^T row (64 keys + pad): 136 B, 8-B aligned, 2-way-free writes

// 3 waves per SIMD (<= 168 VGPRs): the softmax phase of one wave overlaps the MFMA phases of the other two (+10 % over 2)
template <bool HOLES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_mfma_bf16_v1_kernel(const AttnParams p) {
    constexpr int D = 64, KT = 64;
    __shared__ __attribute__((aligned(16))) char Ks[KT * 128];
    __shared__ __attribute__((aligned(16))) bf16_t Vt[D * FA_VT_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lrow = lane & 31, hi = lane >> 5, sw = (lane >> 1) & 7;
    const int h = blockIdx.y;
    const long r = blockIdx.z;
    const int C3 = p.ldkv;
    const bf16_t* base = (const bf16_t*)p.qkv + r * (long)p.Lmax * C3;
    const bf16_t* kbase = base + p.k_col + h * D;
    const bf16_t* vbase = base + p.v_col + h * D;

    const int q0 = blockIdx.x * 128 + w * 32;
    const int qi = q0 + lrow;
    const int qrow = min(qi, p.l - 1);
    const Vis vis = vis_of(p, p.q_off + qrow);
    const int kv_end = kv_len_of(p, p.q_off + min(p.l, (int)(blockIdx.x + 1) * 128) - 1);

    bf16x8_t qf[4];
    {
        const bf16_t* qp = (const bf16_t*)p.q + (r * p.q_rows + (p.q_off + qrow - p.q_pos0)) * (long)p.ldq + h * D;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8_t*)(qp + (2 * ks + hi) * 8);
    }
    // staging maps
    const int k_key = tid >> 3, k_chunk = tid & 7;                       // K: keys k_key and k_key+32, chunk k_chunk
    const int v_kp = 16 * (w >> 1) + (lane & 15), v_chunk = 4 * (w & 1) + (lane >> 4);   // V: keys 2*v_kp, 2*v_kp+1
    bf16x8_t kreg[2], vreg[2];
    auto load_tile = [&](int kt0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = kt0 + k_key + 32 * i;
            bf16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
            kreg[i] = key < kv_end ? *(const bf16x8_t*)(kbase + (long)key * C3 + k_chunk * 8) : z;
            const int vkey = kt0 + 2 * v_kp + i;
            vreg[i] = vkey < kv_end ? *(const bf16x8_t*)(vbase + (long)vkey * C3 + v_chunk * 8) : z;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = k_key + 32 * i;
            *(bf16x8_t*)(Ks + key * 128 + ((k_chunk ^ ((key >> 1) & 7)) << 4)) = kreg[i];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned packed = (unsigned)(unsigned short)vreg[0][e] | ((unsigned)(unsigned short)vreg[1][e] << 16);
            *(unsigned*)(Vt + (v_chunk * 8 + e) * FA_VT_STRIDE + 2 * v_kp) = packed;
        }
    };

    f32x16_t o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    float m = -INFINITY;
    const float c2 = p.scale * 1.4426950408889634f;
    float lsum = 0.f;

    // One KV tile.  MASK is a compile-time flag: tiles that every query of the wave sees completely (all but the last one at
    // inference, all but the level-boundary ones under the training mask) run without any per-score compare / select - left
    // as a run-time flag the compiler if-converts the masking into ~100 extra vector instructions on every tile.
    auto tile = [&](int kt0, auto MASK) {
        store_tile();
        __syncthreads();
        if (kt0 + KT < kv_end) load_tile(kt0 + KT);
        // ---- S^T = K Q^T
        f32x16_t s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8_t kf = *(const bf16x8_t*)(Ks + (32 * kb + lrow) * 128 + (((2 * ks + hi) ^ sw) << 4));
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
            }
        }
        // softmax bookkeeping in the exp2 domain: p = 2^(s*c - m), c = scale*log2(e)  (one fma + one v_exp_f32 per score)
        float tmax = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (decltype(MASK)::value) {
                    const int key = kt0 + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * hi;
                    if (!vis_key_t<HOLES>(vis, key)) s[kb][i] = -INFINITY;
                }
                tmax = fmaxf(tmax, s[kb][i]);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m, tmax * c2);            // c2 > 0; finite from the first tile on (key 0 is always visible)
        if (!__all(m_new == m)) {                           // rescale only when some row's running max moved (exact skip)
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            lsum *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
            m = m_new;
        }
        bf16x8_t pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float pr[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    pr[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][8 * t + j], c2, -m));
                    lsum += pr[j];
                }
                pf[kb][t] = pack_bf16x8(pr);
            }
        // ---- O^T += V^T P^T
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16_t* vp = Vt + (32 * db + lrow) * FA_VT_STRIDE + 32 * kb + 16 * t + 4 * hi;
                    const bf16x4_t v0 = *(const bf16x4_t*)vp;
                    const bf16x4_t v1 = *(const bf16x4_t*)(vp + 8);
                    const bf16x8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][t], o[db], 0, 0, 0);
                }
        __syncthreads();
    };
    typedef std::integral_constant<bool, true> MaskOn;
    typedef std::integral_constant<bool, false> MaskOff;
    load_tile(0);
    int kt0 = 0;
    // wave_min_kv is wave-uniform per construction but tiles are shared by the workgroup (barriers inside): the split point
    // must be the same for all four waves, so it is taken over the workgroup's first query
    const int wg_min_kv = range_full_prefix(p, p.q_off + min((int)blockIdx.x * 128, p.l - 1), p.q_off + min(p.l, (int)(blockIdx.x + 1) * 128) - 1);
    for (; kt0 + KT <= wg_min_kv && kt0 < kv_end; kt0 += KT) tile(kt0, MaskOff{});
    for (; kt0 < kv_end; kt0 += KT) tile(kt0, MaskOn{});
    lsum += __shfl_xor(lsum, 32, 64);
    if (qi < p.l) {
        if (p.lse && hi == 0) p.lse[(r * p.H + h) * (long)p.l + qi] = (m + log2f(lsum)) * 0.6931471805599453f;
        const float inv = 1.0f / lsum;
        bf16_t* op = (bf16_t*)p.out + (r * p.l + qi) * (long)(p.H * D) + h * D;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float ov[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = o[db][4 * g + e] * inv;
                *(bf16x4_t*)(op + 32 * db + 8 * g + 4 * hi) = pack_bf16x4(ov);
            }
    }
}

// ================================================================================================
// attn_mfma_bf16_kernel: flash-style attention on the matrix cores (bf16 in, fp32 accumulate), head_dim 64.
//
// Per workgroup: 4 waves x 32 queries of one (row, head); KV tiles of 64 keys.  Per wave and tile:
//   S^T = K . Q^T   (swapped product: each lane then holds 32 scores of ONE query -> the row max / row sum are
//                    31 in-lane ops + one cross-half shuffle, no LDS)
//   P   = 2^(...)   in fp32, packed to bf16 straight into the B-operand layout of the next product
//   O^T += V^T . P^T
// K tile: row-major 128-B rows with the 16-B chunk index XOR-swizzled by (key>>1)&7 (conflict-free ds_read_b128).
// V tile (round 3): row-major as it lies in memory (two ds_write_b128 per thread, no packing), 32-B segments XOR-swizzled by key & 3;
//   the V^T fragments are built by `ds_read_b64_tr_b16` transpose-reads (a [4 keys][16 d] block -> lane i gets column i).  Round 2
//   transposed V on its way INTO LDS: 8 pack operations + 8 ds_write_b32 per thread and tile, all of them on the vector pipe that
//   bounds this kernel.
// Global -> register prefetch of tile i+1 is issued before the MFMA work on tile i and written to LDS after it.
//
// QPRE (inference, cvar_attention_prescaled): the query rows already carry scale * log2(e) (folded into the q columns of the QKV GEMM's
//   epilogue, one rounding), so a score needs no multiply; the running maximum is subtracted ON THE MATRIX PIPE: a fifth k-step whose A
//   operand is a column of ones and whose B operand holds -m~ per query (m~ = the running maximum rounded to bf16 - softmax is invariant
//   under ANY per-query shift, so the rounding is exact mathematics, it only has to stay within a few units of the true maximum).  The
//   accumulators then hold s - m~ and P = v_exp_f32 of them: per score one transcendental + one add (row sum) + 1/2 max3 + 1/2 cvt_pk
//   instead of those plus one fma - the vector pipe is what bounds this kernel (VALU-busy 81 %, MFMA-busy 32 % in round 2), the matrix
//   pipe has room for the 2 extra MFMAs per tile.  m~ moves only when a tile's maximum exceeds it by more than THR (= 2 + |m~|/64 in the
//   log2 domain: P <= 4 ... ), then the tile's scores, O and the row sum are shifted by the same exact delta.
// Workgroup -> (query block, head, row): 1-D grid; the query blocks of one (row, head) - which stream the same K/V - are given block
//   ids congruent mod 8, i.e. the same XCD / L2, and adjacent in dispatch order.
// ================================================================================================
typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4_t lds_tr16_b64(const char* lds_ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)lds_ptr);
}

// 3 waves per SIMD (<= 168 VGPRs): the softmax phase of one wave overlaps the MFMA phases of the other two (+10 % over 2)
template <bool HOLES, bool QPRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_mfma_bf16_kernel(const AttnParams p) {
    constexpr int D = 64, KT = 64;
    // one K and one V tile, two barriers per tile.  (Two buffers and one barrier per tile: 7 % SLOWER - profiles/r03_attn_ab3_lds_double_buffer_rejected.txt.)
    constexpr int TILE_B = KT * 128;
    __shared__ __attribute__((aligned(1024))) char KVs[2 * TILE_B];
    char* const Ks = KVs;
    char* const Vs = KVs + TILE_B;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lrow = lane & 31, hi = lane >> 5, sw = (lane >> 1) & 7;
    // ---- block id -> (query block, pair = row * H + head)
    const int nqb = (p.l + 127) >> 7, pairs = p.R * p.H;
    int qb, pair;
    {
        const int id = blockIdx.x;
        if ((pairs & 7) == 0) { const int xcd = id & 7, local = id >> 3; qb = local % nqb; pair = (local / nqb) * 8 + xcd; }
        else { qb = id % nqb; pair = id / nqb; }
    }
    const int h = pair % p.H;
    const long r = pair / p.H;
    const int C3 = p.ldkv;
    const bf16_t* base = (const bf16_t*)p.qkv + r * (long)p.Lmax * C3;
    const bf16_t* kbase = base + p.k_col + h * D;
    const bf16_t* vbase = base + p.v_col + h * D;

    const int q0 = qb * 128 + w * 32;
    const bool active = q0 < p.l;                     // a wave without a single query of the scale only helps staging the tiles
    const int qi = q0 + lrow;
    const int qrow = min(qi, p.l - 1);
    const Vis vis = vis_of(p, p.q_off + qrow);
    const int kv_end = kv_len_of(p, p.q_off + min(p.l, (qb + 1) * 128) - 1);

    bf16x8_t qf[4];
    {
        const bf16_t* qp = (const bf16_t*)p.q + (r * p.q_rows + (p.q_off + qrow - p.q_pos0)) * (long)p.ldq + h * D;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8_t*)(qp + (2 * ks + hi) * 8);
    }
    // staging map (K and V alike): thread -> keys k_key and k_key + 32, 16-byte chunk k_chunk of the 128-byte head row
    const int k_key = tid >> 3, k_chunk = tid & 7;
    bf16x8_t kreg[2], vreg[2];
    // buffer loads: the resource covers rows [0, kv_end) of this (row, head)'s K (V) columns, a lane's offset inside a tile never changes
    // and the tile's base is a scalar - no vector address arithmetic, no predication (rows past kv_end come back as zeros)
    const int row_bytes = C3 * 2;
    const int rec = (kv_end - 1) * row_bytes + 128;
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, rec, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, rec, 0x00020000);
    int ld_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) ld_off[i] = (k_key + 32 * i) * row_bytes + k_chunk * 16;
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    auto load_tile = [&](int kt0) {
        const int so = kt0 * row_bytes;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            kreg[i] = __builtin_bit_cast(bf16x8_t, (v4i_t)__builtin_amdgcn_raw_buffer_load_b128(k_rsrc, ld_off[i], so, 0));
            vreg[i] = __builtin_bit_cast(bf16x8_t, (v4i_t)__builtin_amdgcn_raw_buffer_load_b128(v_rsrc, ld_off[i], so, 0));
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = k_key + 32 * i;
            *(bf16x8_t*)(Ks + key * 128 + ((k_chunk ^ ((key >> 1) & 7)) << 4)) = kreg[i];
            *(bf16x8_t*)(Vs + key * 128 + ((((k_chunk >> 1) ^ (key & 3)) << 5) | ((k_chunk & 1) << 4))) = vreg[i];
        }
    };
    // transpose-read addressing of V: lane l of a 16-lane group points at key row (l & 15) >> 2 of a [4 keys][16 d] block, d columns
    // 4 (l & 3) .. +3; the two groups of a half-wave are the two 16-d halves of a 32-d MFMA block, the upper half-wave takes the keys
    // 4 further (the key order of an 8-key fragment is 4 hi + {0..3}, 8 + 4 hi + {0..3} - the order the swapped QK^T leaves P in)
    const int v_jrow = (lane & 15) >> 2, v_g = (lane >> 4) & 1;
    const char* v_lane[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) v_lane[db] = Vs + (4 * hi + v_jrow) * 128 + (((2 * db + v_g) ^ v_jrow) << 5) + (lane & 3) * 8;

    f32x16_t o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    float m = QPRE ? 0.f : -INFINITY;                      // QPRE: m~ (a bf16 value), the shift the bias k-step applies
    const float c2 = p.scale * 1.4426950408889634f;
    float lsum = 0.f;
    // QPRE: operands of the bias k-step.  A: k index 0 is a column of ones (lanes hi == 0, element 0), B: k index 0 holds -m~ of the lane's query
    const short one_bf = hi == 0 ? (short)0x3f80 : (short)0;
    const bf16x8_t k_ones = {one_bf, 0, 0, 0, 0, 0, 0, 0};
    bf16x8_t q_m = {0, 0, 0, 0, 0, 0, 0, 0};

    // One KV tile.  MASK is a compile-time flag: tiles that every query of the wave sees completely (all but the last one at
    // inference, all but the level-boundary ones under the training mask) run without any per-score compare / select - left
    // as a run-time flag the compiler if-converts the masking into ~100 extra vector instructions on every tile.
    auto tile = [&](int kt0, auto MASK, auto FIRST) {
        store_tile();
        __syncthreads();
        if (kt0 + KT < kv_end) load_tile(kt0 + KT);
        if (active) {
        // ---- S^T = K Q^T (+ the bias k-step: - m~), invisible keys -> -inf
        f32x16_t s[2];
        auto scores = [&]() {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
                if constexpr (QPRE && !decltype(FIRST)::value) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k_ones, q_m, s[kb], 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8_t kf = *(const bf16x8_t*)(Ks + (32 * kb + lrow) * 128 + (((2 * ks + hi) ^ sw) << 4));
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
                }
            }
            if constexpr (decltype(MASK)::value) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = kt0 + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * hi;
                        if (!vis_key_t<HOLES>(vis, key)) s[kb][i] = -INFINITY;
                    }
            }
        };
        auto row_max = [&]() {
            float tmax = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, s[kb][i]);
            return fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        };
        bf16x8_t pf[2][2];
        scores();
        if constexpr (QPRE) {
            // The accumulators hold s - m~ (log2 domain) and P = 2^that.  m~ is set by the first tile (its exact maximum, rounded to bf16) and
            // moves later only when a tile's maximum exceeds it by more than 2 + |m~|/64; everything still at the old shift - the tile's
            // scores, O, the row sum - then moves by the same delta = m~_new - m~_old (both bf16 values: the fp32 difference is exact).
            auto shift = [&](bool always) {
                const float tmax = row_max();
                const bool need = always || tmax > 2.0f + fabsf(m) * 0.015625f;
                const float m_new = need ? bf16_to_f32(f32_to_bf16(m + tmax)) : m;
                const float delta = m_new - m;
                if (!always) {           // the first tile has nothing to rescale (O = row sum = 0) - and its delta may be hugely NEGATIVE (every score
                                         // far below zero): 2^-delta would be +inf and 0 * inf a NaN.  Later deltas are >= 0: alpha <= 1.
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    lsum *= alpha;
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) s[kb][i] -= delta;
                m = m_new;
                q_m[0] = hi == 0 ? (short)f32_to_bf16(-m_new) : (short)0;
            };
            // (Tried and rejected, profiles/r03_attn_ab4_optimistic_max_rejected.txt: no maximum at all on the hot path - P is bounded by the
            // tile's row sum, a wave-uniform test of the sum sends the rare tile through a careful second pass.  16 v_max3 fewer per tile,
            // but the second copy of the tile body costs registers (spills around the loops) and the short scales lose 20-30 %.)
            if (decltype(FIRST)::value) shift(true);
            else if (__any(row_max() > 2.0f + fabsf(m) * 0.015625f)) shift(false);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float pr[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        pr[j] = __builtin_amdgcn_exp2f(s[kb][8 * t + j]);
                        lsum += pr[j];
                    }
                    pf[kb][t] = pack_bf16x8(pr);
                }
        } else {
            // softmax bookkeeping in the exp2 domain: p = 2^(s*c - m), c = scale*log2(e)  (one fma + one v_exp_f32 per score)
            const float tmax = row_max();
            const float m_new = fmaxf(m, tmax * c2);            // c2 > 0; finite from the first tile on (key 0 is always visible)
            if (!__all(m_new == m)) {                           // rescale only when some row's running max moved (exact skip)
                const float alpha = __builtin_amdgcn_exp2f(m - m_new);
                lsum *= alpha;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
                m = m_new;
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float pr[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        pr[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][8 * t + j], c2, -m));
                        lsum += pr[j];
                    }
                    pf[kb][t] = pack_bf16x8(pr);
                }
        }
        // ---- O^T += V^T P^T
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const char* vp = v_lane[db] + (32 * kb + 16 * t) * 128;
                    const s16x4_t v0 = lds_tr16_b64(vp);
                    const s16x4_t v1 = lds_tr16_b64(vp + 8 * 128);
                    const bf16x8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][t], o[db], 0, 0, 0);
                }
        }
        __syncthreads();
    };
    typedef std::integral_constant<bool, true> Yes;
    typedef std::integral_constant<bool, false> No;
    load_tile(0);
    int kt0 = 0;
    // wave_min_kv is wave-uniform per construction but tiles are shared by the workgroup (barriers inside): the split point
    // must be the same for all four waves, so it is taken over the workgroup's first query
    const int wg_min_kv = range_full_prefix(p, p.q_off + min(qb * 128, p.l - 1), p.q_off + min(p.l, (qb + 1) * 128) - 1);
    if constexpr (QPRE) {                                  // the first tile sets m~ unconditionally
        if (KT <= wg_min_kv) tile(0, No{}, Yes{}); else tile(0, Yes{}, Yes{});
        kt0 = KT;
    }
    for (; kt0 + KT <= wg_min_kv && kt0 < kv_end; kt0 += KT) tile(kt0, No{}, No{});
    for (; kt0 < kv_end; kt0 += KT) tile(kt0, Yes{}, No{});
    lsum += __shfl_xor(lsum, 32, 64);
    if (qi < p.l) {
        if (p.lse && hi == 0) p.lse[(r * p.H + h) * (long)p.l + qi] = (m + log2f(lsum)) * 0.6931471805599453f;
        const float inv = 1.0f / lsum;
        bf16_t* op = (bf16_t*)p.out + (r * p.l + qi) * (long)(p.H * D) + h * D;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float ov[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = o[db][4 * g + e] * inv;
                *(bf16x4_t*)(op + 32 * db + 8 * g + 4 * hi) = pack_bf16x4(ov);
            }
    }
}


// ================================================================================================
// attn_mfma_bf16_q64_kernel (round 4, second half; inference form with prescaled queries, scales of >= 256 queries): the QPRE kernel above with TWO 32-query
// groups per wave - a workgroup owns 256 queries of one (row, head).  Every K fragment read out of LDS, every transposed V fragment, every K / V tile a CU
// pulls and every barrier now serve 64 queries of the wave instead of 32: the ablation table (profiles/r04_attn_ablation.txt) puts the tile traffic at 23 %
// and the kernel at ~13 B per clock and CU.  Costs ~100 registers: two waves per SIMD instead of three.  Same arithmetic per query as the kernel above
// (same tile order, same shift rule per 32-query group): bit-identical output.
// ================================================================================================
template <bool HOLES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_mfma_bf16_q64_kernel(const AttnParams p) {
    constexpr int D = 64, KT = 64, G = 2;
    constexpr int TILE_B = KT * 128;
    __shared__ __attribute__((aligned(16))) char KVs[2 * TILE_B];
    char* const Ks = KVs;
    char* const Vs = KVs + TILE_B;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lrow = lane & 31, hi = lane >> 5, sw = (lane >> 1) & 7;
    const int nqb = (p.l + 255) >> 8, pairs = p.R * p.H;
    int qb, pair;
    {
        const int id = blockIdx.x;
        if ((pairs & 7) == 0) { const int xcd = id & 7, local = id >> 3; qb = local % nqb; pair = (local / nqb) * 8 + xcd; }
        else { qb = id % nqb; pair = id / nqb; }
    }
    const int h = pair % p.H;
    const long r = pair / p.H;
    const int C3 = p.ldkv;
    const bf16_t* base = (const bf16_t*)p.qkv + r * (long)p.Lmax * C3;
    const bf16_t* kbase = base + p.k_col + h * D;
    const bf16_t* vbase = base + p.v_col + h * D;

    const int q0 = qb * 256 + w * 64;
    const bool active = q0 < p.l;
    int qi[G];
    Vis vis[G];
    bf16x8_t qf[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        qi[g] = q0 + 32 * g + lrow;
        const int qrow = min(qi[g], p.l - 1);
        vis[g] = vis_of(p, p.q_off + qrow);
        const bf16_t* qp = (const bf16_t*)p.q + (r * p.q_rows + (p.q_off + qrow - p.q_pos0)) * (long)p.ldq + h * D;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[g][ks] = *(const bf16x8_t*)(qp + (2 * ks + hi) * 8);
    }
    const int kv_end = kv_len_of(p, p.q_off + min(p.l, (qb + 1) * 256) - 1);

    const int k_key = tid >> 3, k_chunk = tid & 7;
    bf16x8_t kreg[2], vreg[2];
    const int row_bytes = C3 * 2;
    const int rec = (kv_end - 1) * row_bytes + 128;
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, rec, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, rec, 0x00020000);
    int ld_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) ld_off[i] = (k_key + 32 * i) * row_bytes + k_chunk * 16;
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    auto load_tile = [&](int kt0) {
        const int so = kt0 * row_bytes;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            kreg[i] = __builtin_bit_cast(bf16x8_t, (v4i_t)__builtin_amdgcn_raw_buffer_load_b128(k_rsrc, ld_off[i], so, 0));
            vreg[i] = __builtin_bit_cast(bf16x8_t, (v4i_t)__builtin_amdgcn_raw_buffer_load_b128(v_rsrc, ld_off[i], so, 0));
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = k_key + 32 * i;
            *(bf16x8_t*)(Ks + key * 128 + ((k_chunk ^ ((key >> 1) & 7)) << 4)) = kreg[i];
            *(bf16x8_t*)(Vs + key * 128 + ((((k_chunk >> 1) ^ (key & 3)) << 5) | ((k_chunk & 1) << 4))) = vreg[i];
        }
    };
    const int v_jrow = (lane & 15) >> 2, v_g = (lane >> 4) & 1;
    const char* v_lane[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) v_lane[db] = Vs + (4 * hi + v_jrow) * 128 + (((2 * db + v_g) ^ v_jrow) << 5) + (lane & 3) * 8;

    f32x16_t o[G][2];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[g][db][i] = 0.f;
    float m[G] = {0.f, 0.f}, lsum[G] = {0.f, 0.f};
    const short one_bf = hi == 0 ? (short)0x3f80 : (short)0;
    const bf16x8_t k_ones = {one_bf, 0, 0, 0, 0, 0, 0, 0};
    bf16x8_t q_m[G];
#pragma unroll
    for (int g = 0; g < G; ++g) q_m[g] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};

    auto tile = [&](int kt0, auto MASK, auto FIRST) {
        store_tile();
        __syncthreads();
        if (kt0 + KT < kv_end) load_tile(kt0 + KT);
        if (active) {
        f32x16_t s[G][2];
        // ---- S^T = K Q^T for both query groups off ONE K fragment
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int i = 0; i < 16; ++i) s[g][kb][i] = 0.f;
                if constexpr (!decltype(FIRST)::value) s[g][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k_ones, q_m[g], s[g][kb], 0, 0, 0);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8_t kf = *(const bf16x8_t*)(Ks + (32 * kb + lrow) * 128 + (((2 * ks + hi) ^ sw) << 4));
#pragma unroll
                for (int g = 0; g < G; ++g) s[g][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[g][ks], s[g][kb], 0, 0, 0);
            }
        }
        bf16x8_t pf[G][2][2];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if constexpr (decltype(MASK)::value) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = kt0 + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * hi;
                        if (!vis_key_t<HOLES>(vis[g], key)) s[g][kb][i] = -INFINITY;
                    }
            }
            auto row_max = [&]() {
                float tmax = -INFINITY;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, s[g][kb][i]);
                return fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            };
            auto shift = [&](bool always) {
                const float tmax = row_max();
                const bool need = always || tmax > 2.0f + fabsf(m[g]) * 0.015625f;
                const float m_new = need ? bf16_to_f32(f32_to_bf16(m[g] + tmax)) : m[g];
                const float delta = m_new - m[g];
                if (!always) {
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    lsum[g] *= alpha;
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int i = 0; i < 16; ++i) o[g][db][i] *= alpha;
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) s[g][kb][i] -= delta;
                m[g] = m_new;
                q_m[g][0] = hi == 0 ? (short)f32_to_bf16(-m_new) : (short)0;
            };
            if (decltype(FIRST)::value) shift(true);
            else if (__any(row_max() > 2.0f + fabsf(m[g]) * 0.015625f)) shift(false);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float pr[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        pr[j] = __builtin_amdgcn_exp2f(s[g][kb][8 * t + j]);
                        lsum[g] += pr[j];
                    }
                    pf[g][kb][t] = pack_bf16x8(pr);
                }
        }
        // ---- O^T += V^T P^T for both groups off ONE transposed V fragment
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const char* vp = v_lane[db] + (32 * kb + 16 * t) * 128;
                    const s16x4_t v0 = lds_tr16_b64(vp);
                    const s16x4_t v1 = lds_tr16_b64(vp + 8 * 128);
                    const bf16x8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                    for (int g = 0; g < G; ++g) o[g][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[g][kb][t], o[g][db], 0, 0, 0);
                }
        }
        __syncthreads();
    };
    typedef std::integral_constant<bool, true> Yes;
    typedef std::integral_constant<bool, false> No;
    load_tile(0);
    int kt0 = 0;
    const int wg_min_kv = range_full_prefix(p, p.q_off + min(qb * 256, p.l - 1), p.q_off + min(p.l, (qb + 1) * 256) - 1);
    if (KT <= wg_min_kv) tile(0, No{}, Yes{}); else tile(0, Yes{}, Yes{});
    kt0 = KT;
    for (; kt0 + KT <= wg_min_kv && kt0 < kv_end; kt0 += KT) tile(kt0, No{}, No{});
    for (; kt0 < kv_end; kt0 += KT) tile(kt0, Yes{}, No{});
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float ls = lsum[g] + __shfl_xor(lsum[g], 32, 64);
        if (qi[g] < p.l) {
            if (p.lse && hi == 0) p.lse[(r * p.H + h) * (long)p.l + qi[g]] = (m[g] + log2f(ls)) * 0.6931471805599453f;
            const float inv = 1.0f / ls;
            bf16_t* op = (bf16_t*)p.out + (r * p.l + qi[g]) * (long)(p.H * D) + h * D;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    float ov[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = o[g][db][4 * gg + e] * inv;
                    *(bf16x4_t*)(op + 32 * db + 8 * gg + 4 * hi) = pack_bf16x4(ov);
                }
        }
    }
}


// ================================================================================================
// attn_mfma_bf16_pipe_kernel (round 4; the inference kernel behind cvar_attention_prescaled): the QPRE kernel above, software-pipelined
// INSIDE a wave.  profiles/r03_attn_pmc_table.txt: matrix-busy 0.45 + vector-busy 0.57 ~ 1 - the three waves of a SIMD sit in the same
// phase (two workgroup barriers per tile keep them there), so the matrix pipe idles during everybody's softmax and the vector pipe during
// everybody's MFMAs.  Here a wave's instruction stream itself alternates between the two pipes: while the softmax of tile t runs on the
// vector pipe (row maximum, 64 v_exp, row sum, packing), the matrix pipe computes S^T of tile t+1 (K runs one tile ahead of V in LDS) and
// the PV products of the P fragments already packed.  The order is written out: 13 MFMA tokens (5 QK^T steps of the next tile's second
// key block, 8 PV steps) are dealt over 8 softmax chunks of 4 scores, each slot pinned with sched_barrier(0); LDS fragments are read one
// token ahead.  The second score block costs 32 VGPRs: 2 waves per SIMD instead of 3.
//   step t:   region 1   QK^T(t+1), key block 0 (bias step with the CURRENT m~ + 4 k-steps)   ||  mask(t), row maximum(t)
//             [rare]     m~ moves: scores(t), O, row sum and the block of tile t+1 just computed shift by the same exact delta
//             region 2   QK^T(t+1), key block 1 + PV(t)                                       ||  exp / row sum / pack of tile t
//             barrier - K(t+2), V(t+1) registers -> LDS - barrier - global loads of K(t+3), V(t+2)
// ================================================================================================

extern "C" int cvar_attention_rowwise(const void* qkv, const void* q, int dtype, int R, int H, int Lmax, int q_off, int l, float scale,
                                      const int* lvl_end_host, int n_lvl, const int* hole_host, void* out, float* lse, void* stream) {
    return cvar_attention_impl(qkv, q, dtype, R, H, Lmax, q_off, l, scale, lvl_end_host, n_lvl, hole_host, out, lse, stream, 1);
}
extern "C" int cvar_attention(const void* qkv, const void* q, int dtype, int R, int H, int Lmax, int q_off, int l, float scale,
                              const int* lvl_end_host, int n_lvl, const int* hole_host, void* out, float* lse, void* stream) {
    return cvar_attention_impl(qkv, q, dtype, R, H, Lmax, q_off, l, scale, lvl_end_host, n_lvl, hole_host, out, lse, stream, 0);
}
extern "C" int cvar_attention_prescaled(const void* kv, const void* q, int dtype, int R, int H, int Lmax, int q_off, int l,
                                        const int* lvl_end_host, int n_lvl, const int* hole_host, void* out, float* lse, void* stream) {
    return cvar_attention_impl(kv, q, dtype, R, H, Lmax, q_off, l, 1.0f, lvl_end_host, n_lvl, hole_host, out, lse, stream, 4);
}
extern "C" int cvar_attention_v1(const void* qkv, const void* q, int dtype, int R, int H, int Lmax, int q_off, int l, float scale,
                                 const int* lvl_end_host, int n_lvl, const int* hole_host, void* out, float* lse, void* stream) {
    return cvar_attention_impl(qkv, q, dtype, R, H, Lmax, q_off, l, scale, lvl_end_host, n_lvl, hole_host, out, lse, stream, 2);
}

// ================================================================================================
// Backward of the (level-masked) attention, exact fp32 math, row-per-lane like attn_rowwise_kernel.
//   D[q]   = sum_d dO[q,d] O[q,d]
//   P      = exp(S*scale - lse)            (recomputed, never stored)
//   dV[k] += P^T dO ;  dP = dO V^T ;  dS = P * (dP - D) ;  dQ = dS K * scale ;  dK = dS^T Q * scale
// dqkv has the arena layout [R][Lmax][3*H*64] (q | k | v thirds).
// ================================================================================================
struct AttnBwdParams {
    const void* qkv; const void* o; const void* dout; const float* lse; float* dsum; void* dqkv;
    int R, H, Lmax, q_off, l;
    float scale;
    int n_lvl;
    int lvl_end[CVAR_ATTN_MAX_LVL];
    int hole_lo[CVAR_ATTN_MAX_LVL], hole_hi[CVAR_ATTN_MAX_LVL];
};
__device__ __forceinline__ int kv_len_of_b(const AttnBwdParams& p, int pos) { return vis_of(p, pos).kvlen; }
// first query position (relative to q_off) that can see key position `key`
__device__ __forceinline__ int first_query_of(const AttnBwdParams& p, int key) {
    if (p.n_lvl == 0) return 0;
    int b = 0;
    for (int k = 0; k < p.n_lvl; ++k) { if (key < p.lvl_end[k]) break; b = p.lvl_end[k]; }
    return max(0, b - p.q_off);
}

template <typename T>
__global__ void attn_bwd_prep_kernel(const AttnBwdParams p) {      // dsum[r][h][q] = sum_d dO * O
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // over R*l*H
    const long total = (long)p.R * p.l * p.H;
    if (i >= total) return;
    const int h = (int)(i % p.H);
    const long rq = i / p.H;
    const T* op = (const T*)p.o + rq * (long)(p.H * 64) + h * 64;
    const T* dp = (const T*)p.dout + rq * (long)(p.H * 64) + h * 64;
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < 64; ++d) a = fmaf(Elem<T>::ld(op + d), Elem<T>::ld(dp + d), a);
    const long r = rq / p.l, q = rq % p.l;
    p.dsum[(r * p.H + h) * (long)p.l + q] = a;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnBwdParams p) {
    constexpr int D = 64, KT = 64;
    constexpr int VEC = 16 / sizeof(T);
    __shared__ __attribute__((aligned(16))) float Ks[KT][D];
    __shared__ __attribute__((aligned(16))) float Vs[KT][D];
    const int tid = threadIdx.x, h = blockIdx.y;
    const long r = blockIdx.z;
    const int C3 = 3 * p.H * D;
    const T* base = (const T*)p.qkv + r * (long)p.Lmax * C3;
    const int qi = blockIdx.x * 256 + tid;
    const bool valid = qi < p.l;
    const int qrow = valid ? qi : p.l - 1;
    const int pos = p.q_off + qrow;
    const Vis vis = vis_of(p, pos);
    const int kvlen = vis.kvlen;
    const int kv_end = kv_len_of_b(p, p.q_off + min(p.l, (int)(blockIdx.x + 1) * 256) - 1);
    float q[D], dO[D], dq[D];
    {
        const T* qp = base + (long)pos * C3 + h * D;
        const T* dp = (const T*)p.dout + (r * p.l + qrow) * (long)(p.H * D) + h * D;
#pragma unroll
        for (int d = 0; d < D; ++d) { q[d] = Elem<T>::ld(qp + d); dO[d] = Elem<T>::ld(dp + d); dq[d] = 0.f; }
    }
    const float lse = p.lse[(r * p.H + h) * (long)p.l + qrow];
    const float Dq = p.dsum[(r * p.H + h) * (long)p.l + qrow];
    for (int kt0 = 0; kt0 < kv_end; kt0 += KT) {
        for (int v = tid; v < KT * D / VEC; v += 256) {
            const int kk = v / (D / VEC), d0 = (v % (D / VEC)) * VEC;
            const int key = kt0 + kk;
            if (key < kv_end) {
                const T* kp = base + (long)key * C3 + p.H * D + h * D + d0;
                const T* vp = kp + p.H * D;
#pragma unroll
                for (int e = 0; e < VEC; ++e) { Ks[kk][d0 + e] = Elem<T>::ld(kp + e); Vs[kk][d0 + e] = Elem<T>::ld(vp + e); }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) { Ks[kk][d0 + e] = 0.f; Vs[kk][d0 + e] = 0.f; }
            }
        }
        __syncthreads();
        for (int kk = 0; kk < KT; ++kk) {
            if (kt0 + kk >= kvlen) break;                 // keys are visible as a prefix ...
            if (kt0 + kk >= vis.hlo && kt0 + kk < vis.hhi) continue;      // ... minus the level's hole
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < D; d += 4) {
                const f32x4_t kv = *(const f32x4_t*)&Ks[kk][d];
                const f32x4_t vv = *(const f32x4_t*)&Vs[kk][d];
                s = fmaf(q[d], kv[0], s); s = fmaf(q[d + 1], kv[1], s); s = fmaf(q[d + 2], kv[2], s); s = fmaf(q[d + 3], kv[3], s);
                dp = fmaf(dO[d], vv[0], dp); dp = fmaf(dO[d + 1], vv[1], dp); dp = fmaf(dO[d + 2], vv[2], dp); dp = fmaf(dO[d + 3], vv[3], dp);
            }
            const float pr = __expf(s * p.scale - lse);
            const float ds = pr * (dp - Dq);
#pragma unroll
            for (int d = 0; d < D; d += 4) {
                const f32x4_t kv = *(const f32x4_t*)&Ks[kk][d];
                dq[d] = fmaf(ds, kv[0], dq[d]); dq[d + 1] = fmaf(ds, kv[1], dq[d + 1]);
                dq[d + 2] = fmaf(ds, kv[2], dq[d + 2]); dq[d + 3] = fmaf(ds, kv[3], dq[d + 3]);
            }
        }
        __syncthreads();
    }
    if (valid) {
        T* op = (T*)p.dqkv + (r * p.Lmax + pos) * (long)C3 + h * D;
#pragma unroll
        for (int d = 0; d < D; ++d) Elem<T>::st(op + d, dq[d] * p.scale);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnBwdParams p) {
    constexpr int D = 64, QT = 32;
    constexpr int VEC = 16 / sizeof(T);
    __shared__ __attribute__((aligned(16))) float Qs[QT][D];
    __shared__ __attribute__((aligned(16))) float Os[QT][D];
    __shared__ float Ls[QT], Ds[QT];
    __shared__ int Kv[QT], Hlo[QT], Hhi[QT];
    const int tid = threadIdx.x, h = blockIdx.y;
    const long r = blockIdx.z;
    const int C3 = 3 * p.H * D;
    const T* base = (const T*)p.qkv + r * (long)p.Lmax * C3;
    const int nkeys = p.q_off + p.l;
    const int kj = blockIdx.x * 256 + tid;               // key position
    const bool valid = kj < nkeys;
    const int krow = valid ? kj : nkeys - 1;
    float k[D], v[D], dk[D], dv[D];
    {
        const T* kp = base + (long)krow * C3 + p.H * D + h * D;
        const T* vp = kp + p.H * D;
#pragma unroll
        for (int d = 0; d < D; ++d) { k[d] = Elem<T>::ld(kp + d); v[d] = Elem<T>::ld(vp + d); dk[d] = 0.f; dv[d] = 0.f; }
    }
    const int q_begin = first_query_of(p, blockIdx.x * 256);        // first query that sees the block's first key
    for (int qt0 = (q_begin / QT) * QT; qt0 < p.l; qt0 += QT) {
        for (int vv = tid; vv < QT * D / VEC; vv += 256) {
            const int qq = vv / (D / VEC), d0 = (vv % (D / VEC)) * VEC;
            const int qi = qt0 + qq;
            if (qi < p.l) {
                const T* qp = base + (long)(p.q_off + qi) * C3 + h * D + d0;
                const T* dp = (const T*)p.dout + (r * p.l + qi) * (long)(p.H * D) + h * D + d0;
#pragma unroll
                for (int e = 0; e < VEC; ++e) { Qs[qq][d0 + e] = Elem<T>::ld(qp + e); Os[qq][d0 + e] = Elem<T>::ld(dp + e); }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) { Qs[qq][d0 + e] = 0.f; Os[qq][d0 + e] = 0.f; }
            }
        }
        if (tid < QT) {
            const int qi = qt0 + tid;
            const bool ok = qi < p.l;
            Ls[tid] = ok ? p.lse[(r * p.H + h) * (long)p.l + qi] : 0.f;
            Ds[tid] = ok ? p.dsum[(r * p.H + h) * (long)p.l + qi] : 0.f;
            const Vis vq = vis_of(p, p.q_off + (ok ? qi : 0));
            Kv[tid] = ok ? vq.kvlen : 0; Hlo[tid] = vq.hlo; Hhi[tid] = vq.hhi;
        }
        __syncthreads();
        for (int qq = 0; qq < QT; ++qq) {
            if (krow >= Kv[qq] || (krow >= Hlo[qq] && krow < Hhi[qq])) continue;      // query does not see this key (or is padding)
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < D; d += 4) {
                const f32x4_t qv = *(const f32x4_t*)&Qs[qq][d];
                const f32x4_t ov = *(const f32x4_t*)&Os[qq][d];
                s = fmaf(qv[0], k[d], s); s = fmaf(qv[1], k[d + 1], s); s = fmaf(qv[2], k[d + 2], s); s = fmaf(qv[3], k[d + 3], s);
                dp = fmaf(ov[0], v[d], dp); dp = fmaf(ov[1], v[d + 1], dp); dp = fmaf(ov[2], v[d + 2], dp); dp = fmaf(ov[3], v[d + 3], dp);
            }
            const float pr = __expf(s * p.scale - Ls[qq]);
            const float ds = pr * (dp - Ds[qq]);
#pragma unroll
            for (int d = 0; d < D; d += 4) {
                const f32x4_t qv = *(const f32x4_t*)&Qs[qq][d];
                const f32x4_t ov = *(const f32x4_t*)&Os[qq][d];
                dv[d] = fmaf(pr, ov[0], dv[d]); dv[d + 1] = fmaf(pr, ov[1], dv[d + 1]); dv[d + 2] = fmaf(pr, ov[2], dv[d + 2]); dv[d + 3] = fmaf(pr, ov[3], dv[d + 3]);
                dk[d] = fmaf(ds, qv[0], dk[d]); dk[d + 1] = fmaf(ds, qv[1], dk[d + 1]); dk[d + 2] = fmaf(ds, qv[2], dk[d + 2]); dk[d + 3] = fmaf(ds, qv[3], dk[d + 3]);
            }
        }
        __syncthreads();
    }
    if (valid) {
        T* kp = (T*)p.dqkv + (r * p.Lmax + kj) * (long)C3 + p.H * D + h * D;
        T* vp = kp + p.H * D;
#pragma unroll
        for (int d = 0; d < D; ++d) { Elem<T>::st(kp + d, dk[d] * p.scale); Elem<T>::st(vp + d, dv[d]); }
    }
}

template <bool HOLES> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_bwd_dq_mfma_kernel(const AttnBwdParams p);
template <bool HOLES> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dkv_mfma_kernel(const AttnBwdParams p);

// ws: R*H*l floats (D = rowsum(dO * O)).  impl: 0 = auto (MFMA kernels for bf16, row-wise exact kernels for fp32), 1 = row-wise
static int cvar_attention_bwd_impl(const void* qkv, int dtype, const void* o, const void* dout, const float* lse, int R, int H, int Lmax,
                                   int q_off, int l, float scale, const int* lvl_end_host, int n_lvl, const int* hole_host, void* dqkv, float* ws,
                                   void* stream, int impl) {
    if (!qkv || !o || !dout || !lse || !dqkv || !ws || R <= 0 || H <= 0 || l <= 0 || q_off != 0 || l > Lmax) return CVAR_EINVAL;
    AttnBwdParams p;
    p.qkv = qkv; p.o = o; p.dout = dout; p.lse = lse; p.dsum = ws; p.dqkv = dqkv;
    p.R = R; p.H = H; p.Lmax = Lmax; p.q_off = q_off; p.l = l; p.scale = scale;
    { const int rc = fill_levels(p, lvl_end_host, n_lvl, hole_host); if (rc != CVAR_OK) return rc; }
    const long tot = (long)R * l * H;
    hipStream_t st = as_stream(stream);
    if (dtype == CVAR_BF16) {
        hipLaunchKernelGGL(attn_bwd_prep_kernel<bf16_t>, dim3(cdiv(tot, 256)), dim3(256), 0, st, p);
        if (impl == 0) {
            bool holes = false;
            for (int i = 0; i < p.n_lvl; ++i) holes = holes || p.hole_lo[i] < p.hole_hi[i];
            if (holes) {
                hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<true>, dim3(cdiv(l, 128), H, R), dim3(256), 0, st, p);
                hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<true>, dim3(cdiv(l, 128), H, R), dim3(256), 0, st, p);
            } else {
                hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<false>, dim3(cdiv(l, 128), H, R), dim3(256), 0, st, p);
                hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<false>, dim3(cdiv(l, 128), H, R), dim3(256), 0, st, p);
            }
        } else {
            hipLaunchKernelGGL(attn_bwd_dq_kernel<bf16_t>, dim3(cdiv(l, 256), H, R), dim3(256), 0, st, p);
            hipLaunchKernelGGL(attn_bwd_dkv_kernel<bf16_t>, dim3(cdiv(l, 256), H, R), dim3(256), 0, st, p);
        }
    } else if (dtype == CVAR_F32) {
        hipLaunchKernelGGL(attn_bwd_prep_kernel<float>, dim3(cdiv(tot, 256)), dim3(256), 0, st, p);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<float>, dim3(cdiv(l, 256), H, R), dim3(256), 0, st, p);
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<float>, dim3(cdiv(l, 256), H, R), dim3(256), 0, st, p);
    } else return CVAR_EUNSUPPORTED;
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// ================================================================================================
// bf16 MFMA backward of the level-masked attention (training).  Two kernels, no atomics:
//   attn_bwd_dq_mfma  : 4 waves x 32 queries, loops over 64-key tiles:  S^T = K Q^T, dP^T = V dO^T (same operand roles as
//                       the forward), dS = P (dP - D), dQ^T += K^T dS^T   (K^T staged transposed like V^T in the forward)
//   attn_bwd_dkv_mfma : 4 waves x 32 keys (K, V fragments in registers), loops over 64-query tiles: S = Q K^T, dP = dO V^T
//                       (lane = key), P / dS packed to bf16 in the B-operand layout, dV^T += dO^T P, dK^T += Q^T dS with
//                       Q^T / dO^T staged transposed.
// P is recomputed from the saved log-sum-exp; fp32 accumulation everywhere.
// ================================================================================================
// Round 5: the transposed operands (K^T for dQ, Q^T / dO^T for dK / dV) are no longer built on the way INTO LDS (8 pack operations + 8 ds_write_b32
// per thread, tile and image, on the vector pipe that bounds these kernels): every 64 x 64 tile is stored row-major ONCE (two ds_write_b128 per thread) under a
// swizzle that both ds_read_b128 fragment reads along a row and ds_read_b64_tr_b16 transpose reads take without bank conflicts (bw_swz below).  Tiles whose keys
// every query of the tile sees (all but the level-boundary ones) run a compile-time unmasked body: no per-score compare / select, no visibility tables.

// staging map: thread -> rows (tid >> 3) and (tid >> 3) + 32 of a 64 x 64 bf16 tile, 16-byte chunk tid & 7 (the forward kernel's map)
struct BwRows { bf16x8_t a, b; };
__device__ __forceinline__ BwRows bw_load_rows(const bf16_t* src, long row_stride, int row0, int row_limit, int tid) {
    const int row = tid >> 3, chunk = tid & 7;
    const bf16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    BwRows v;
    v.a = (row0 + row) < row_limit ? *(const bf16x8_t*)(src + (long)(row0 + row) * row_stride + chunk * 8) : z;
    v.b = (row0 + row + 32) < row_limit ? *(const bf16x8_t*)(src + (long)(row0 + row + 32) * row_stride + chunk * 8) : z;
    return v;
}
// ONE image serves both read patterns (round 5, second step: the stores of a second, transpose-read image cost 21 % of the dK/dV kernel - tools ablation,
// profiles/r05_attn_bwd.txt).  Row r keeps its eight 16-byte chunks at chunk ^ bw_swz(r), bw_swz(r) = ((r >> 1) & 1) << 2 | ((r >> 2) & 3):
//   * ds_read_b128 fragment reads along a row (lane = row, one chunk): the hardware's 16-lane groups hold rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31};
//     rows of one parity inside a group get eight DIFFERENT swizzles ({0,4,3,7,1,5,2,6} / {1,5,2,6,0,4,3,7}) - with the 128-byte row pitch all 64 banks once;
//   * ds_read_b64_tr_b16 transpose reads (a half-wave covers rows r0 .. r0 + 3 x four adjacent chunks): rows r0 and r0 + 2 share a bank half, and their swizzles
//     differ in bit 2, i.e. they land in different 64-byte halves of the row - conflict-free as well.
// Rows r and r + 32 share the swizzle (the two rows a thread stages).
__device__ __forceinline__ int bw_swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ void bw_store(char* dst, const BwRows& v, int tid) {
    const int row = tid >> 3, chunk = tid & 7;
    char* d = dst + row * 128 + ((chunk ^ bw_swz(row)) << 4);
    *(bf16x8_t*)d = v.a;
    *(bf16x8_t*)(d + 32 * 128) = v.b;
}
// lane address of the row fragment (MFMA operand: lane & 31 = row inside a 32-row block, chunk 2 ks + hi): add (32 * block) * 128
__device__ __forceinline__ int bw_row_off(int lrow, int chunk) { return lrow * 128 + ((chunk ^ bw_swz(lrow)) << 4); }
// lane addresses of the transposed fragments of column block db (columns 32 db .. 32 db + 31 become the fragment's rows): lane l of a 16-lane group points at tile
// row (l & 15) >> 2 of a [4 rows][16 columns] block, 8 bytes = columns 4 (l & 3) .. + 3; the upper half-wave 4 rows further.  Two bases: rows r0 + 4 hi + j and the
// same + 8 (their swizzles differ: (row >> 2) & 3 = hi resp. hi + 2 for r0 a multiple of 16).
struct BwTrLane { const char* lo; const char* up; };
__device__ __forceinline__ BwTrLane bw_tr_lane(const char* img, int lane, int db) {
    const int hi = lane >> 5, jrow = (lane & 15) >> 2, g = (lane >> 4) & 1, x = lane & 3;
    const int c = 4 * db + 2 * g + (x >> 1);
    const int f_lo = ((jrow >> 1) << 2) | hi, f_up = ((jrow >> 1) << 2) | (hi + 2);
    BwTrLane r;
    r.lo = img + (4 * hi + jrow) * 128 + ((c ^ f_lo) << 4) + (x & 1) * 8;
    r.up = img + (8 + 4 * hi + jrow) * 128 + ((c ^ f_up) << 4) + (x & 1) * 8;
    return r;
}
// A-operand fragment T^T[32 db + (lane & 31)][row0 + {4 hi .. 4 hi + 3, 8 + 4 hi .. 8 + 4 hi + 3}], row0 a multiple of 16
__device__ __forceinline__ bf16x8_t bw_tr_frag(const BwTrLane& b, int row0) {
    const s16x4_t v0 = lds_tr16_b64(b.lo + row0 * 128);
    const s16x4_t v1 = lds_tr16_b64(b.up + row0 * 128);
    const bf16x8_t f = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    return f;
}

template <bool HOLES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_bwd_dq_mfma_kernel(const AttnBwdParams p) {
    constexpr int D = 64, KT = 64;
    __shared__ __attribute__((aligned(1024))) char Ks[KT * 128];      // K: row fragments (S^T = K Q^T) and transposed fragments (dQ^T += K^T dS^T) out of one image
    __shared__ __attribute__((aligned(1024))) char Vs[KT * 128];      // V: row fragments (dP^T = V dO^T)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lrow = lane & 31, hi = lane >> 5;
    const int h = blockIdx.y;
    const long r = blockIdx.z;
    const int C3 = 3 * p.H * D;
    const bf16_t* base = (const bf16_t*)p.qkv + r * (long)p.Lmax * C3;
    const bf16_t* kbase = base + p.H * D + h * D;
    const bf16_t* vbase = kbase + p.H * D;
    const int q0 = blockIdx.x * 128 + w * 32;
    const int qi = q0 + lrow;
    const int qrow = min(qi, p.l - 1);
    const Vis vis = vis_of(p, p.q_off + qrow);
    const int wave_min_kv = range_full_prefix(p, p.q_off + min(q0, p.l - 1), p.q_off + min(q0 + 31, p.l - 1));
    const int kv_end = kv_len_of_b(p, p.q_off + min(p.l, (int)(blockIdx.x + 1) * 128) - 1);
    bf16x8_t qf[4], of[4];
    {
        const bf16_t* qp = base + (long)(p.q_off + qrow) * C3 + h * D;
        const bf16_t* dp = (const bf16_t*)p.dout + (r * p.l + qrow) * (long)(p.H * D) + h * D;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qf[ks] = *(const bf16x8_t*)(qp + (2 * ks + hi) * 8); of[ks] = *(const bf16x8_t*)(dp + (2 * ks + hi) * 8); }
    }
    const float c2 = p.scale * 1.4426950408889634f;
    const float lse2 = p.lse[(r * p.H + h) * (long)p.l + qrow] * 1.4426950408889634f;
    const float Dq = p.dsum[(r * p.H + h) * (long)p.l + qrow];
    const BwTrLane k2_lane[2] = {bw_tr_lane(Ks, lane, 0), bw_tr_lane(Ks, lane, 1)};
    int row_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) row_off[ks] = bw_row_off(lrow, 2 * ks + hi);
    f32x16_t dq[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[db][i] = 0.f;

    // one key tile's arithmetic, one 32-key half at a time (scores, dP, dS and the dQ update of a half are finished before the next half starts: 32
    // accumulator registers live instead of 64 - three waves per SIMD); MASK compile-time (tiles below wave_min_kv are seen completely by all 32 queries of the wave)
    auto compute = [&](int kt0, auto MASK) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16_t s, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = 32 * kb * 128 + row_off[ks];
                const bf16x8_t kf = *(const bf16x8_t*)(Ks + off);
                const bf16x8_t vf = *(const bf16x8_t*)(Vs + off);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, of[ks], dp, 0, 0, 0);
            }
            bf16x8_t dsf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float ds[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = 8 * t + j;
                    float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], c2, -lse2));
                    if constexpr (decltype(MASK)::value) {
                        const int key = kt0 + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * hi;
                        if (!vis_key_t<HOLES>(vis, key)) pr = 0.f;
                    }
                    ds[j] = pr * (dp[i] - Dq);
                }
                dsf[t] = pack_bf16x8(ds);
            }
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw_tr_frag(k2_lane[db], 32 * kb + 16 * t), dsf[t], dq[db], 0, 0, 0);
        }
    };
    typedef std::integral_constant<bool, true> Yes;
    typedef std::integral_constant<bool, false> No;
    BwRows kv_k = bw_load_rows(kbase, C3, 0, kv_end, tid), kv_v = bw_load_rows(vbase, C3, 0, kv_end, tid);
    for (int kt0 = 0; kt0 < kv_end; kt0 += KT) {
        bw_store(Ks, kv_k, tid);
        bw_store(Vs, kv_v, tid);
        __syncthreads();
        if (kt0 + KT < kv_end) {            // next K / V tile travels while this one is computed
            kv_k = bw_load_rows(kbase, C3, kt0 + KT, kv_end, tid);
            kv_v = bw_load_rows(vbase, C3, kt0 + KT, kv_end, tid);
        }
        if (kt0 + KT > wave_min_kv) compute(kt0, Yes{}); else compute(kt0, No{});      // wave-uniform; no barrier inside
        __syncthreads();
    }
    if (qi < p.l) {
        bf16_t* op = (bf16_t*)p.dqkv + (r * p.Lmax + p.q_off + qi) * (long)C3 + h * D;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float ov[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = dq[db][4 * g + e] * p.scale;
                *(bf16x4_t*)(op + 32 * db + 8 * g + 4 * hi) = pack_bf16x4(ov);
            }
    }
}

template <bool HOLES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dkv_mfma_kernel(const AttnBwdParams p) {
    constexpr int D = 64, QT = 64;
    __shared__ __attribute__((aligned(1024))) char Qs[QT * 128];      // Q / dO tiles: row fragments (S = Q K^T, dP = dO V^T) and transposed fragments
    __shared__ __attribute__((aligned(1024))) char Os[QT * 128];      // (dK^T += Q^T dS, dV^T += dO^T P) out of one image each
    __shared__ __attribute__((aligned(16))) float Ls[QT];
    __shared__ __attribute__((aligned(16))) float Dsum[QT];
    __shared__ __attribute__((aligned(16))) int Kv[QT];
    __shared__ __attribute__((aligned(16))) int Hlo[QT];
    __shared__ __attribute__((aligned(16))) int Hhi[QT];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lrow = lane & 31, hi = lane >> 5;
    const int h = blockIdx.y;
    const long r = blockIdx.z;
    const int C3 = 3 * p.H * D;
    const bf16_t* base = (const bf16_t*)p.qkv + r * (long)p.Lmax * C3;
    const bf16_t* qbase = base + (long)p.q_off * C3 + h * D;
    const bf16_t* obase = (const bf16_t*)p.dout + r * (long)p.l * (p.H * D) + h * D;
    const int nkeys = p.q_off + p.l;
    const int kj = blockIdx.x * 128 + w * 32 + lrow;                 // key position owned by this lane
    const int krow = min(kj, nkeys - 1);
    const int key_hi = min((int)(blockIdx.x + 1) * 128, nkeys);      // one past the workgroup's last key
    bf16x8_t kf[4], vf[4];
    {
        const bf16_t* kp = base + (long)krow * C3 + p.H * D + h * D;
        const bf16_t* vp = kp + p.H * D;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kf[ks] = *(const bf16x8_t*)(kp + (2 * ks + hi) * 8); vf[ks] = *(const bf16x8_t*)(vp + (2 * ks + hi) * 8); }
    }
    const float c2 = p.scale * 1.4426950408889634f;
    const BwTrLane q2_lane[2] = {bw_tr_lane(Qs, lane, 0), bw_tr_lane(Qs, lane, 1)};
    const BwTrLane o2_lane[2] = {bw_tr_lane(Os, lane, 0), bw_tr_lane(Os, lane, 1)};
    int row_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) row_off[ks] = bw_row_off(lrow, 2 * ks + hi);
    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[db][i] = 0.f; dv[db][i] = 0.f; }

    // one query tile's arithmetic.  MASK compile-time: a tile whose 64 queries all see every key of this workgroup needs no visibility test
    auto compute = [&](auto MASK) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16_t s, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = 32 * qb * 128 + row_off[ks];
                const bf16x8_t qa = *(const bf16x8_t*)(Qs + off);
                const bf16x8_t oa = *(const bf16x8_t*)(Os + off);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s, 0, 0, 0);       // S[q][key]: lane = key, regs = queries
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, vf[ks], dp, 0, 0, 0);
            }
            bf16x8_t pf[2], df[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float pv[8], dsv[8];
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int g = 2 * t + g2;                                   // regs 4g..4g+3 <-> queries 32qb + 8g + 4hi + 0..3
                    const int qloc = 32 * qb + 8 * g + 4 * hi;
                    const f32x4_t l4 = *(const f32x4_t*)&Ls[qloc];
                    const f32x4_t d4 = *(const f32x4_t*)&Dsum[qloc];
                    int kvl[4] = {0, 0, 0, 0}, hlo4[4] = {0, 0, 0, 0}, hhi4[4] = {0, 0, 0, 0};
                    if constexpr (decltype(MASK)::value) {
                        const int4 k4 = *(const int4*)&Kv[qloc];
                        kvl[0] = k4.x; kvl[1] = k4.y; kvl[2] = k4.z; kvl[3] = k4.w;
                        if constexpr (HOLES) {
                            const int4 l4i = *(const int4*)&Hlo[qloc];
                            const int4 h4i = *(const int4*)&Hhi[qloc];
                            hlo4[0] = l4i.x; hlo4[1] = l4i.y; hlo4[2] = l4i.z; hlo4[3] = l4i.w;
                            hhi4[0] = h4i.x; hhi4[1] = h4i.y; hhi4[2] = h4i.z; hhi4[3] = h4i.w;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 4 * g + e;
                        float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], c2, -l4[e]));
                        if constexpr (decltype(MASK)::value) {
                            if (krow >= kvl[e] || (HOLES && krow >= hlo4[e] && krow < hhi4[e])) pr = 0.f;
                        }
                        pv[4 * g2 + e] = pr;
                        dsv[4 * g2 + e] = pr * (dp[i] - d4[e]);
                    }
                }
                pf[t] = pack_bf16x8(pv);
                df[t] = pack_bf16x8(dsv);
            }
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int row0 = 32 * qb + 16 * t;
                    dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw_tr_frag(o2_lane[db], row0), pf[t], dv[db], 0, 0, 0);
                    dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw_tr_frag(q2_lane[db], row0), df[t], dk[db], 0, 0, 0);
                }
        }
    };
    typedef std::integral_constant<bool, true> Yes;
    typedef std::integral_constant<bool, false> No;

    const int q_begin = first_query_of(p, blockIdx.x * 128);            // first query that sees the block's first key
    const int qt_first = (q_begin / QT) * QT;
    BwRows qv = bw_load_rows(qbase, C3, qt_first, p.l, tid), ov = bw_load_rows(obase, (long)(p.H * D), qt_first, p.l, tid);
    float ls_v = 0.f, ds_v = 0.f;
    auto load_rowstats = [&](int qt0) {
        if (tid < QT) {
            const int qi = qt0 + tid;
            const bool ok = qi < p.l;
            ls_v = ok ? p.lse[(r * p.H + h) * (long)p.l + qi] * 1.4426950408889634f : 0.f;
            ds_v = ok ? p.dsum[(r * p.H + h) * (long)p.l + qi] : 0.f;
        }
    };
    load_rowstats(qt_first);
    // workgroup-uniform: all 64 queries of the tile exist and each sees every key of this workgroup
    auto is_full = [&](int qt0) { return qt0 + QT <= p.l && range_full_prefix(p, p.q_off + qt0, p.q_off + qt0 + QT - 1) >= key_hi; };
    auto tile = [&](int qt0, auto MASK) {
        bw_store(Qs, qv, tid);
        bw_store(Os, ov, tid);
        if (tid < QT) {
            const int qi = qt0 + tid;
            const bool ok = qi < p.l;
            Ls[tid] = ls_v;
            Dsum[tid] = ds_v;
            if constexpr (decltype(MASK)::value) {
                const Vis vq = vis_of(p, p.q_off + (ok ? qi : 0));
                Kv[tid] = ok ? vq.kvlen : 0;
                if constexpr (HOLES) { Hlo[tid] = vq.hlo; Hhi[tid] = vq.hhi; }
            }
        }
        __syncthreads();
        if (qt0 + QT < p.l) {               // next tile's operands travel while this tile is computed
            qv = bw_load_rows(qbase, C3, qt0 + QT, p.l, tid);
            ov = bw_load_rows(obase, (long)(p.H * D), qt0 + QT, p.l, tid);
            load_rowstats(qt0 + QT);
        }
        compute(MASK);
        __syncthreads();
    };
    // three loops, one instantiation of the tile body each (both bodies inside ONE loop made the compiler spill 46-95 registers): the masked level-boundary
    // tiles, the run of completely visible tiles behind them (the visible prefix only grows with the query position when no level has a hole), the tail
    int qt0 = qt_first;
    if constexpr (!HOLES) {
        for (; qt0 < p.l && !is_full(qt0); qt0 += QT) tile(qt0, Yes{});
        for (; qt0 < p.l && is_full(qt0); qt0 += QT) tile(qt0, No{});
    }
    for (; qt0 < p.l; qt0 += QT) tile(qt0, Yes{});
    if (kj < nkeys) {
        bf16_t* kp = (bf16_t*)p.dqkv + (r * p.Lmax + kj) * (long)C3 + p.H * D + h * D;
        bf16_t* vp = kp + p.H * D;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float a[4], b[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] = dk[db][4 * g + e] * p.scale; b[e] = dv[db][4 * g + e]; }
                *(bf16x4_t*)(kp + 32 * db + 8 * g + 4 * hi) = pack_bf16x4(a);
                *(bf16x4_t*)(vp + 32 * db + 8 * g + 4 * hi) = pack_bf16x4(b);
            }
    }
}

extern "C" int cvar_attention_bwd_rowwise(const void* qkv, int dtype, const void* o, const void* dout, const float* lse, int R, int H, int Lmax,
                                          int q_off, int l, float scale, const int* lvl_end_host, int n_lvl, const int* hole_host, void* dqkv,
                                          float* ws, void* stream) {
    return cvar_attention_bwd_impl(qkv, dtype, o, dout, lse, R, H, Lmax, q_off, l, scale, lvl_end_host, n_lvl, hole_host, dqkv, ws, stream, 1);
}
extern "C" int cvar_attention_bwd(const void* qkv, int dtype, const void* o, const void* dout, const float* lse, int R, int H, int Lmax,
                                  int q_off, int l, float scale, const int* lvl_end_host, int n_lvl, const int* hole_host, void* dqkv, float* ws,
                                  void* stream) {
    return cvar_attention_bwd_impl(qkv, dtype, o, dout, lse, R, H, Lmax, q_off, l, scale, lvl_end_host, n_lvl, hole_host, dqkv, ws, stream, 0);
}
