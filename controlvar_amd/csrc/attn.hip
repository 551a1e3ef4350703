// Multi-scale KV-cached attention over the qkv arena [R][Lmax][3*H*64].
//
// attn_rowwise_kernel: exact-order fp32 row-per-lane kernel (VALU).  It is the parity-mode (CVAR_F32)
// implementation and the in-library reference the MFMA flash kernel is A/B-checked against.  One lane owns one
// query row (q, o in registers), K/V tiles of 64 keys are staged in LDS as fp32 and read as wave-wide
// broadcasts; block-wise online softmax (one rescale per 64-key tile).
#include "cvar_common.h"

struct AttnParams {
    const void* qkv;
    void* out;
    int R, H, Lmax, q_off, l;
    float scale;
    int n_lvl;
    int lvl_end[16];
};

__device__ __forceinline__ int kv_len_of(const AttnParams& p, int pos) {
    if (p.n_lvl == 0) return p.q_off + p.l;
    int e = p.lvl_end[p.n_lvl - 1];
    for (int k = p.n_lvl - 1; k >= 0; --k)
        if (pos < p.lvl_end[k]) e = p.lvl_end[k];
    return e;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_rowwise_kernel(const AttnParams p) {
    constexpr int D = 64, KT = 64;
    constexpr int VEC = 16 / sizeof(T);
    __shared__ __attribute__((aligned(16))) float Ks[KT][D];
    __shared__ __attribute__((aligned(16))) float Vs[KT][D];
    const int tid = threadIdx.x;
    const int h = blockIdx.y;
    const long r = blockIdx.z;
    const int C3 = 3 * p.H * D;
    const T* base = (const T*)p.qkv + r * (long)p.Lmax * C3;
    const int qi = blockIdx.x * 256 + tid;
    const bool valid = qi < p.l;
    const int pos = p.q_off + (valid ? qi : p.l - 1);
    const int kvlen = kv_len_of(p, pos);
    const int last_q = min(p.l, (int)(blockIdx.x + 1) * 256) - 1;
    const int kv_end = kv_len_of(p, p.q_off + last_q);          // monotone in pos -> block maximum

    float q[D], o[D];
    {
        const T* qp = base + (long)pos * C3 + h * D;
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
            s[kk] = (kt0 + kk < kvlen) ? a : -INFINITY;
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
    }
}

extern "C" int cvar_attention(const void* qkv, int dtype, int R, int H, int Lmax, int q_off, int l, float scale,
                              const int* lvl_end_host, int n_lvl, void* out, void* stream) {
    if (!qkv || !out || R <= 0 || H <= 0 || l <= 0 || q_off < 0 || q_off + l > Lmax) return CVAR_EINVAL;
    if (n_lvl < 0 || n_lvl > 16 || (n_lvl > 0 && !lvl_end_host)) return CVAR_EINVAL;
    AttnParams p;
    p.qkv = qkv; p.out = out; p.R = R; p.H = H; p.Lmax = Lmax; p.q_off = q_off; p.l = l; p.scale = scale; p.n_lvl = n_lvl;
    for (int i = 0; i < 16; ++i) p.lvl_end[i] = i < n_lvl ? lvl_end_host[i] : 0;
    dim3 grid(cdiv(l, 256), H, R), block(256);
    if (dtype == CVAR_BF16) hipLaunchKernelGGL(attn_rowwise_kernel<bf16_t>, grid, block, 0, as_stream(stream), p);
    else if (dtype == CVAR_F32) hipLaunchKernelGGL(attn_rowwise_kernel<float>, grid, block, 0, as_stream(stream), p);
    else return CVAR_EUNSUPPORTED;
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}
