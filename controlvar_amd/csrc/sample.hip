// CFG combine + greedy / top-k / top-p sampling over the 4096-way codebook (one workgroup per token).
//   combine: ((c0*l0 + c1*l1) + c2*l2) + c3*l3 with separately rounded products - the evaluation order of
//            control_var.py:295-298 / 501-502, so the combined logits are bit-identical to the reference's.
//   greedy : argmax, lowest index on ties, plus the top1-top2 margin (used for margin-aware parity checks).
//   sample : bitonic sort (value desc, index asc) in LDS, top-k threshold (ties kept, helpers.py:9-10), nucleus cut
//            on the ascending cumulative mass (helpers.py:12-15), inverse-CDF draw from a counter-based generator.
#include "cvar_common.h"

struct SampleParams {
    const float* logits;
    int B, nrep, l, V;
    float coef[4];
    int top_k;
    float top_p;
    unsigned long long seed;
    int stage, n_draw;
    int* idx_out;
    float* combined;
    float* margin;
    int* kept;
};

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float uniform01(unsigned long long seed, int stage, long row, int t) {
    unsigned long long h = splitmix64(seed ^ 0xC0FFEE1234ull);
    h = splitmix64(h ^ ((unsigned long long)stage << 48) ^ ((unsigned long long)row << 16) ^ (unsigned long long)t);
    return (float)(h >> 40) * (1.0f / 16777216.0f);           // 24 random bits -> [0,1)
}

__device__ __forceinline__ float combine_logits(const SampleParams& p, long b, long t, int e) {
    const long rowstride = (long)p.l * p.V;
    const float* base = p.logits + (b * p.l + t) * p.V + e;
    float v = __fmul_rn(p.coef[0], base[0]);
    for (int r = 1; r < p.nrep; ++r) v = __fadd_rn(v, __fmul_rn(p.coef[r], base[(long)r * p.B * rowstride]));
    return v;
}

struct Top2 { float b1; int i1; float b2; };
__device__ __forceinline__ Top2 merge_top2(Top2 a, Top2 b) {
    Top2 o;
    const bool a_first = (a.b1 > b.b1) || (a.b1 == b.b1 && a.i1 < b.i1);
    if (a_first) { o.b1 = a.b1; o.i1 = a.i1; o.b2 = fmaxf(a.b2, b.b1); }
    else { o.b1 = b.b1; o.i1 = b.i1; o.b2 = fmaxf(b.b2, a.b1); }
    return o;
}

__global__ __launch_bounds__(256) void cfg_greedy_kernel(const SampleParams p) {
    __shared__ float sb1[4], sb2[4];
    __shared__ int si1[4];
    const int tid = threadIdx.x;
    const long bt = blockIdx.x;
    const long b = bt / p.l, t = bt % p.l;
    Top2 best = {-INFINITY, 0x7fffffff, -INFINITY};
    for (int e = tid; e < p.V; e += 256) {
        const float v = combine_logits(p, b, t, e);
        if (p.combined) p.combined[bt * p.V + e] = v;
        Top2 cur = {v, e, -INFINITY};
        best = merge_top2(best, cur);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Top2 other;
        other.b1 = __shfl_xor(best.b1, o, 64); other.i1 = __shfl_xor(best.i1, o, 64); other.b2 = __shfl_xor(best.b2, o, 64);
        best = merge_top2(best, other);
    }
    if ((tid & 63) == 0) { sb1[tid >> 6] = best.b1; si1[tid >> 6] = best.i1; sb2[tid >> 6] = best.b2; }
    __syncthreads();
    if (tid == 0) {
        Top2 r = {sb1[0], si1[0], sb2[0]};
        for (int w = 1; w < 4; ++w) { Top2 o = {sb1[w], si1[w], sb2[w]}; r = merge_top2(r, o); }
        for (int d = 0; d < p.n_draw; ++d) p.idx_out[((long)d * p.B + b) * p.l + t] = r.i1;
        if (p.margin) p.margin[bt] = r.b1 - r.b2;
        if (p.kept) p.kept[bt] = 1;
    }
}

__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

__global__ __launch_bounds__(256) void cfg_sample_kernel(const SampleParams p) {
    constexpr int N = 4096, EPT = 16;
    __shared__ float sv[N];
    __shared__ int si[N];
    __shared__ float chunk_tot[256];
    __shared__ float scratch[4];
    const int tid = threadIdx.x;
    const long bt = blockIdx.x;
    const long b = bt / p.l, t = bt % p.l;
    for (int e = tid; e < N; e += 256) {
        float v = -INFINITY;
        if (e < p.V) {
            v = combine_logits(p, b, t, e);
            if (p.combined) p.combined[bt * p.V + e] = v;
        }
        sv[e] = v; si[e] = e;
    }
    __syncthreads();
    // bitonic sort: value descending, index ascending on ties
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < N; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const float a = sv[i], c = sv[ixj];
                    const int ia = si[i], ic = si[ixj];
                    const bool a_before = (a > c) || (a == c && ia < ic);
                    const bool want = ((i & k) == 0);
                    if (a_before != want) { sv[i] = c; sv[ixj] = a; si[i] = ic; si[ixj] = ia; }
                }
            }
            __syncthreads();
        }
    }
    if (p.margin && tid == 0) p.margin[bt] = sv[0] - sv[1];
    // top-k: keep everything >= the k-th largest value (ties kept)
    int nk = p.V;
    if (p.top_k > 0 && p.top_k < p.V) {
        const float thr = sv[p.top_k - 1];
        float cnt = 0.f;
        for (int i = tid; i < p.V; i += 256) cnt += (sv[i] >= thr) ? 1.f : 0.f;
        nk = (int)(block_sum(cnt, scratch) + 0.5f);
    }
    // exp weights + inclusive prefix sums over the sorted order (thread owns EPT consecutive entries)
    const float vmax = sv[0];
    float ex[EPT], run = 0.f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid * EPT + e;
        const float w = (i < nk) ? __expf(sv[i] - vmax) : 0.f;
        run += w; ex[e] = run;
    }
    __syncthreads();
    chunk_tot[tid] = run;
    __syncthreads();
    if (tid < 64) {          // scan 256 chunk totals with one wave (4 per lane)
        float c0 = chunk_tot[tid * 4], c1 = chunk_tot[tid * 4 + 1], c2 = chunk_tot[tid * 4 + 2], c3 = chunk_tot[tid * 4 + 3];
        const float tot = ((c0 + c1) + c2) + c3;
        float inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const float n = __shfl_up(inc, o, 64); if (tid >= o) inc += n; }
        const float excl = inc - tot;
        chunk_tot[tid * 4] = excl; chunk_tot[tid * 4 + 1] = excl + c0; chunk_tot[tid * 4 + 2] = excl + c0 + c1; chunk_tot[tid * 4 + 3] = excl + c0 + c1 + c2;
    }
    __syncthreads();
    const float off = chunk_tot[tid];
    __syncthreads();
    // reuse sv as the inclusive prefix mass P_i (values no longer needed except through P)
    float prevP = off;
#pragma unroll
    for (int e = 0; e < EPT; ++e) { ex[e] += off; }
    const float Zall = block_sum((tid == 255) ? ex[EPT - 1] : 0.f, scratch);       // total mass of the top-k set
    int nkeep = nk;
    if (p.top_p > 0.f) {
        const float lim = 1.0f - p.top_p;
        float cnt = 0.f;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid * EPT + e;
            const float tail = Zall - prevP;              // mass of entries i.. in descending order
            if (i < nk && (i == 0 || tail / Zall > lim)) cnt += 1.f;
            prevP = ex[e];
        }
        nkeep = (int)(block_sum(cnt, scratch) + 0.5f);
        if (nkeep < 1) nkeep = 1;
    }
    if (p.kept && tid == 0) p.kept[bt] = nkeep;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) sv[tid * EPT + e] = ex[e];
    __syncthreads();
    const float Zkeep = sv[nkeep - 1];
    for (int d = 0; d < p.n_draw; ++d) {
        const float u = uniform01(p.seed, p.stage, (long)d * p.B + b, (int)t);
        const float target = u * Zkeep;
        float cnt = 0.f;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid * EPT + e;
            if (i < nkeep && ex[e] <= target) cnt += 1.f;
        }
        int pick = (int)(block_sum(cnt, scratch) + 0.5f);
        if (pick > nkeep - 1) pick = nkeep - 1;
        if (tid == 0) p.idx_out[((long)d * p.B + b) * p.l + t] = si[pick];
    }
}

extern "C" int cvar_cfg_sample(const float* logits, int B, int nrep, int l, int V, const float* coef_host,
                               int top_k, float top_p, uint64_t seed, int stage, int n_draw,
                               int32_t* idx_out, float* combined, float* margin, int32_t* kept, void* stream) {
    if (!logits || !coef_host || !idx_out || B <= 0 || l <= 0 || V <= 1) return CVAR_EINVAL;
    if (nrep < 1 || nrep > 4 || n_draw < 1 || n_draw > 4 || V > 4096) return CVAR_EUNSUPPORTED;
    SampleParams p;
    p.logits = logits; p.B = B; p.nrep = nrep; p.l = l; p.V = V;
    for (int i = 0; i < 4; ++i) p.coef[i] = i < nrep ? coef_host[i] : 0.f;
    p.top_k = top_k; p.top_p = top_p; p.seed = seed; p.stage = stage; p.n_draw = n_draw;
    p.idx_out = idx_out; p.combined = combined; p.margin = margin; p.kept = kept;
    dim3 grid((unsigned)((long)B * l)), block(256);
    if (top_k == 1) hipLaunchKernelGGL(cfg_greedy_kernel, grid, block, 0, as_stream(stream), p);
    else hipLaunchKernelGGL(cfg_sample_kernel, grid, block, 0, as_stream(stream), p);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}
