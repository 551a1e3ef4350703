// CFG combine + greedy / top-k / top-p sampling over the 4096-way codebook (one workgroup per token).
//   combine: ((c0*l0 + c1*l1) + c2*l2) + c3*l3 with separately rounded products - the evaluation order of
//            control_var.py:295-298 / 501-502, so the combined logits are bit-identical to the reference's.
//   greedy : argmax, lowest index on ties, plus the top1-top2 margin (used for margin-aware parity checks).
//   sample : radix-select thresholds for top-k (ties kept, helpers.py:9-10) and the nucleus cut on the ascending
//            cumulative mass (helpers.py:12-15), inverse-CDF draw from a counter-based generator (no sort).
#include "cvar_common.h"

struct SampleParams {
    const float* logits;
    int B, nrep, l, V;
    float coef[4];
    int top_k;
    float top_p;
    unsigned long long seed;
    const unsigned long long* seed_dev;     // optional device-resident seed added to `seed` (lets a captured hipGraph draw fresh samples)
    int stage, n_draw;
    int* idx_out;
    float* combined;
    float* margin;
    int* kept;
    int ldv;                                // row stride of `logits` in floats (>= V: a wider head whose first V columns are the codes)
    // more_smooth (control_var.py:511-515; helpers.py:22-36): soft code embedding of the top-k / top-p MASKED logits
    //   soft_out[(d*B + b)][t][:] = softmax_v( (lg_v * smooth_mul + g_v) / smooth_tau ) . codebook[v][:]      over the kept v
    const float* codebook;                  // [V][Cvae] fp32; NULL: no soft output
    int Cvae;
    float smooth_mul, smooth_tau;
    const float* gumbel;                    // optional injected Gumbel noise [n_draw*B][l][V] (tests); NULL: drawn from the counter generator
    float* soft_out;
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
    const long rowstride = (long)p.l * p.ldv;
    const float* base = p.logits + (b * p.l + t) * p.ldv + e;
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

// ---- top-k / top-p sampling without a sort ---------------------------------------------------------------------
// Both filters of helpers.py:8-15 are VALUE thresholds: top-k keeps v >= (k-th largest value) (ties kept), the nucleus
// cut keeps the elements whose ascending cumulative probability exceeds 1-p, i.e. v >= tau_p.  Each threshold is found
// by a 4-pass byte-wise radix select over order-preserving integer keys (count histogram for top-k, probability-mass
// histogram for top-p).  Masses are quantised to 2^-40 and accumulated as 64-bit integers: exact, order independent,
// bit-reproducible.  The draw is an inverse-CDF lookup over the kept elements in a fixed (thread-major) order.
__device__ __forceinline__ unsigned f2key(float f) {           // monotone: a < b  <=>  key(a) < key(b)
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <typename TV>
__device__ __forceinline__ TV wave_incl_scan(TV v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const TV n = __shfl_up(v, o, 64); if (lane >= o) v += n; }
    return v;
}
// exclusive prefix of one value per thread over the 256-thread block (+ block total through `total`)
template <typename TV>
__device__ __forceinline__ TV block_excl_scan(TV v, TV* wsum /*[4]*/, TV& total) {
    const TV inc = wave_incl_scan(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 63) wsum[w] = inc;
    __syncthreads();
    TV base = 0;
    for (int i = 0; i < w; ++i) base += wsum[i];
    total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return base + inc - v;
}

// SOFT = false (every launch without more_smooth): the soft-embedding block and its registers compile away (146 -> ~90 VGPRs: the kernel is
// bound by memory and LDS latency, and went 44 % slower when the block cost it two of its five waves per SIMD)
template <bool SOFT>
__global__ __launch_bounds__(256) void cfg_sample_kernel(const SampleParams p) {
    constexpr int EPT = 16;
    typedef unsigned long long u64;
    __shared__ int hist_cnt[256];
    __shared__ u64 hist_mass[256];
    __shared__ u64 wsum64[4];
    __shared__ int wsum32[4];
    __shared__ float wmaxs[4], wmax2[4];
    __shared__ unsigned sel_digit;
    __shared__ u64 sel_below;
    __shared__ int sel_k;
    const int tid = threadIdx.x;
    const long bt = blockIdx.x;
    const long b = bt / p.l, t = bt % p.l;
    float v[EPT];
    unsigned key[EPT];
    float best = -INFINITY, second = -INFINITY;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = i * 256 + tid;                   // thread-strided ownership: coalesced loads
        float x = -INFINITY;
        if (e < p.V) {
            x = combine_logits(p, b, t, e);
            if (p.combined) p.combined[bt * p.V + e] = x;
        }
        v[i] = x; key[i] = f2key(x);
        if (x > best) { second = best; best = x; } else if (x > second) second = x;
    }
    // block max (+ second max for the margin output)
    {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64), os = __shfl_xor(second, o, 64);
            second = fmaxf(fminf(best, ob), fmaxf(second, os));
            best = fmaxf(best, ob);
        }
        if ((tid & 63) == 0) { wmaxs[tid >> 6] = best; wmax2[tid >> 6] = second; }
        __syncthreads();
        best = wmaxs[0]; second = wmax2[0];
        for (int w = 1; w < 4; ++w) { second = fmaxf(fminf(best, wmaxs[w]), fmaxf(second, wmax2[w])); best = fmaxf(best, wmaxs[w]); }
        if (p.margin && tid == 0) p.margin[bt] = best - second;
    }
    const float vmax = best;

    // ---- top-k threshold key
    unsigned tk = 0;
    if (p.top_k > 0 && p.top_k < p.V) {
        unsigned prefix = 0;
        int kk = p.top_k;
        for (int ps = 3; ps >= 0; --ps) {
            hist_cnt[tid] = 0;
            if (tid == 0) { sel_digit = 0u; sel_k = kk; }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const bool match = (ps == 3) || ((key[i] >> (8 * (ps + 1))) == prefix);
                if (match && i * 256 + tid < p.V) atomicAdd(&hist_cnt[(key[i] >> (8 * ps)) & 255], 1);
            }
            __syncthreads();
            const int dd = 255 - tid;                              // thread t owns digit 255-t: suffix counts become a prefix scan
            const int c = hist_cnt[dd];
            int tot;
            const int above = block_excl_scan<int>(c, wsum32, tot);   // elements with a larger digit
            if (above < kk && above + c >= kk) { sel_digit = (unsigned)dd; sel_k = kk - above; }
            __syncthreads();
            prefix = (prefix << 8) | sel_digit;
            kk = sel_k;
            __syncthreads();
        }
        tk = prefix;
    }
    // ---- quantised masses of the top-k set
    u64 wq[EPT];
    u64 local = 0;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const bool in = (i * 256 + tid < p.V) && key[i] >= tk;
        wq[i] = in ? (u64)(__expf(v[i] - vmax) * 1099511627776.0f) : 0ull;      // 2^40
        local += wq[i];
    }
    u64 Z;
    (void)block_excl_scan<u64>(local, wsum64, Z);
    // ---- nucleus threshold key: smallest key whose ascending cumulative mass exceeds (1 - top_p) * Z
    unsigned tp = 0;
    if (p.top_p > 0.f) {
        // (1 - top_p) in double and clamped: top_p < 2^-24 would round 1 - top_p to 1 in float (lim == Z: no bucket qualifies), and
        // top_p > 1 would cast a negative double to u64.  lim <= Z - 1 guarantees that exactly one bucket per pass satisfies the
        // selection below; Z > 0 always (the maximum carries mass 2^40).  The last (largest) element is therefore always kept, as
        // helpers.py:14 does.  Elements TIED with the cut value are all kept here (value threshold), where the reference cuts
        // inside the tie by sort order - a difference only for exactly equal logits.
        const double keep_from = fmin(fmax(1.0 - (double)p.top_p, 0.0), 1.0);
        u64 lim = (u64)(keep_from * (double)Z);
        if (lim >= Z) lim = Z - 1;
        unsigned prefix = 0;
        u64 below = 0;
        for (int ps = 3; ps >= 0; --ps) {
            hist_mass[tid] = 0ull;
            if (tid == 0) { sel_digit = 255u; sel_below = below; }      // defined even if no bucket were selected
            __syncthreads();
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const bool match = (ps == 3) || ((key[i] >> (8 * (ps + 1))) == prefix);
                if (match && wq[i]) atomicAdd(&hist_mass[(key[i] >> (8 * ps)) & 255], wq[i]);
            }
            __syncthreads();
            const u64 c = hist_mass[tid];
            u64 tot;
            const u64 before = below + block_excl_scan<u64>(c, wsum64, tot);   // mass of smaller digits (ascending)
            if (before <= lim && before + c > lim) { sel_digit = (unsigned)tid; sel_below = before; }
            __syncthreads();
            prefix = (prefix << 8) | sel_digit;
            below = sel_below;
            __syncthreads();
        }
        tp = prefix;
    }
    const unsigned thr = tk > tp ? tk : tp;
    u64 keepsum = 0;
    int nkeep = 0;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        if (!(key[i] >= thr && wq[i])) wq[i] = 0ull;
        keepsum += wq[i];
        nkeep += wq[i] ? 1 : 0;
    }
    u64 Zk;
    const u64 excl = block_excl_scan<u64>(keepsum, wsum64, Zk);
    if (p.kept) {
        int tot;
        (void)block_excl_scan<int>(nkeep, wsum32, tot);
        if (tid == 0) p.kept[bt] = tot;
    }
    for (int d = 0; d < p.n_draw; ++d) {
        unsigned long long h = splitmix64((p.seed + (p.seed_dev ? p.seed_dev[0] : 0ull)) ^ 0xC0FFEE1234ull);
        h = splitmix64(h ^ ((unsigned long long)p.stage << 48) ^ ((unsigned long long)((long)d * p.B + b) << 16) ^ (unsigned long long)t);
        const double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);       // 53 bits -> [0,1)
        u64 target = (u64)(u * (double)Zk);
        if (target >= Zk) target = Zk > 0 ? Zk - 1 : 0;
        if (excl <= target && target < excl + keepsum) {                        // exactly one thread
            u64 run = excl;
            int pick = tid;
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                run += wq[i];
                if (run > target) { pick = i * 256 + tid; break; }
            }
            p.idx_out[((long)d * p.B + b) * p.l + t] = pick;
        }
    }
    if constexpr (SOFT) {
    if (p.soft_out) {
        // Gumbel-softmax over the KEPT logits (the reference masks `logits` in place before it calls gumbel_softmax_with_rng), then the
        // expectation of the code vectors under it.  One pass per draw row: the reference draws fresh noise for every replicated row.
        __shared__ float red[4];
        __shared__ float accs[4][32];
        for (int d = 0; d < p.n_draw; ++d) {
            const long row = (long)d * p.B + b;
            float z[EPT];
            float zmax = -INFINITY;
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int e = i * 256 + tid;
                z[i] = -INFINITY;
                if (e < p.V && key[i] >= thr) {
                    float g;
                    if (p.gumbel) g = p.gumbel[(row * p.l + t) * (long)p.V + e];
                    else {
                        unsigned long long h = splitmix64((p.seed + (p.seed_dev ? p.seed_dev[0] : 0ull)) ^ 0x5EEDF00D77ull);
                        h = splitmix64(h ^ ((unsigned long long)p.stage << 52) ^ ((unsigned long long)row << 28) ^ ((unsigned long long)t << 12) ^ (unsigned long long)e);
                        const float u = ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);          // (0, 1)
                        g = -__logf(-__logf(u));                                                 // -log(Exp(1) sample)
                    }
                    z[i] = (v[i] * p.smooth_mul + g) / p.smooth_tau;
                    zmax = fmaxf(zmax, z[i]);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) zmax = fmaxf(zmax, __shfl_xor(zmax, o, 64));
            __syncthreads();
            if ((tid & 63) == 0) red[tid >> 6] = zmax;
            __syncthreads();
            zmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            float acc[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) acc[c] = 0.f;
            float wsum = 0.f;
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const float w = z[i] > -INFINITY ? __expf(z[i] - zmax) : 0.f;
                if (w > 0.f) {
                    wsum += w;
                    const float* er = p.codebook + (long)(i * 256 + tid) * p.Cvae;
                    for (int c = 0; c < p.Cvae; ++c) acc[c] = fmaf(w, er[c], acc[c]);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                wsum += __shfl_xor(wsum, o, 64);
#pragma unroll
                for (int c = 0; c < 32; ++c) acc[c] += __shfl_xor(acc[c], o, 64);
            }
            __syncthreads();
            if ((tid & 63) == 0) {
                red[tid >> 6] = wsum;
                for (int c = 0; c < p.Cvae; ++c) accs[tid >> 6][c] = acc[c];
            }
            __syncthreads();
            if (tid < p.Cvae) {
                const float tot = (red[0] + red[1]) + (red[2] + red[3]);
                const float a = (accs[0][tid] + accs[1][tid]) + (accs[2][tid] + accs[3][tid]);
                p.soft_out[(row * p.l + t) * (long)p.Cvae + tid] = a / tot;
            }
        }
    }
    }
}

extern "C" int cvar_cfg_sample(const float* logits, int B, int nrep, int l, int V, const float* coef_host,
                               int top_k, float top_p, uint64_t seed, const uint64_t* seed_dev, int stage, int n_draw,
                               int32_t* idx_out, float* combined, float* margin, int32_t* kept, int ldv,
                               const float* codebook, int Cvae, float smooth_mul, float smooth_tau, const float* gumbel, float* soft_out,
                               void* stream) {
    if (!logits || !coef_host || !idx_out || B <= 0 || l <= 0 || V <= 1) return CVAR_EINVAL;
    if (nrep < 1 || nrep > 4 || n_draw < 1 || n_draw > 4 || V > 4096) return CVAR_EUNSUPPORTED;
    if (ldv != 0 && ldv < V) return CVAR_EINVAL;
    if (soft_out && (!codebook || Cvae < 1 || Cvae > 32 || !(smooth_tau > 0.f))) return CVAR_EINVAL;
    if (soft_out && top_k == 1) return CVAR_EUNSUPPORTED;          // greedy: the masked softmax is one-hot, the soft embedding IS codebook[idx]
    SampleParams p;
    p.ldv = ldv ? ldv : V;
    p.codebook = codebook; p.Cvae = Cvae; p.smooth_mul = smooth_mul; p.smooth_tau = smooth_tau; p.gumbel = gumbel; p.soft_out = soft_out;
    p.logits = logits; p.B = B; p.nrep = nrep; p.l = l; p.V = V;
    for (int i = 0; i < 4; ++i) p.coef[i] = i < nrep ? coef_host[i] : 0.f;
    p.top_k = top_k; p.top_p = top_p; p.seed = seed; p.seed_dev = (const unsigned long long*)seed_dev; p.stage = stage; p.n_draw = n_draw;
    p.idx_out = idx_out; p.combined = combined; p.margin = margin; p.kept = kept;
    dim3 grid((unsigned)((long)B * l)), block(256);
    if (top_k == 1) hipLaunchKernelGGL(cfg_greedy_kernel, grid, block, 0, as_stream(stream), p);
    else if (soft_out) hipLaunchKernelGGL(cfg_sample_kernel<true>, grid, block, 0, as_stream(stream), p);
    else hipLaunchKernelGGL(cfg_sample_kernel<false>, grid, block, 0, as_stream(stream), p);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}
