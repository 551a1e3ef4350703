/* cvar.h - C ABI of libcvar_hip.so: the MI355X (gfx950) hot path of lxa9867/ControlVAR.
 *
 * The reference has no FFI; its operator boundary is Python:
 *   L2 model API  models/__init__.py:6-45, models/control_var.py:223-651, models/vqvae.py:73-104
 *   L0 op slots   models/basic_var.py:15-29 (flash_attn_func, fused_mlp_func, slow_attn, ...)
 * Each entry point below replaces the ATen / fused-op sequence cited next to it, and is what a
 * ctypes stub on the reference side binds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (no allocation inside the library), except the
 *     small parameter arrays suffixed `_host` (copied into the launch); row-major, dims passed explicitly; `stream` is a hipStream_t passed as void* (NULL = default);
 *   - all calls are asynchronous on `stream`, re-entrant, thread-safe for distinct streams; no global mutable state, no environment
 *     variables (workspaces and tuning knobs are arguments);
 *   - return 0 on success, a negative cvar_status otherwise (never aborts across the ABI);
 *   - dtype codes: CVAR_F32 (parity mode, exact-f32 MFMA) or CVAR_BF16 (throughput mode, f32 accumulate).
 */
#ifndef CVAR_H
#define CVAR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { CVAR_OK = 0, CVAR_EINVAL = -1, CVAR_EUNSUPPORTED = -2, CVAR_ELAUNCH = -3 } cvar_status;
typedef enum { CVAR_F32 = 0, CVAR_BF16 = 1 } cvar_dtype;
typedef enum { CVAR_ACT_NONE = 0, CVAR_ACT_GELU_TANH = 1, CVAR_ACT_GELU_GRAD = 2 } cvar_act;

int cvar_abi_version(void);                 /* bumps on any signature change */
const char* cvar_status_str(int status);

/* ---------------------------------------------------------------------------------------------
 * GEMM with fused epilogue, optionally an implicit-GEMM 3x3 convolution over NHWC activations.
 *   C[row(m), n] = cast( residual[m,n] + gate[m / gate_rows, n] * act(alpha * sum_k A[m,k] W[n,k] + bias[n]) )
 * Replaces: F.linear qkv/proj/fc1/fc2/head/ada_lin (basic_var.py:51,92,119,207; control_var.py:221,700),
 *           fused_mlp_func slot (basic_var.py:44-49), the gated residual x + gamma*f(x) (basic_var.py:208-209),
 *           nn.Conv2d 3x3/1x1 of the VQVAE (vae_modules.py:25,34,48-53,69-71,113,142,180,208; vqvae.py:48-49).
 * A and W have element type `dtype`, K contiguous.  conv: A is X[B][Hin][Win][Cin], W is [N][ky][kx][Cin]
 * (K = 9*Cin), output pixel m=(b,oy,ox); `up`=1 reads X through a nearest x2 upsample (vae_modules.py:28),
 * `stride`=2 is the asymmetric (0,1,0,1)-padded downsample (vae_modules.py:37).
 * Output row remap (remap_l > 0): row(m) = (m / remap_l) * remap_L + remap_off + m % remap_l  - used to write
 * the qkv rows of one scale straight into the per-sequence KV arena [R][Lmax][3C] (replaces torch.cat,
 * basic_var.py:106-108), or - with the column split below - k | v into a [R][Lmax][2C] arena and q next to it. */
typedef struct {
    int M, N, K;
    int dtype;                       /* cvar_dtype of A and W */
    const void* A; int64_t lda;      /* plain mode: row stride in elements (ignored for conv) */
    const void* W; int64_t ldw;
    int batch;                       /* >= 1; grid.z */
    int64_t strideA, strideW, strideC, strideR;   /* per-batch strides in elements */
    /* conv mode */
    int conv;                        /* 0 plain GEMM, 1 implicit 3x3 conv */
    int Hin, Win, Cin, Hout, Wout, stride, up;
    /* epilogue */
    float alpha;
    const float* bias;               /* [N] or NULL */
    int act;                         /* cvar_act */
    const float* gate; int64_t ldg; int gate_rows;   /* gate[(m / gate_rows) * ldg + n] or NULL */
    const void* residual; int res_dtype; int64_t ldr;
    void* C; int out_dtype; int64_t ldc;
    int remap_l, remap_L, remap_off;
    /* training-side fusions (all optional; leading dimension ldc):
     *   pre_act     also store alpha*acc + bias - the value BEFORE activation / gate / residual - in the OPERAND dtype: the tensor
     *               backward needs (fc1's pre-activation; the proj / fc2 branch output f of x + gate*f);
     *   aux         act == GELU_GRAD: multiply by gelu'(aux[m,n]) (dtype of C) instead of applying an activation:
     *               dH = (dY W2) * gelu'(A) (fc2 data gradient);
     *   gate_scale  per-gate-row multiplier of the gate (DropPath keep-scale per sample, basic_var.py:208-209 under training) */
    void* pre_act;
    const void* aux;
    const float* gate_scale;
    /* per-call execution options (ABI 10; zero = defaults).  Nothing here is process-wide: the library keeps no mutable state
     * and reads no environment variables.
     *   ws, ws_bytes  optional caller-owned, 16-byte aligned device workspace for split-K: fp32 partial tiles of small-M / long-K
     *                 GEMMs, summed in a fixed order by a second kernel that applies the epilogue.  NULL: never splits.  The caller
     *                 must not share one workspace between streams that may run concurrently;
     *   tile_cfg      0 automatic; 1 only 128x128 tiles; 2 the 8-wave 256x256 tile wherever it applies; 3 the 4-wave 256x256
     *                 tile (A/B measurements - results are bit-identical across tile choices of one split-K decision); 5: 3x3 convs
     *                 stay on the implicit-GEMM tiles, 6: eligible 3x3 convs take the LDS-halo kernel at any size (conv_halo.hip sums
     *                 K chunk-major instead of tap-major: same math, different fp32 rounding); 7 / 8: aliases of 2 / 3 (the A/B arms of the
     *                 round-4 persistent-kernel experiment, which is no longer in the library: experiments/README.md);
     *                 12: as 0, and a small-M bf16 call (M <= 512) may take the weight-streaming kernel of gemm_skinny.hip - whole K per workgroup, no
     *                 reduction launch (long K: a few K slices finished row-wise).  Same math; the fp32 summation order then depends on the plan chosen for
     *                 (M, N, K), so callers that promise bit-identical rows across batch sizes (the VQVAE) stay on 0.  The transformer's passes use 12
     *                 (which also gives mid-size calls - 64 < M <= 1024, or one round of 128x128 tiles - the three-LDS-stage tile instance);
     *                 13 / 14: the 128x128 tile with three / four LDS stages wherever that tile is chosen (A/B measurements); 26: the 128x128 tile on four
     *                 waves instead of eight (the form before the second half of round 4; A/B measurements, bit-identical);
     *                 27: the 256x192 tile (round 5) on every unsliced plain bf16 call - automatic plans (0 / 12) take it for launches whose 256x256 tiles would
     *                 fill their last round of the chip badly (N % 192 == 0, M >= 2048); bit-identical to the other tiles;
     *                 28: as 2, with the round-6 LDS-prefetched gate + residual epilogue (RPF) switched off - full 256x256 tiles of an fp32 gate + residual call then run the
     *                 register-prefetch epilogue that partial tiles always run; bit-identical (A/B measurements, tests);
     *   stagger       > 0: the first workgroup of every CU starts delayed by up to this many shader cycles (by its index), which
     *                 de-phases the output bursts of equally long tiles; 0: off.  Never changes results. */
    void* ws; int64_t ws_bytes;
    int tile_cfg;
    int stagger;
    int group_m;                     /* row tiles per scheduling group; 0 = automatic (4 for long-K / fp32-output GEMMs, else 8).  Never changes results */
    /* column split (ABI 11; split_n = 0: off): result columns [0, split_n) are stored to C_split[m * ld_split + n] (rows NOT
     * remapped), columns [split_n, N) to C at column n - split_n with the row remap.  The qkv GEMM of inference uses it to put the
     * queries of a scale into a scratch buffer and only k | v into the KV arena [R][Lmax][2C] (the reference caches k, v only:
     * basic_var.py:108-111).  Needs remap_l > 0, no activation / gate / residual, split_n, N, ldc, ld_split multiples of 8. */
    void* C_split; int split_n; int64_t ld_split;
    /* ABI 14: the values of the split columns [0, split_n) are multiplied by split_alpha after the bias (0 is read as 1).  Inference folds
     * softmax scale * log2(e) into the query rows here - ONE rounding of q * c instead of a multiply per score in the attention kernel
     * (cvar_attention_prescaled). */
    float split_alpha;
    /* ABI 17: adaLN of the FINISHED rows in the same call (ln_out != NULL): after C[m,:] is complete,
     *   ln_out[m,:] = cast( LN(C[m,:]) * (1 + ln_scale[(m / ln_rows) * ld_ln + :]) + ln_shift[(m / ln_rows) * ld_ln + :] )
     * - exactly the cvar_ln_modulate the next op would run on this GEMM's output (basic_var.py:208-209: the x that proj / fc2 update is the input of the
     * next ln_wo_grad).  Needs N = the row width (N % 4 == 0, N <= 2048), an fp32 C with ldc == N, no row remap / column split / conv / batch.  Calls that
     * are sliced along K (small M) are finished row-wise - slices summed, bias / gate / residual, LayerNorm and modulation in one kernel instead of the
     * reduction launch plus cvar_ln_modulate; every other call gets a cvar_ln_modulate launch behind it.  Same bits either way. */
    void* ln_out; int ln_out_dtype;
    const float* ln_scale; const float* ln_shift; int64_t ld_ln; int ln_rows; float ln_eps;
    /* ABI 18: GroupNorm statistics of the OUTPUT from the conv's own epilogue (gn_part != NULL; stride-1 3x3 bf16 convs that cvar_conv3x3_gn_partials accepts):
     * gn_part[((b * tiles + t) * N + c) * 3 + {0, 1, 2}] = (sum (y - piv), sum (y - piv)^2, piv) over the 256 pixels of output tile t of image b for channel c,
     * y = the values as stored (bf16), piv = the channel's value at one pixel of the tile.  cvar_groupnorm_silu_partials finishes them - the GroupNorm that
     * follows a ResnetBlock conv (vae_modules.py:40-66) then needs no pass over the tensor for its statistics.  A call that cannot emit them returns
     * CVAR_EUNSUPPORTED (never silently skips). */
    float* gn_part;
} cvar_gemm_desc;
/* Cache policy (round 6; nothing to set): a call whose output is >= 128 MB is treated as a stream - the specialised epilogues of the 256-row tiles store it non-temporally
 * and the fp32 gate + residual read-modify-write (proj / fc2) takes its LDS-prefetched form with non-temporal residual reads - so that it does not evict the operand panels
 * the CUs of an XCD share in their L2; smaller outputs keep the default policy.  Results do not depend on it (tests/test_gpu_kernels.py::test_gemm_streaming_policy_changes_no_bit). */
int cvar_gemm(const cvar_gemm_desc* d, void* stream);
/* 1 (and the partial geometry) when a conv of this shape honours cvar_gemm_desc.gn_part, else 0 */
int cvar_conv3x3_gn_partials(int dtype, int stride, int Cin, int Cout, int Hin, int Win, int Hout, int Wout, int* tiles_per_image, int* pixels_per_tile);

/* Weight-gradient GEMM on token-major operands (ABI 13): C[n][k] = sum_t A[t][n] * B[t][k], A = [T][lda], B = [T][ldb] bf16, C fp32 [Nn][ldc]
 * (dW = dY^T X, the parameter gradients of every nn.Linear under autograd, train_control_var_hpu.py:231).  No transposed copies: the fragments
 * are read from the [t][column] LDS tiles with gfx950's transpose-read.  Nn % 128 == 0, Kk % 128 == 0 (a half-filled last 256-wide column tile stores its first 128 columns: d30's C = 1920), lda / ldb % 8 == 0 (any T).
 * ws (optional, 16-byte aligned, ws_bytes): fp32 partial tiles of the token split, summed in a fixed order; NULL: one slice per tile.
 * colsum_a (optional, ABI 14): colsum_a[n] = sum_t A[t][n] for n < Nn - the bias gradient of the same nn.Linear (dY summed over the tokens),
 * taken from the A fragments on the matrix pipe instead of a second pass over dY (cvar_colsum). */
int cvar_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int T, int Nn, int Kk,
                 float* ws, int64_t ws_bytes, float* colsum_a, void* stream);

/* ---------------------------------------------------------------------------------------------
 * adaLN: out[m,:] = cast( LN(x[m,:]) * (1 + scale[m / rows_per, :]) + shift[m / rows_per, :] ), LN over C with
 * biased variance, no affine.  Replaces ln_wo_grad(x).mul(scale.add(1)).add_(shift) (basic_var.py:208-209;
 * control_var.py:701).  x fp32 [M,C]; scale/shift fp32 rows of stride ld_ada. */
int cvar_ln_modulate(const float* x, const float* scale, const float* shift, int64_t ld_ada, int rows_per,
                     void* out, int out_dtype, int M, int C, float eps, void* stream);

/* SiLU + cast (the nn.SiLU in front of every ada_lin, basic_var.py:198): out = cast(x * sigmoid(x)). */
int cvar_silu_cast(const float* x, void* out, int out_dtype, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-scale KV-cached attention (slow_attn / flash_attn_func slots, basic_var.py:106-117).
 * qkv arena: [R][Lmax][3*H*64] of `dtype` (q | k | v thirds, head-major inside a third), written by cvar_gemm
 * with the row remap.  Queries are rows [q_off, q_off+l) of every sequence, keys rows [0, kv_len(q)).
 * kv_len: if n_lvl == 0 every query sees [0, q_off+l) (inference, attn_bias=None); otherwise lvl_end[] holds the
 * n_lvl (<= 32) strictly increasing level ends and a query at position p sees keys < lvl_end[level(p)] - the block-causal
 * attn_bias_for_masking of training (control_var.py:158-168; levels = scales) and of `separate_decoding` (:170-180; levels =
 * half scales: a control token does not see the image half of its own scale).  hole_host (optional, 2*n_lvl ints): per level a key
 * range [lo, hi) in FRONT of the level that its queries do not see (lo >= hi: none) - the `indep` mask (:182-191), where the image
 * half of a scale is blind to the control half of the same scale.  out: [R*l][H*64] of `dtype`.
 * q (optional): when non-NULL, `qkv` is a K/V arena [R][Lmax][2*H*64] (k | v halves) and `q` holds the queries of THIS call,
 * [R][l][H*64] (row t of sequence r = position q_off + t).  Inference uses this form: the queries of a scale are dead once its
 * attention has run, so the cache keeps only K and V (the reference caches k, v only as well: basic_var.py:108-111). */
int cvar_attention(const void* qkv, const void* q, int dtype, int R, int H, int Lmax, int q_off, int l, float scale,
                   const int* lvl_end_host, int n_lvl, const int* hole_host, void* out,
                   float* lse /* optional [R][H][l], saved for backward */, void* stream);
/* same contract, always the exact row-per-lane fp32-math kernel (the parity-mode implementation; also the in-library
 * reference the bf16 MFMA flash kernel is A/B-tested against). */
int cvar_attention_rowwise(const void* qkv, const void* q, int dtype, int R, int H, int Lmax, int q_off, int l, float scale,
                           const int* lvl_end_host, int n_lvl, const int* hole_host, void* out, float* lse, void* stream);
/* ABI 14, inference form only (K/V arena in `kv` + the call's queries in `q`, bf16): the query rows already hold
 * q * scale * log2(e) (cvar_gemm_desc.split_alpha on the QKV GEMM, or cvar_cos_qk_norm's q_mul for cos-attention), so
 * softmax(q k^T scale) v = sum_k 2^(q' k) v / sum_k 2^(q' k)  (basic_var.py:99-117, same function).  The kernel subtracts the running maximum on
 * the matrix pipe (see attn.hip) - one vector operation less per score than cvar_attention. */
int cvar_attention_prescaled(const void* kv, const void* q, int dtype, int R, int H, int Lmax, int q_off, int l,
                             const int* lvl_end_host, int n_lvl, const int* hole_host, void* out, float* lse, void* stream);
/* the round-2 MFMA kernel behind cvar_attention's contract: kept as the in-library A/B reference of the round-3 kernel (tools/attn_bench.py) */
int cvar_attention_v1(const void* qkv, const void* q, int dtype, int R, int H, int Lmax, int q_off, int l, float scale,
                      const int* lvl_end_host, int n_lvl, const int* hole_host, void* out, float* lse, void* stream);
/* backward of the level-masked attention (training forward, control_var.py:626-639 under autograd): given dO and the saved
 * lse, writes dQ | dK | dV into dqkv with the arena layout [R][Lmax][3*H*64].  ws: R*H*l floats.  q_off must be 0. */
int cvar_attention_bwd(const void* qkv, int dtype, const void* o, const void* dout, const float* lse, int R, int H, int Lmax,
                       int q_off, int l, float scale, const int* lvl_end_host, int n_lvl, const int* hole_host, void* dqkv, float* ws, void* stream);
/* same contract, always the exact row-per-lane kernels (fp32-mode implementation / A-B reference of the bf16 MFMA backward) */
int cvar_attention_bwd_rowwise(const void* qkv, int dtype, const void* o, const void* dout, const float* lse, int R, int H, int Lmax,
                               int q_off, int l, float scale, const int* lvl_end_host, int n_lvl, const int* hole_host, void* dqkv, float* ws,
                               void* stream);

/* cos-attention pre-pass (basic_var.py:99-104), in place on rows [q_off, q_off+l) of the arena:
 * q = normalize(q) * exp(min(scale_mul[h], log 100)),  k = normalize(k).  q (optional): as for cvar_attention - K/V arena in
 * `qkv` + the call's queries [R][l][H*64] in `q`. */
int cvar_cos_qk_norm(void* qkv, void* q, int dtype, int R, int H, int Lmax, int q_off, int l, const float* scale_mul,
                     float* norms /* optional [R][l][H][2] = |q|, |k|, saved for training */,
                     float q_mul /* ABI 14: extra factor on the query side (log2(e) in front of cvar_attention_prescaled; 1 otherwise) */, void* stream);
/* backward of the pre-pass, in place on dqkv (arena layout, q_off 0): gradients w.r.t. the normalised q, k become gradients
 * w.r.t. the raw projections; dsm_tok[R*l][H] receives d loss / d scale_mul per token (summed over tokens by cvar_colsum). */
int cvar_cos_qk_norm_bwd(const void* qkv, void* dqkv, int dtype, int R, int H, int Lmax, int l, const float* scale_mul,
                         const float* norms, float* dsm_tok, void* stream);

/* ---------------------------------------------------------------------------------------------
 * CFG combine + sampling (control_var.py:295-307,501-505; helpers.py:6-19).
 * logits: [nrep*B][l][V] fp32 (row groups: cond, then unconditional variants); coef[nrep] combine weights
 * ((1+t, -t) or the 4-branch form).  top_k == 1: greedy argmax (lowest index on ties).  Otherwise top-k /
 * top-p filtering as the reference, then one draw per row of `n_draw` independent rows from a counter-based
 * generator keyed by (seed, stage, row).  idx_out: [n_draw*B][l] int32.  Optional outputs (may be NULL):
 * combined [B][l][V] fp32, margin [B][l] fp32 (top1 - top2 of the combined logits), kept [B][l] int32.
 * ldv: row stride of `logits` in floats (0 = V): a head with extra columns behind the V codes (`separator`, control_var.py:202,504).
 * more_smooth (control_var.py:326-330,459-463,511-515; helpers.py:22-36), sampling mode only: with soft_out != NULL also
 *   soft_out[d*B + b][t][:] = softmax_v((combined_v * smooth_mul + g_v) / smooth_tau) . codebook[v][:]  over the top-k / top-p KEPT v
 * (the reference masks its logits in place before the Gumbel softmax); g = Gumbel noise, injected ([n_draw*B][l][V]) or drawn from
 * the counter-based generator. */
int cvar_cfg_sample(const float* logits, int B, int nrep, int l, int V, const float* coef_host,
                    int top_k, float top_p, uint64_t seed, const uint64_t* seed_dev /* optional, added to seed */, int stage, int n_draw,
                    int32_t* idx_out, float* combined, float* margin, int32_t* kept, int ldv,
                    const float* codebook /* [V][Cvae] */, int Cvae, float smooth_mul, float smooth_tau, const float* gumbel, float* soft_out,
                    void* stream);

/* ---------------------------------------------------------------------------------------------
 * Token-pyramid helpers of VectorQuantizer2 (models/quant.py).
 * Operator tables (host-computed, device-resident fp32): up[k]  = bicubic (S x pn_k), down[k] = area (pn_k x S),
 * packed back to back in pyramid order (see controlvar_amd/pyramid.py).
 *
 * cvar_ms_next_input: get_next_autoregressive_input (quant.py:243-260) for `nmaps` maps per batch row:
 *   f_hat[b][map] += phi_si( bicubic( E[idx[b][map*pn^2 ...]] ) );  tok_out[b][map*pn'^2 + t][:] = area(f_hat, pn')
 *   (tok_out may be NULL at the last scale).  idx int32 [nb][nmaps*pn*pn]; f_hat fp32 [nb][nmaps][Cvae][S][S]. */
int cvar_ms_next_input(const int32_t* idx, const float* codebook, const float* phi_w, const float* phi_b,
                       const float* up_mat, const float* down_mat, float* f_hat, float* tok_out,
                       int nb, int nmaps, int pn, int pn_next, int S, int Cvae, void* stream);

/* cvar_ms_encode: f_to_idxBl_or_fhat (quant.py:184-215): 10-stage residual nearest-code quantisation of
 * f [B][Cvae][S][S] fp32.  idx_out int32 [B][sum pn^2] (scale-major); f_hat_out (optional) [B][Cvae][S][S];
 * margin_out (optional) [B][sum pn^2] = second-best minus best distance. */
int cvar_ms_encode(const float* f, const float* codebook, int V, const float* phi_w, const float* phi_b,
                   const int* phi_map_host, const int* patch_nums_host, int nscale, const float* up_mats,
                   const float* down_mats, int32_t* idx_out, float* f_hat_out, float* margin_out,
                   int B, int S, int Cvae, void* stream);

/* word_embed + level/position embedding of the next scale (control_var.py:534-560):
 * x[rep*nb + b][x_off + t][:] = tok[b][t][:] @ W^T + bias + lvl_pos[t][:], for rep in [0, nrep); x has x_rows
 * rows per sequence (x_rows = l, x_off = 0 for one scale; x_rows = L for the teacher-forced forward).  fp32. */
int cvar_word_embed(const float* tok, const float* W, const float* bias, const float* lvl_pos, float* x,
                    int nb, int nrep, int l, int Cvae, int C, int x_rows, int x_off, void* stream);

/* first-scale tokens (control_var.py:381-409): x[r][j][:] = src_j[id_j[r]][:] + pos_start[j][:] + lvl_pos[j][:]
 * with src_0 = cond_embed / class_emb(mf=1), src_1 = class_emb; also cond[r][:] = class_emb[labels[r]][:]. */
int cvar_first_tokens(const float* class_emb, const float* cond_embed, const int32_t* labels, const int32_t* types,
                      const float* pos_start, const float* lvl_pos, float* x, float* cond, int R, int first_l,
                      int C, int x_rows, void* stream);

/* ---------------------------------------------------------------------------------------------
 * VQVAE conv-stack helpers (models/vae_modules.py).
 * GroupNorm(32 groups, eps, affine) [+ SiLU] over NHWC activations (vae_modules.py:18-19,58-59):
 * ws: caller workspace of cvar_groupnorm_ws_bytes(B, HW, C) bytes (per-chunk partial sums, reduced in a fixed
 * order: the result is bit-reproducible run to run). */
int64_t cvar_groupnorm_ws_bytes(int B, int HW, int C);
int cvar_groupnorm_silu(const void* x, int dtype, const float* weight, const float* bias, void* out,
                        int B, int HW, int C, int groups, float eps, int silu, void* ws, void* stream);
/* The same GroupNorm with its statistics taken from per-tile partials (cvar_gemm_desc.gn_part of the conv that produced x; ABI 18): tiles are combined in
 * double precision in a fixed order, then out = silu?(x * a_c + d_c) as above.  ws: cvar_groupnorm_ws_bytes(B, HW, C) bytes (only the coefficient part is used). */
int cvar_groupnorm_silu_partials(const void* x, int dtype, const float* weight, const float* bias, void* out, int B, int HW, int C, int groups, float eps,
                                 int silu, const float* gn_part, int tiles_per_image, int pixels_per_tile, void* ws, void* stream);
/* Split-bf16 operands ("bf16x3", ABI 20) - the middle precision of the VQVAE ENCODER (vae_modules.py:144-160 run in fp32 by the reference; quant.py:196-213 then
 * takes the nearest code of every feature, so encoder noise moves ids).  An fp32 value x is carried as hi = bf16(x), lo = bf16(x - hi) and a product as
 * a_hi w_hi + a_lo w_hi + a_hi w_lo on the bf16 matrix pipe with fp32 accumulation (relative error ~2^-16 instead of bf16's 2^-9).  The three products are three
 * K segments of ONE bf16 conv / GEMM (cvar_gemm, dtype bf16, fp32 output + fp32 residual): activation rows [hi(C) | lo(C) | hi(C)], weight rows [w_hi | w_hi | w_lo]
 * per tap.  cvar_split3 makes the activation rows from an fp32 tensor x[M][ldx] (Cpad >= 3 C: zero padding behind them, multiples of 4);
 * cvar_groupnorm_silu_split3 from GroupNorm(+SiLU) of an fp32 NHWC tensor (fp32 arithmetic, exact quotient - the parity mode's GroupNorm; ws as cvar_groupnorm_silu). */
int cvar_split3(const float* x, int64_t ldx, void* out_bf16, int64_t M, int C, int Cpad, void* stream);
int cvar_groupnorm_silu_split3(const float* x, const float* weight, const float* bias, void* out_bf16, int B, int HW, int C, int groups, float eps, int silu,
                               void* ws, void* stream);
/* row softmax of fp32 scores -> dtype probabilities (AttnBlock, vae_modules.py:84). */
int cvar_softmax_rows(const float* s, void* p, int out_dtype, int rows, int cols, void* stream);
/* [B][n][c] -> [B][c][n] transpose of `dtype` (V operand of AttnBlock's second bmm, vae_modules.py:87-89). */
int cvar_transpose(const void* in, void* out, int dtype, int B, int n, int c, int64_t ld_in, int64_t ld_out /* >= n */, void* stream);
/* NCHW fp32 -> NHWC dtype with channel padding to Cpad (zero filled). */
int cvar_nchw_to_nhwc(const float* in, void* out, int dtype, int B, int C, int HW, int Cpad, void* stream);
/* NHWC (ld = ldc) -> NCHW fp32 with y = clamp(x, lo, hi) * mul + add  (vqvae.py:89; control_var.py:563-564). */
int cvar_nhwc_to_nchw(const void* in, int dtype, int64_t ld_in, float* out, int B, int C, int HW,
                      float lo, float hi, float mul, float add, void* stream);

/* Low-resolution pyramid reconstruction, embed_to_fhat(all_to_max_scale=False) (quant.py:171-180; idxBl_to_img(same_shape=False),
 * vqvae.py:97-104) - ABI 12.  embed_rows: out[n][:] = codebook[idx[n]][:] (the nn.Embedding lookup, vqvae.py:103; ids clamped to [0, V)).
 * resample_sep: NHWC fp32 out[b][y][x][c] = sum_ij wy[y][i] wx[x][j] in[b][i][j][c] with dense (H x h), (W x w) matrices - F.interpolate
 * (bicubic / area) as the host tables of controlvar_amd/pyramid.py state it. */
int cvar_embed_rows(const int32_t* idx, const float* codebook, int V, float* out, int64_t n, int C, void* stream);
int cvar_resample_sep(const float* in, const float* wy, const float* wx, float* out, int B, int h, int w, int H, int W, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training step (train_control_var_hpu.py:207-250 under autograd; SURVEY.md 8a A5 backward / A20).  The backward
 * GEMMs reuse cvar_gemm on transposed operands; these are the remaining pieces.  All reductions have a fixed order.
 * x[m,:] += gate[m / gate_rows,:] * rowscale[m / gate_rows] * f[m,:]   (x + drop_path(gamma * f), basic_var.py:208-209) */
int cvar_gate_residual(float* x, const void* f, int dtype, const float* gate, int64_t ldg, int gate_rows,
                       const float* rowscale, int64_t M, int C, void* stream);
/* floats of workspace the two per-sequence reductions below may use for M = R * l rows (row statistics + one partial row per row
 * segment: the sequences are cut into enough segments that R * segments blocks fill the 256 CUs) */
int64_t cvar_train_ws_floats(int64_t M, int R, int C);
/* df = dx * gate * rowscale;  dgate[r,:] = rowscale[r] * sum_{m in r} dx[m,:] * f[m,:].  ws: cvar_train_ws_floats(R*l, R, C) floats;
 * ws_floats (ABI 16) = floats the caller allocated: smaller than that -> CVAR_EINVAL instead of an out-of-bounds device write. */
int cvar_gated_grad(const float* dx, const void* f, int dtype, const float* gate, int64_t ldg, const float* rowscale,
                    void* df, float* dgate, int64_t ldo, int R, int l, int C, float* ws, int64_t ws_floats, void* stream);
int cvar_gelu(const void* a, void* h, int dtype, int64_t n, void* stream);            /* h = gelu_tanh(a) */
int cvar_gelu_bwd(const void* a, void* dh, int dtype, int64_t n, void* stream);       /* dh *= gelu_tanh'(a) */
/* backward of cvar_ln_modulate: dx_out = dx_in + dLN(dy * (1+scale)); dscale[r,:] = sum dy*xhat; dshift[r,:] = sum dy.
 * bf16 dy: ONE pass over x, dy, dx_in (row in registers, column sums carried per wave, folded through LDS in a fixed order).
 * ws: cvar_train_ws_floats(M, M / rows_per, C) floats, ws_floats as above (ABI 16). */
int cvar_ln_modulate_bwd(const float* x, const void* dy, int dtype, const float* scale, int64_t ld_ada, int rows_per,
                         const float* dx_in, float* dx_out, float* dscale, float* dshift, int64_t ldo,
                         int M, int C, float eps, float* ws, int64_t ws_floats, void* stream);
/* ABI 14.  Gradient of word_embed = nn.Linear(Cvae, C) (control_var.py:74; its input is the token tensor of idxBl_to_var_input) from the token-major
 * fp32 tensors in place: dW[c][j] = sum_t dx[row(t)][c] * tok[t][j], db[c] = sum_t dx[row(t)][c], t < B * n_per_sample, row(t) = (t / n) *
 * rows_per_sample + skip + t % n (the first `skip` positions of every sample are not word-embedded).  Cvae == 32, C % 64 == 0.
 * ws: cvar_wordembed_grad_ws_bytes(B * n_per_sample, C) bytes (per-slice partials, summed in a fixed order). */
int64_t cvar_wordembed_grad_ws_bytes(int64_t ntok, int C);
int cvar_wordembed_grad(const float* dx, int64_t ldx, int rows_per_sample, int skip, const float* tok, int n_per_sample, int B, int C, int Cvae,
                        float* dW, float* db, float* ws, void* stream);
/* out[n] (+)= sum_m A[m,n]  (bias gradients).  ws: 64*N floats. */
int cvar_colsum(const void* A, int dtype, int64_t lda, float* out, int64_t M, int N, int accumulate, float* ws, void* stream);
/* out[r] (+)= sum_j A[r][j], j < ncols (16-byte aligned rows): the same bias gradient read from the transposed dY that the
 * weight-gradient GEMM needs anyway - contiguous rows instead of a strided column walk. */
int cvar_rowsum(const void* A, int dtype, int64_t lda, float* out, int nrows, int ncols, int accumulate, void* stream);
/* token cross-entropy (CrossEntropyLoss(reduction='none'), train_control_var_hpu.py:135,231) fused with its gradient:
 * loss_tok[m] = logsumexp(logits[m,:]) - logits[m,target[m]];  dlogits = (softmax - onehot) * weight[m] * gscale (optional). */
int cvar_ce_fwd_bwd(const float* logits, const int32_t* target, const float* weight, float gscale, float* loss_tok,
                    void* dlogits, int out_dtype, int64_t M, int V, void* stream);
/* dst[idx[i],:] += src[i,:] in row order (nn.Embedding gradients of class_emb / cond_embed). */
int cvar_scatter_add_rows(const float* src, int64_t ld_src, const int32_t* idx, float* dst, int n, int C, void* stream);
int cvar_silu_bwd(const float* cond, const float* dsilu, float* dcond, int64_t n, void* stream);
/* torch.optim.AdamW step of one tensor (train_control_var_hpu.py:631-633); gradient scaled by gscale * gscale_dev[0]
 * (all-reduce mean and clip_grad_norm_ coefficient folded in). */
int cvar_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
               float weight_decay, int step, const float* gscale_dev, float gscale, void* stream);
/* gradient norm (clip_grad_norm_, train_control_var_hpu.py:244-245): 256 double partial sums per tensor, then
 * out2 = {pre_scale * sqrt(sum), min(1, max_norm / (norm + 1e-6))}. */
int cvar_sumsq(const float* x, int64_t n, double* partial256, void* stream);
int cvar_clip_coef(const double* partials, int64_t count, float pre_scale, float max_norm, float* out2, void* stream);
/* Multi-tensor forms of the two above: ONE launch over a device table of cvar_adam_tensor (a d24 model has ~830 parameters).
 * Per tensor the arithmetic and summation order are exactly cvar_sumsq's / cvar_adamw's; partials is [n_tensors][256];
 * lr / weight decay come per parameter group (<= 8 groups, utils/lr_control.py:67-101) as small host arrays. */
/* w16 (may be NULL): a bf16 array of n elements that receives the rounded (nearest-even) updated parameter in the same pass - the
 * stacked GEMM-ready copy of a weight matrix, so a step does not re-read 4 GB of fp32 masters to rebuild 2 GB of bf16 weights. */
typedef struct { float* p; const float* g; float* m; float* v; int64_t n; int32_t group; int32_t pad; uint16_t* w16; } cvar_adam_tensor;
int cvar_sumsq_multi(const void* table_dev /* cvar_adam_tensor[n_tensors] */, int n_tensors, double* partials, void* stream);
int cvar_adamw_multi(const void* table_dev, int n_tensors, const float* lr_by_group_host, const float* wd_by_group_host, int n_groups,
                     float beta1, float beta2, float eps, int step, const float* gscale_dev, float gscale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Input pipeline of the tokenizer: what datasets/imagenetC.py:128-188 and
 * datasets/transforms_image.py:103-121 do on the CPU with PIL / torchvision.
 *
 * cvar_resample_u8: ONE pass of PIL's 8-bit separable resampler (Pillow src/libImaging/Resample.c,
 *   ImagingResampleHorizontal_8bpc / ImagingResampleVertical_8bpc) over an interleaved uint8 image
 *   src[src_h][src_w][channels].  axis 0 resamples rows to `dst_extent` columns, axis 1 columns to
 *   `dst_extent` rows.  bounds[dst_extent][2] = (first source index, tap count), coeffs[dst_extent][ksize] =
 *   fixed-point taps with 22 fraction bits - both DEVICE arrays filled from the host tables of
 *   precompute_coeffs / normalize_coeffs_8bpc (controlvar_amd/preprocess.py).  Replaces Image.resize
 *   (F.resize LANCZOS, transforms_image.py:16-18; cond.resize(image.size), imagenetC.py:147).
 * cvar_crop_flip_normalize: F.crop / F.center_crop + F.hflip + to_tensor + normalize(0.5, 0.5)
 *   (transforms_image.py:22-66,84-89): window (top, left, out_h, out_w) of src -> fp32 dst[channels][out_h][out_w].
 * cvar_ignore_mask: loss-ignore weights of a segmentation-mask condition (imagenetC.py:152-185):
 *   cond (B,3,H,W) fp32 in [-1,1]; out (B, L) fp32, L = sum 2*pn^2; the control half of every scale with index
 *   >= first_masked_scale carries the nearest-downsampled (background ? 0 : 1) map, everything else 1.
 *   image_first = 0: token order [control | image] per scale ('ignore_mask'); 1: [image | control] ('ignore_mask_').
 * cvar_rle_paint: raster half of process_anns (imagenetC.py:15-29): n_ann column-major COCO-RLE masks given as exclusive
 *   prefix sums of their runs (run_ends, annotation a = [ann_offsets[a], ann_offsets[a+1])), painted in order onto a black
 *   out[H][W][3] with colours[a][3]; later annotations overwrite earlier ones.  (RLE string decoding, the area filter and the
 *   centroid -> colour index rule are host code: controlvar_amd/preprocess.py.)
 */
int cvar_resample_u8(const void* src, int src_h, int src_w, int channels, int axis, int dst_extent,
                     const int* bounds, const int* coeffs, int ksize, void* dst, void* stream);
int cvar_crop_flip_normalize(const void* src, int src_h, int src_w, int channels, int top, int left, int out_h, int out_w,
                             int flip, float* dst, void* stream);
int cvar_ignore_mask(const float* cond, int B, int H, int W, const int* patch_nums_host, int n_scales, int first_masked_scale,
                     int image_first, float* out, int L, void* stream);
int cvar_rle_paint(const int* run_ends, const int* ann_offsets, const void* colours, int n_ann, int H, int W, void* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement aid (ABI 19; bench.py's roofline.sustained_*; replaces nothing in the reference): a register-fed stream of the product GEMM's MFMA
 * (v_mfma_f32_16x16x32_bf16, 16 independent accumulators per wave, two waves per SIMD on every CU, no memory traffic in the loop) on the caller's
 * operand values - >= 256 KB of bf16, 16-byte aligned.  MI355X clocks to its power budget, so what this launch reaches on operands of the bench's
 * kind is the ceiling a GEMM kernel can approach by scheduling alone on that device; on zeros it reaches the 2.4 GHz peak.
 * cvar_probe_mfma_flops(iters) = flop of one launch. */
int cvar_probe_mfma_bf16(const void* operands, int64_t operand_bytes, int iters, float* sink, void* stream);
/* ABI 20: the same stream on v_mfma_f32_32x32x16_bf16 (2 x 2 blocks of 32x32, same flop count per round); sink: 3 floats - [1] s_memtime ticks, [2] s_memrealtime
 * ticks (100 MHz) of one wave's loop.  Tells issue rate from clock beside the 16x16x32 probe (bench.py roofline.telemetry_probe). */
int cvar_probe_mfma_bf16_32x32(const void* operands, int64_t operand_bytes, int iters, float* sink, void* stream);
double cvar_probe_mfma_flops(int iters);

#ifdef __cplusplus
}
#endif
#endif /* CVAR_H */
