// Shared device helpers for libcvar_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cvar.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;    // 8 bf16 = one 16-byte MFMA fragment
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef unsigned short bf16_t;                                  // raw bf16 bits

#define CVAR_CHECK_LAUNCH()                                      \
    do {                                                         \
        hipError_t e__ = hipGetLastError();                      \
        if (e__ != hipSuccess) return CVAR_ELAUNCH;              \
    } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t b) { return __uint_as_float(((unsigned)b) << 16); }

// f32 -> bf16, round-to-nearest-even (same rounding as torch's .to(bfloat16)): gfx950 has v_cvt_pk_bf16_f32
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2n_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    const bf16x2n_t r = __builtin_convertvector(v, bf16x2n_t);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ bf16x8_t pack_bf16x8(const float* v) {
    const u32x4_t u = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    return __builtin_bit_cast(bf16x8_t, u);
}
__device__ __forceinline__ bf16x4_t pack_bf16x4(const float* v) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
    const u32x2_t u = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    return __builtin_bit_cast(bf16x4_t, u);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kDtype = CVAR_F32;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int kDtype = CVAR_BF16;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

__device__ __forceinline__ float ld_any(const void* p, int dtype, int64_t i) {
    return dtype == CVAR_BF16 ? bf16_to_f32(((const bf16_t*)p)[i]) : ((const float*)p)[i];
}
__device__ __forceinline__ void st_any(void* p, int dtype, int64_t i, float v) {
    if (dtype == CVAR_BF16) ((bf16_t*)p)[i] = f32_to_bf16(v);
    else ((float*)p)[i] = v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float gelu_tanh_f(float t) {
    // 0.5 t (1 + tanh(u)), u = sqrt(2/pi) (t + 0.044715 t^3)   (nn.GELU(approximate='tanh'), basic_var.py:39)
    // evaluated as t * sigmoid(2u) = t / (1 + exp(-2u)): one v_exp_f32 + one v_rcp_f32 instead of tanhf's long path
    const float k2 = 2.0f * 0.7978845608028654f;
    const float u2 = k2 * (t + 0.044715f * t * t * t);
    return t / (1.0f + __expf(-u2));
}

__device__ __forceinline__ float gelu_tanh_fast(float t) {
    // same function for the bf16 mode (the result is rounded to bf16 right away): v_exp_f32 and v_rcp_f32 directly,
    // t * rcp(1 + 2^(-2u log2 e)) - about 1 ulp of f32 each, far inside the bf16 rounding step
    const float c1 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, c3 = c1 * 0.044715f;
    const float t2 = t * t;
    const float arg = t * (c1 + c3 * t2);
    return t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(arg));
}

__device__ __forceinline__ float gelu_tanh_grad(float t) {
    // g(t) = t * sig(2u), u = k (t + 0.044715 t^3)  ->  g' = sig + t * sig * (1 - sig) * 2k (1 + 3*0.044715 t^2)
    const float k2 = 2.0f * 0.7978845608028654f;
    const float u2 = k2 * (t + 0.044715f * t * t * t);
    const float sg = 1.0f / (1.0f + __expf(-u2));
    return sg + t * sg * (1.0f - sg) * k2 * (1.0f + 0.134145f * t * t);
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// conv_halo.hip: 3x3 stride-1 NHWC bf16 conv with the input tile + halo resident in LDS (dispatched from cvar_gemm; not part of the C ABI)
int cvar_conv3x3_halo_bf16(const void* X, const void* Wt, const float* bias, const void* residual, void* out, int out_f32, int B, int H, int W, int Cin,
                           int Cout, int up, float* gn_part, hipStream_t st);
// conv_c8.hip: the same for an image of <= 8 (padded) input channels into 160-multiples of couts - the VQVAE encoder's conv_in (dispatched from cvar_gemm)
int cvar_conv3x3_c8_bf16(const void* X, const void* Wt, const float* bias, void* out, int B, int H, int W, int Cout, float* gn_part, hipStream_t st);
