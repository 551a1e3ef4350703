// bf16 implicit-GEMM convolution instantiations of the GEMM kernel: a separate translation unit so that it compiles in parallel
// with the bf16 GEMM kernels (gemm.hip) and the fp32 ones (gemm_f32.hip).
#define CVAR_GEMM_CONV_TU 1
#include "gemm.hip"
