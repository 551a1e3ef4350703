// fp32 (parity mode) instantiations of the GEMM / implicit-conv kernel: a separate translation unit so that it compiles in
// parallel with the bf16 kernels (controlvar_amd/build.py builds all sources concurrently).
#define CVAR_GEMM_F32_TU 1
#include "gemm.hip"
