// Internal declarations for the transformer-denoiser kernels.
#pragma once
#include <hip/hip_runtime.h>

struct GemmArgs {
    const float* A; long a_rs, a_cs;   // A(i,k) = A[i*a_rs + k*a_cs]
    const float* B; long b_rs, b_cs;   // B(k,j) = B[k*b_rs + j*b_cs]
    float* C; long c_rs;               // C[i*c_rs + j]
    const float* bias;                 // [N] or nullptr
    int M, N, K;
    const float* res;                  // [M][c_rs] residual added before relu, may alias C; or nullptr
    int relu;
    int splitk; long c_split_stride;   // splitk > 1: partial z goes to C + z*c_split_stride (raw)
};
int tdm_launch_gemm(const GemmArgs& g, hipStream_t st);

// bf16 MFMA GEMMs (gemm_bf16.hip); nprod = 3 (hi/lo split operands, ~1e-5) or 1 (plain bf16 operands)
int tdm_launch_gemm_nt_bf16(const GemmArgs& g, int nprod, hipStream_t st);
int tdm_launch_gemm_tn_bf16(const GemmArgs& g, int nprod, hipStream_t st);
int tdm_launch_transpose(const float* in, float* out, int R, int Cn, hipStream_t st);
