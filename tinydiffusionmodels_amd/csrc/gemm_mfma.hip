// General strided GEMM on fp32-input MFMA for the transformer denoiser
// (src/shakespeare.py:105-120: in_proj / out_proj / FFN linears and their
// gradients):   C[i][j] (+)= sum_k A(i,k) * B(k,j) (+ bias[j]) (relu)
// with A(i,k) = A[i*a_rs + k*a_cs], B(k,j) = B[k*b_rs + j*b_cs]; each operand
// must be contiguous along one of its two dimensions.  That covers
//   forward   Y = X W^T        (A k-contiguous, B k-contiguous)
//   dgrad     dX = dY W        (A k-contiguous, B j-contiguous)
//   wgrad     dW = dY^T X      (A i-contiguous, B j-contiguous; split over K = tokens)
// Tile 128x128x16, 4 waves (2x2), each wave 2x2 MFMA 32x32 tiles.  LDS tiles are
// [k][row] with a 132-float row pitch so MFMA operand reads (one dword per lane,
// lanes = consecutive rows) are bank-conflict free.  Exact fp32 (MFMA = fmaf chain).
#include "tdm_common.h"
#include "tdm_transformer.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDT = 132;

// stage a [128 rows][16 k] operand tile into LDS as T[k][row]
// element(row, k) = P[row*rs + k*cs]; rows valid < R, k valid < Kend
__device__ __forceinline__ void stage_tile(float* T, const float* __restrict__ P, long rs, long cs, int row0, int R,
                                           int k0, int Kend, int tid) {
    if (cs == 1) {  // k-contiguous: float4 along k, transposing scatter into LDS
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int f = tid + 256 * p;
            const int row = f >> 2, kq = f & 3;
            const int gr = row0 + row, gk = k0 + kq * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr < R) {
                const float* src = P + (long)gr * rs + gk;
                if (gk + 3 < Kend) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (gk + 0 < Kend) v.x = src[0];
                    if (gk + 1 < Kend) v.y = src[1];
                    if (gk + 2 < Kend) v.z = src[2];
                }
            }
            T[(kq * 4 + 0) * LDT + row] = v.x;
            T[(kq * 4 + 1) * LDT + row] = v.y;
            T[(kq * 4 + 2) * LDT + row] = v.z;
            T[(kq * 4 + 3) * LDT + row] = v.w;
        }
    } else {  // row-contiguous (rs == 1): float4 along rows, straight copy
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int f = tid + 256 * p;
            const int kk = f >> 5, r4 = f & 31;
            const int gk = k0 + kk, gr = row0 + r4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gk < Kend) {
                const float* src = P + (long)gk * cs + gr;
                if (gr + 3 < R) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (gr + 0 < R) v.x = src[0];
                    if (gr + 1 < R) v.y = src[1];
                    if (gr + 2 < R) v.z = src[2];
                }
            }
            *reinterpret_cast<float4*>(T + kk * LDT + r4 * 4) = v;
        }
    }
}

__global__ __launch_bounds__(256) void gemm_mfma_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[BK * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = blockIdx.y * BM, j0 = blockIdx.x * BN;
    int kbeg = 0, kend = g.K;
    if (g.splitk > 1) {
        const int chunk = ((g.K + g.splitk - 1) / g.splitk + BK - 1) / BK * BK;
        kbeg = blockIdx.z * chunk;
        kend = min(g.K, kbeg + chunk);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();
        stage_tile(As, g.A, g.a_rs, g.a_cs, i0, g.M, k0, kend, tid);
        stage_tile(Bs, g.B, g.b_cs, g.b_rs, j0, g.N, k0, kend, tid);  // B(k,j): "row" = j, stride b_cs; k stride b_rs
        __syncthreads();
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            const int kk = 2 * kp + h;
            const float a0 = As[kk * LDT + wm * 64 + j];
            const float a1 = As[kk * LDT + wm * 64 + 32 + j];
            const float b0 = Bs[kk * LDT + wn * 64 + j];
            const float b1 = Bs[kk * LDT + wn * 64 + 32 + j];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    float* C = g.C + (g.splitk > 1 ? (long)blockIdx.z * g.c_split_stride : 0L);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = j0 + wn * 64 + nt * 32 + j;
            if (col < g.N) {
                const float bz = (g.bias != nullptr) ? g.bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < g.M) {
                        float v = acc[mt][nt][r] + bz;
                        const long o = (long)row * g.c_rs + col;
                        if (g.res != nullptr) v += g.res[o];
                        if (g.relu) v = (v < 0.f) ? 0.f : v;
                        if (g.gate != nullptr) v = g.gate[o] > 0.f ? v * g.gate_scale : 0.f;
                        if (g.drop.thr != 0u)
                            v = tdm_keep(g.drop, (unsigned long long)row * (unsigned)g.N + (unsigned)col) ? v * g.drop.scale : 0.f;
                        C[o] = v;
                    }
                }
            }
        }
}

}  // namespace

int tdm_launch_gemm(const GemmArgs& g, hipStream_t st) {
    TDM_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "gemm: empty problem %dx%dx%d", g.M, g.N, g.K);
    TDM_REQUIRE(g.a_cs == 1 || g.a_rs == 1, "gemm: A must be contiguous along one dimension");
    TDM_REQUIRE(g.b_cs == 1 || g.b_rs == 1, "gemm: B must be contiguous along one dimension");
    // float4 loads need 16-byte aligned rows
    TDM_REQUIRE(((g.a_cs == 1 ? g.a_rs : g.a_cs) % 4) == 0 && ((g.b_cs == 1 ? g.b_rs : g.b_cs) % 4) == 0,
                "gemm: leading dimensions must be multiples of 4 floats");
    TDM_REQUIRE((((uintptr_t)g.A | (uintptr_t)g.B) & 15) == 0, "gemm: operands must be 16-byte aligned");
    TDM_REQUIRE(g.colsum == nullptr, "gemm: fused column sums exist in the bf16 TN kernel only");
    const int sk = g.splitk > 1 ? g.splitk : 1;
    TDM_REQUIRE(sk == 1 || (g.bias == nullptr && !g.relu && g.res == nullptr && g.gate == nullptr && g.drop.thr == 0u),
                "gemm: split-K output must be raw");
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, sk);
    hipLaunchKernelGGL(gemm_mfma_kernel, grid, dim3(256), 0, st, g);
    TDM_CHECK_LAUNCH("gemm_mfma");
    return 0;
}
