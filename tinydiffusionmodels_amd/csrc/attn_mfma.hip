// Softmax self-attention of nn.TransformerEncoderLayer (src/shakespeare.py:108-111;
// no mask, dropout on the probabilities in train mode) forward and backward on the
// fp32-input matrix cores (v_mfma_f32_32x32x2_f32: bitwise an fp32 fmaf chain, so the
// result stays within fp32 rounding of the reference).
//
// Everything is computed TRANSPOSED so that a lane owns one query (forward, dQ) or one
// key (dK/dV) and its accumulator registers run over the other index:
//   S^T[key][query] = K Q^T      A = K rows from LDS (ds_read_b128 along d), B = Q registers
//   O^T[d][query]   = V^T P^T    A = V[key][d] from LDS (b32, lanes = consecutive d),
//                                B = the P^T accumulator registers as they are
// In the 32x32 C layout lane (j = lane & 31, h = lane >> 5) holds column j and rows
// (r & 3) + 8 (r >> 2) + 4 h; an MFMA K-step consumes k = h, so feeding accumulator
// register r straight back as a B operand contracts over exactly that lane's rows if
// the A operand is read at the same row index — no shuffles, no LDS round trip for P.
// Row-wise softmax statistics are per-lane scalars plus one exchange with lane ^ 32.
// A workgroup = 4 waves = 128 queries (keys) of one (batch, head); the other side is
// streamed through LDS in blocks of 128 rows; any L, head_dim in {8, 16, 32, 64}.
#include <math.h>
#include "tdm_common.h"
#include "tdm_transformer.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int RB = 128;   // rows per LDS block

template <int HD> struct AttnCfg {
    static constexpr int KP = HD + 4;              // row pitch (floats): odd number of 16-B slots -> b128 reads conflict-free
    static constexpr int NT = HD > 32 ? 2 : 1;     // 32-wide d tiles of the transposed outputs
    static constexpr int NB = HD / 2;              // B-operand registers per lane for a full contraction over d
    static constexpr int TILE = RB * KP + 32;      // floats per staged block (+ slack: b32 reads of d >= HD stay in bounds)
};

// stage rows [r0, r0+128) of a [rows][HD] slice (row stride ld) into dst[128][KP]; rows >= nrows -> 0
template <int HD>
__device__ __forceinline__ void stage_block(float* dst, const float* __restrict__ src, long ld, int r0, int nrows, int tid) {
    constexpr int KP = AttnCfg<HD>::KP;
    for (int e = tid; e < RB * (HD / 4); e += 256) {
        const int rr = e / (HD / 4), d4 = e - rr * (HD / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + rr < nrows) v = *reinterpret_cast<const float4*>(src + (long)(r0 + rr) * ld + d4 * 4);
        *reinterpret_cast<float4*>(dst + rr * KP + d4 * 4) = v;
    }
}

// B-operand registers of one row: reg[4*c8 + jj] = row[8*c8 + 4*h + jj]
template <int HD>
__device__ __forceinline__ void load_breg(float (&reg)[HD / 2], const float* __restrict__ row, int h, bool valid) {
#pragma unroll
    for (int c8 = 0; c8 < HD / 8; ++c8) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) v = *reinterpret_cast<const float4*>(row + 8 * c8 + 4 * h);
        reg[4 * c8 + 0] = v.x; reg[4 * c8 + 1] = v.y; reg[4 * c8 + 2] = v.z; reg[4 * c8 + 3] = v.w;
    }
}

// acc[row = LDS row c32 + j][col = lane's own row] = sum_d T[c32 + j][d] * reg[d]
template <int HD>
__device__ __forceinline__ f32x16 rows_dot_reg(const float* T, int c32, int j, int h, const float (&reg)[HD / 2]) {
    constexpr int KP = AttnCfg<HD>::KP;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int c8 = 0; c8 < HD / 8; ++c8) {
        const float4 a = *reinterpret_cast<const float4*>(T + (c32 + j) * KP + 8 * c8 + 4 * h);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, reg[4 * c8 + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, reg[4 * c8 + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, reg[4 * c8 + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, reg[4 * c8 + 3], acc, 0, 0, 0);
    }
    return acc;
}

// out[t][d = 32 t + j][col] += sum_{rows of chunk c32} T[row][32 t + j] * w[row][col], w = accumulator-layout registers
template <int HD>
__device__ __forceinline__ void accum_T_times(f32x16 (&out)[AttnCfg<HD>::NT], const float* T, int c32, int j, int h,
                                              const f32x16& w) {
    constexpr int KP = AttnCfg<HD>::KP;
#pragma unroll
    for (int t = 0; t < AttnCfg<HD>::NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = c32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            out[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(T[row * KP + 32 * t + j], w[r], out[t], 0, 0, 0);
        }
}

// write the transposed accumulator tiles as rows: dst_row[32 t + 8 q + 4 h + (0..3)] = out[t][4 q + ..] * mul
template <int HD>
__device__ __forceinline__ void store_rows(float* __restrict__ dst_row, const f32x16 (&out)[AttnCfg<HD>::NT], int h, float mul) {
#pragma unroll
    for (int t = 0; t < AttnCfg<HD>::NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int d = 32 * t + 8 * q + 4 * h;
            if (d < HD)
                *reinterpret_cast<float4*>(dst_row + d) = make_float4(out[t][4 * q] * mul, out[t][4 * q + 1] * mul,
                                                                      out[t][4 * q + 2] * mul, out[t][4 * q + 3] * mul);
        }
}

// ------------------------------------------------------------------ forward
template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                            float* __restrict__ lse, int L, int D, int H, float scale,
                                                            DropArgs dr) {
    using C = AttnCfg<HD>;
    extern __shared__ float4 sm4[];
    float* Ks = reinterpret_cast<float*>(sm4);
    float* Vs = Ks + C::TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / H, hh = bh - b * H;
    const int qw0 = blockIdx.y * RB + wave * 32;
    const int qi = qw0 + j;
    const bool qvalid = qi < L;
    const float* base = qkv + (long)b * L * 3 * D + hh * HD;

    float qreg[C::NB];
    load_breg<HD>(qreg, base + (long)qi * 3 * D, h, qvalid);
    f32x16 acc_o[C::NT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[t][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    for (int k0 = 0; k0 < L; k0 += RB) {
        __syncthreads();
        stage_block<HD>(Ks, base + D, 3L * D, k0, L, tid);
        stage_block<HD>(Vs, base + 2 * D, 3L * D, k0, L, tid);
        __syncthreads();
        if (qw0 >= L) continue;   // wave-uniform: this wave has no query rows
        const int nchunk = min(RB / 32, (L - k0 + 31) / 32);
        for (int c = 0; c < nchunk; ++c) {
            f32x16 s = rows_dot_reg<HD>(Ks, c * 32, j, h, qreg);
            float mloc = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + c * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                s[r] = key < L ? s[r] * scale : -INFINITY;
                mloc = fmaxf(mloc, s[r]);
            }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
            const float m_new = fmaxf(m, mloc);
            const float corr = expf(m - m_new);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = expf(s[r] - m_new);
                psum += p;
                s[r] = p;
            }
            if (dr.thr != 0u) {
                const unsigned long long rowbase = ((unsigned long long)bh * L + qi) * L + k0 + c * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    s[r] = tdm_keep(dr, rowbase + (r & 3) + 8 * (r >> 2)) ? s[r] * dr.scale : 0.f;
            }
            psum += __shfl_xor(psum, 32);
            l = l * corr + psum;
            m = m_new;
#pragma unroll
            for (int t = 0; t < C::NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[t][r] *= corr;
            accum_T_times<HD>(acc_o, Vs, c * 32, j, h, s);
        }
    }
    if (qvalid) {
        store_rows<HD>(o + ((long)b * L + qi) * D + hh * HD, acc_o, h, 1.f / l);
        if (h == 0) lse[(long)bh * L + qi] = m + logf(l);
    }
}

// ------------------------------------------------------------------ backward, dQ (+ D_i = dO_i . O_i)
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dq_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                               const float* __restrict__ lse, const float* __restrict__ dO,
                                                               float* __restrict__ dqkv, float* __restrict__ Dvec, int L,
                                                               int D, int H, float scale, DropArgs dr) {
    using C = AttnCfg<HD>;
    extern __shared__ float4 sm4[];
    float* Ks = reinterpret_cast<float*>(sm4);
    float* Vs = Ks + C::TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / H, hh = bh - b * H;
    const int qw0 = blockIdx.y * RB + wave * 32;
    const int qi = qw0 + j;
    const bool qvalid = qi < L;
    const float* base = qkv + (long)b * L * 3 * D + hh * HD;

    float qreg[C::NB], doreg[C::NB];
    load_breg<HD>(qreg, base + (long)qi * 3 * D, h, qvalid);
    load_breg<HD>(doreg, dO + ((long)b * L + qi) * D + hh * HD, h, qvalid);
    float Di = 0.f;
    {
        float oreg[C::NB];
        load_breg<HD>(oreg, o + ((long)b * L + qi) * D + hh * HD, h, qvalid);
#pragma unroll
        for (int k = 0; k < C::NB; ++k) Di = fmaf(doreg[k], oreg[k], Di);
        Di += __shfl_xor(Di, 32);
    }
    const float lse_i = qvalid ? lse[(long)bh * L + qi] : 0.f;
    f32x16 acc_dq[C::NT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_dq[t][r] = 0.f;

    for (int k0 = 0; k0 < L; k0 += RB) {
        __syncthreads();
        stage_block<HD>(Ks, base + D, 3L * D, k0, L, tid);
        stage_block<HD>(Vs, base + 2 * D, 3L * D, k0, L, tid);
        __syncthreads();
        if (qw0 >= L) continue;
        const int nchunk = min(RB / 32, (L - k0 + 31) / 32);
        for (int c = 0; c < nchunk; ++c) {
            f32x16 s = rows_dot_reg<HD>(Ks, c * 32, j, h, qreg);
            f32x16 dp = rows_dot_reg<HD>(Vs, c * 32, j, h, doreg);
            const unsigned long long rowbase = ((unsigned long long)bh * L + qi) * L + k0 + c * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + c * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float p = key < L ? expf(s[r] * scale - lse_i) : 0.f;
                float dpv = dp[r];
                if (dr.thr != 0u) dpv = tdm_keep(dr, rowbase + (r & 3) + 8 * (r >> 2)) ? dpv * dr.scale : 0.f;
                s[r] = p * (dpv - Di);
            }
            accum_T_times<HD>(acc_dq, Ks, c * 32, j, h, s);
        }
    }
    if (qvalid) {
        store_rows<HD>(dqkv + ((long)b * L + qi) * 3 * D + hh * HD, acc_dq, h, scale);
        if (h == 0) Dvec[(long)bh * L + qi] = Di;
    }
}

// ------------------------------------------------------------------ backward, dK and dV
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ lse,
                                                                const float* __restrict__ dO, const float* __restrict__ Dvec,
                                                                float* __restrict__ dqkv, int L, int D, int H, float scale,
                                                                DropArgs dr) {
    using C = AttnCfg<HD>;
    extern __shared__ float4 sm4[];
    float* Qs = reinterpret_cast<float*>(sm4);
    float* Os = Qs + C::TILE;
    float* Ls = Os + C::TILE;   // [128] lse (+inf beyond L: exp(s - inf) = 0)
    float* Ds = Ls + RB;        // [128] D_i
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / H, hh = bh - b * H;
    const int kw0 = blockIdx.y * RB + wave * 32;
    const int kj = kw0 + j;
    const bool kvalid = kj < L;
    const float* base = qkv + (long)b * L * 3 * D + hh * HD;

    float kreg[C::NB], vreg[C::NB];
    load_breg<HD>(kreg, base + (long)kj * 3 * D + D, h, kvalid);
    load_breg<HD>(vreg, base + (long)kj * 3 * D + 2 * D, h, kvalid);
    f32x16 acc_dk[C::NT], acc_dv[C::NT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_dk[t][r] = 0.f; acc_dv[t][r] = 0.f; }

    for (int i0 = 0; i0 < L; i0 += RB) {
        __syncthreads();
        stage_block<HD>(Qs, base, 3L * D, i0, L, tid);
        stage_block<HD>(Os, dO + (long)b * L * D + hh * HD, (long)D, i0, L, tid);
        if (tid < RB) {
            const int ig = i0 + tid;
            Ls[tid] = ig < L ? lse[(long)bh * L + ig] : INFINITY;
            Ds[tid] = ig < L ? Dvec[(long)bh * L + ig] : 0.f;
        }
        __syncthreads();
        if (kw0 >= L) continue;
        const int nchunk = min(RB / 32, (L - i0 + 31) / 32);
        for (int c = 0; c < nchunk; ++c) {
            f32x16 s = rows_dot_reg<HD>(Qs, c * 32, j, h, kreg);     // S[query][key]
            f32x16 dp = rows_dot_reg<HD>(Os, c * 32, j, h, vreg);    // dP[query][key]
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 l4 = *reinterpret_cast<const float4*>(Ls + c * 32 + 8 * q + 4 * h);
                const float4 d4 = *reinterpret_cast<const float4*>(Ds + c * 32 + 8 * q + 4 * h);
                const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int r = 4 * q + jj;
                    const float p = expf(s[r] * scale - lq[jj]);
                    float pd = p, dpv = dp[r];
                    if (dr.thr != 0u) {
                        const int query = i0 + c * 32 + 8 * q + 4 * h + jj;
                        const bool keep = tdm_keep(dr, ((unsigned long long)bh * L + query) * L + kj);
                        pd = keep ? p * dr.scale : 0.f;
                        dpv = keep ? dpv * dr.scale : 0.f;
                    }
                    s[r] = pd;
                    dp[r] = p * (dpv - dq[jj]);
                }
            }
            accum_T_times<HD>(acc_dv, Os, c * 32, j, h, s);
            accum_T_times<HD>(acc_dk, Qs, c * 32, j, h, dp);
        }
    }
    if (kvalid) {
        float* dst = dqkv + ((long)b * L + kj) * 3 * D + hh * HD;
        store_rows<HD>(dst + D, acc_dk, h, scale);
        store_rows<HD>(dst + 2 * D, acc_dv, h, 1.f);
    }
}

template <int HD>
int attn_mfma_launch(int which, const float* qkv, const float* o, const float* lse, const float* dO, float* out, float* aux,
                     long B, int L, int D, int H, DropArgs dr, hipStream_t st) {
    using C = AttnCfg<HD>;
    const float scale = 1.0f / sqrtf((float)HD);
    dim3 grid((unsigned)(B * H), (L + RB - 1) / RB);
    const size_t lds = (size_t)(2 * C::TILE + 2 * RB) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {   // head_dim 64 needs ~71 KB of dynamic LDS
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_mfma_kernel<HD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_mfma_kernel<HD>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_mfma_kernel<HD>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            tdm_set_error("attention (mfma): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return 100 + (int)e;
        }
        attr_set = true;
    }
    if (which == 0) {
        hipLaunchKernelGGL((attn_fwd_mfma_kernel<HD>), grid, dim3(256), lds, st, qkv, out, aux, L, D, H, scale, dr);
        TDM_CHECK_LAUNCH("attn_fwd_mfma");
    } else if (which == 1) {
        hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<HD>), grid, dim3(256), lds, st, qkv, o, lse, dO, out, aux, L, D, H, scale, dr);
        TDM_CHECK_LAUNCH("attn_bwd_dq_mfma");
    } else {
        hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<HD>), grid, dim3(256), lds, st, qkv, lse, dO, aux, out, L, D, H, scale, dr);
        TDM_CHECK_LAUNCH("attn_bwd_dkv_mfma");
    }
    return 0;
}

}  // namespace

// which: 0 forward (out = O, aux = lse), 1 dQ (out = dqkv, aux = D written), 2 dK/dV (out = dqkv, aux = D read)
int tdm_launch_attn_mfma(int which, int hd, const float* qkv, const float* o, const float* lse, const float* dO, float* out,
                         float* aux, long B, int L, int D, int H, DropArgs dr, hipStream_t st) {
    TDM_REQUIRE((D % 4) == 0 && (((uintptr_t)qkv | (uintptr_t)out) & 15) == 0, "attention: 16-byte alignment");
    switch (hd) {
        case 8: return attn_mfma_launch<8>(which, qkv, o, lse, dO, out, aux, B, L, D, H, dr, st);
        case 16: return attn_mfma_launch<16>(which, qkv, o, lse, dO, out, aux, B, L, D, H, dr, st);
        case 32: return attn_mfma_launch<32>(which, qkv, o, lse, dO, out, aux, B, L, D, H, dr, st);
        case 64: return attn_mfma_launch<64>(which, qkv, o, lse, dO, out, aux, B, L, D, H, dr, st);
    }
    tdm_set_error("attention: head_dim %d not supported (8, 16, 32, 64)", hd);
    return 1;
}
