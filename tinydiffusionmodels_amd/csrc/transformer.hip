// TinyTransformer embedding-space denoiser (src/shakespeare.py:105-120):
//   x + Linear(1,D)(t/1000)  ->  depth x post-LN nn.TransformerEncoderLayer
//   (packed in_proj, n_heads-way softmax attention without mask, out_proj,
//    residual + LayerNorm, ReLU FFN, residual + LayerNorm); eval, or train mode
//   with the 1 + 4*depth dropout sites of the reference (masks: tdm_dropout.h).
// Forward, backward and the text reverse step, as launches on one stream.
// Linear layers and their gradients run on the MFMA GEMMs (gemm_bf16.hip /
// gemm_mfma.hip); attention on the fp32 matrix cores (attn_mfma.hip; the scalar
// fp32 kernels below are kept as attention mode 0, the cross-check); LayerNorm is
// one wavefront per token row.
#include <math.h>
#include "tdm_common.h"
#include <cstdlib>
#include "tdm_transformer.h"
#include "tdm_s16.h"

namespace {

// arithmetic of the linear layers: 0 exact fp32 MFMA, 1 bf16x3 split operands (default, meets the 1e-3
// parity bound), 2 plain bf16 operands (throughput mode)
#define g_gemm_mode (tdm_cur_ctx().gemm_mode)   // a field of the calling thread's current context (threading)
// attention: 0 scalar fp32 kernels (this file), 1 fp32 MFMA (attn_mfma.hip), 2 bf16x3 MFMA (attn_bf16.hip, default)
#define g_attn_mode (tdm_cur_ctx().attn_mode)

// train-mode dropout of one call: p = 0 -> off
struct Drop {
    float p; uint64_t seed;
    const uint32_t* salt = nullptr;   // device word XORed into every site key (DropArgs::salt), or nullptr
    DropArgs site(int s) const { DropArgs d = tdm_drop_site(p, seed, s); if (d.thr != 0u) d.salt = salt; return d; }
};

constexpr int LN_SLABS = 1024;  // workgroup partials of the LayerNorm affine / bias gradients (four 4-wave workgroups per CU: the row loop has no prefetch, occupancy hides its latency)
constexpr int CS_SLABS = 256;  // row-block partials of the bias gradients

// Weight-gradient GEMMs contract over tokens (K = B*L, tens of thousands) into small [N][K] outputs:
// each tensor gets its own split-K factor so that tiles x splits ~ 512 workgroups (fewer partial slabs to write and re-read;
// TDM_TN_WGS overrides the target for sweeps).
inline int wgrad_splitk(int N, int K) {
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
    static const int target = getenv("TDM_TN_WGS") ? atoi(getenv("TDM_TN_WGS")) : 512;   // sweep (8-wave NT kernel era): 384 5.65 ms, 512 5.50, 768 5.57, 1024 5.59, 1536 5.68 per step
    int sk = (target + tiles - 1) / tiles;
    if (sk < 1) sk = 1;
    if (sk > 128) sk = 128;
    if (sk >= 8) sk = (sk + 4) / 8 * 8;   // whole splits per XCD (gemm_tn_bf16_kernel's workgroup map)
    return sk;
}
// slab regions (floats) of the 4 weight matrices of one layer, in order in_w, out_w, l1_w, l2_w
struct SlabPlan { long base[8][4]; long bias_base[8][4]; long ln_base[8][2]; int sk[4]; long len[4]; int nout[4]; long total; };
// Weight gradients as PAIRS on the 256 x 256-tile ring kernel (gemm_tn_ring.hip): {in_proj, out_proj} and {linear1, linear2} are
// one launch each, and the pair shares a split count chosen so that its tiles x splits ~ one workgroup per CU.
inline bool wgrad_pairs_shape(int D, int F) {
    static const bool off = getenv("TDM_TN_RING") && atoi(getenv("TDM_TN_RING")) == 0;   // A/B timing
    return !off && D >= 256 && F >= 256 && (D % 16) == 0 && (F % 16) == 0;
}
inline int quad_splitk(int D, int F) {   // all four products of a layer in one launch: tiles x splits ~ one workgroup per CU
    const int d = (D + 255) / 256, f = (F + 255) / 256, d3 = (3 * D + 255) / 256;
    const int tiles = d3 * d + d * d + 2 * f * d;
    const int sk = 256 / tiles;
    return sk < 1 ? 1 : (sk > 128 ? 128 : sk);
}
// pairs: 0 = one launch per product (128 x 128 tiles), 1 = two launches per layer (side-queue overlap), 2 = one launch per layer.
// 1 and 2 cut the tokens into the SAME splits, so the two issue modes (and a graph captured on one queue) give the same bits.
SlabPlan slab_plan(int D, int depth, int F, int pairs) {
    SlabPlan p{};
    const int Ns[4] = {3 * D, D, F, D}, Ks[4] = {D, D, D, F};
    long off = 0;
    for (int k = 0; k < 4; ++k) { p.sk[k] = wgrad_splitk(Ns[k], Ks[k]); p.len[k] = (long)Ns[k] * Ks[k]; p.nout[k] = Ns[k]; }
    if (pairs != 0) p.sk[0] = p.sk[1] = p.sk[2] = p.sk[3] = quad_splitk(D, F);
    for (int l = 0; l < depth; ++l)
        for (int k = 0; k < 4; ++k) { p.base[l][k] = off; off += p.sk[k] * p.len[k]; }
    // per-split partial bias gradients written by the weight-gradient GEMMs (bf16 modes)
    for (int l = 0; l < depth; ++l)
        for (int k = 0; k < 4; ++k) { p.bias_base[l][k] = off; off += (long)p.sk[k] * ((Ns[k] + 63) & ~63); }
    // workgroup partials of the two LayerNorm backward launches of every layer ([LN_SLABS][3 D]: d gamma | d beta | column sums):
    // summed by the backward's ONE slab-reduction launch instead of a reduction launch per LayerNorm
    for (int l = 0; l < depth; ++l)
        for (int k = 0; k < 2; ++k) { p.ln_base[l][k] = off; off += (long)LN_SLABS * 3 * D; }
    p.total = off;
    return p;
}

// ----------------------------- parameter layout ---------------------------------
struct LayerOff { long in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w, n2_b; };
struct TTLayout {
    LayerOff L[8];
    long te_w, te_b, total;
    int ntensor;
    long tensor_off[8 * 12 + 3];
};
TTLayout tt_layout(int D, int depth, int F) {
    TTLayout t{};
    long off = 0;
    int ti = 0;
    auto take = [&](long n) { long o = off; t.tensor_off[ti++] = off; off += n; return o; };
    for (int l = 0; l < depth; ++l) {
        LayerOff& o = t.L[l];
        o.in_w = take(3L * D * D); o.in_b = take(3L * D);
        o.out_w = take((long)D * D); o.out_b = take(D);
        o.l1_w = take((long)F * D); o.l1_b = take(F);
        o.l2_w = take((long)D * F); o.l2_b = take(D);
        o.n1_w = take(D); o.n1_b = take(D); o.n2_w = take(D); o.n2_b = take(D);
    }
    t.te_w = take(D); t.te_b = take(D);
    t.tensor_off[ti] = off;
    t.ntensor = ti;
    t.total = off;
    return t;
}

// --------------------------------- workspace -------------------------------------
struct LayerWs {
    float *hin, *qkv, *o, *lse, *s1, *mean1, *rstd1, *h1, *f1, *s2, *mean2, *rstd2;
    float *hin16, *o16, *h1_16;   // S16 twins of the GEMM operands (bf16 GEMM modes; f1 itself is S16 there)
    unsigned* fmask;              // sign masks of the FFN hidden activation (fused chain, ffn_chain.hip)
};
struct TTWs {
    float *that, *tb, *abuf, *wT, *wT2;
    LayerWs L[8];
    // backward temporaries
    float *g_h, *g_s, *g_s1, *g_d, *g_f, *g_h1, *g_o, *g_qkv, *Dvec, *Sb, *part;
    float *P16, *g16, *g_qkv16;   // S16: the parameter vector, the D-wide gradient operand of the current GEMM pair, d(qkv)
    float* wT_all;                // [depth][in_w^T | out_w^T | l1_w^T | l2_w^T] S16 (backward), or nullptr
    float* ypart;                 // partial sums of a hidden-range-segmented FFN chain (small batches; ffn_chain.hip), or nullptr
    long total;
};
TTWs tt_carve(float* base, long B, int Lq, int D, int H, int depth, int F, int training) {
    TTWs w{};
    long off = 0;
    auto take = [&](long n) {
        float* p = base ? base + off : nullptr;
        off += (n + 63) & ~63L;
        return p;
    };
    const long M = B * Lq;
    w.that = take(B); w.tb = take(B * D); w.abuf = take(M * D);
    w.wT = take((long)(F > 3 * D ? F : 3 * D) * D);   // transposed weight of the current data-gradient GEMM
    w.wT2 = take((long)F * D);                        // second transposed weight of the fused FFN data gradient
    // all four weight matrices of every layer, transposed (S16), written by ONE launch at the start of a backward pass
    w.wT_all = (training && depth <= 8) ? take((long)depth * (4L * D * D + 2L * F * D)) : nullptr;
    for (int l = 0; l < depth; ++l) {
        LayerWs& x = w.L[l];
        x.hin = take(M * D); x.qkv = take(M * 3 * D); x.o = take(M * D); x.lse = take(B * H * Lq);
        x.s1 = take(M * D); x.mean1 = take(M); x.rstd1 = take(M); x.h1 = take(M * D); x.f1 = take(M * F);
        x.s2 = take(M * D); x.mean2 = take(M); x.rstd2 = take(M);
        x.hin16 = take(M * D); x.o16 = take(M * D); x.h1_16 = take(M * D);
        x.fmask = reinterpret_cast<unsigned*>(take(tdm_ffn_chain_mask_elems(M, F)));
    }
    w.P16 = take(tt_layout(D, depth, F).total);
    {
        const long np = (D == 256 && (F % 32) == 0 && F >= 64 && F <= 2048) ? tdm_ffn_chain_part_floats(M, F) : 0;
        w.ypart = np > 0 ? take(np) : nullptr;
    }
    if (training) {
        w.g_h = take(M * D); w.g_s = take(M * D); w.g_s1 = take(M * D); w.g_d = take(M * D); w.g_f = take(M * F); w.g_h1 = take(M * D);
        w.g_o = take(M * D); w.g_qkv = take(M * 3 * D); w.Dvec = take(B * H * Lq); w.Sb = take(B * D);
        long pmax = (long)LN_SLABS * 3 * D;
        const long c1 = (long)CS_SLABS * (F > 3 * D ? F : 3 * D);
        if (c1 > pmax) pmax = c1;
        w.part = take(pmax + (F > 3 * D ? F : 3 * D));
        w.g16 = take(M * D); w.g_qkv16 = take(M * 3 * D);
    }
    w.total = off;
    return w;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// h0[m][d] = x[m][d] + (w[d]*t[b]/1000 + bias[d])      (src/shakespeare.py:116-118); h16 = S16 twin of h0 (or nullptr)
__global__ __launch_bounds__(256) void add_timebias_kernel(const float* __restrict__ x, const int64_t* __restrict__ t,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ that, float* __restrict__ tb,
                                                           float* __restrict__ h0, float* __restrict__ h16, long B, int L, int D,
                                                           DropArgs dr) {
    const long total4 = B * L * D / 4;
    for (long i4 = (long)blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += (long)gridDim.x * 256) {
        const long i = i4 * 4;
        const int d = (int)(i % D);
        const long m = i / D;
        const long b = m / L;
        const float th = __fdiv_rn((float)t[b], 1000.f);
        const float4 w4 = *reinterpret_cast<const float4*>(w + d), b4 = *reinterpret_cast<const float4*>(bias + d);
        const float4 tv = make_float4(fmaf(w4.x, th, b4.x), fmaf(w4.y, th, b4.y), fmaf(w4.z, th, b4.z), fmaf(w4.w, th, b4.w));
        const float4 xv = *reinterpret_cast<const float4*>(x + i);
        float4 v = make_float4(xv.x + tv.x, xv.y + tv.y, xv.z + tv.z, xv.w + tv.w);
        if (dr.thr != 0u) {
            v.x = tdm_keep(dr, (unsigned long long)i) ? v.x * dr.scale : 0.f;
            v.y = tdm_keep(dr, (unsigned long long)i + 1) ? v.y * dr.scale : 0.f;
            v.z = tdm_keep(dr, (unsigned long long)i + 2) ? v.z * dr.scale : 0.f;
            v.w = tdm_keep(dr, (unsigned long long)i + 3) ? v.w * dr.scale : 0.f;
        }
        *reinterpret_cast<float4*>(h0 + i) = v;
        if (h16 != nullptr) tdm_store_s16_4(h16, m, D, d, v);
        if (m - b * L == 0) {
            *reinterpret_cast<float4*>(tb + b * D + d) = tv;
            if (d == 0) that[b] = th;
        }
    }
}

// ------------------------------- attention ---------------------------------------
// one thread per query row, 128 rows per block, keys/values streamed through LDS
// in chunks of 128 (all lanes read the same K/V row: LDS broadcast).
template <int HD>
__global__ __launch_bounds__(128) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                       float* __restrict__ lse, int L, int D, int H, float scale,
                                                       DropArgs dr) {
    extern __shared__ float4 sm4[];
    float* Ks = reinterpret_cast<float*>(sm4);
    float* Vs = Ks + 128 * HD;
    const int tid = threadIdx.x;
    const int bh = blockIdx.x, b = bh / H, hh = bh - b * H;
    const int i = blockIdx.y * 128 + tid;
    const bool valid = i < L;
    float q[HD], acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { q[d] = 0.f; acc[d] = 0.f; }
    if (valid) {
        const float* src = qkv + ((long)(b * L + i)) * 3 * D + hh * HD;
#pragma unroll
        for (int d4 = 0; d4 < HD / 4; ++d4) {
            const float4 v = *reinterpret_cast<const float4*>(src + d4 * 4);
            q[d4 * 4 + 0] = v.x; q[d4 * 4 + 1] = v.y; q[d4 * 4 + 2] = v.z; q[d4 * 4 + 3] = v.w;
        }
    }
    float m = -INFINITY, l = 0.f;
    for (int j0 = 0; j0 < L; j0 += 128) {
        __syncthreads();
        for (int e = tid; e < 128 * (HD / 4); e += 128) {
            const int jj = e / (HD / 4), d4 = e - jj * (HD / 4);
            const int jg = j0 + jj;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (jg < L) {
                const float* base = qkv + ((long)(b * L + jg)) * 3 * D + hh * HD + d4 * 4;
                kv = *reinterpret_cast<const float4*>(base + D);
                vv = *reinterpret_cast<const float4*>(base + 2 * D);
            }
            *reinterpret_cast<float4*>(Ks + jj * HD + d4 * 4) = kv;
            *reinterpret_cast<float4*>(Vs + jj * HD + d4 * 4) = vv;
        }
        __syncthreads();
        const int jmax = min(128, L - j0);
        for (int jj = 0; jj < jmax; ++jj) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) s = fmaf(q[d], Ks[jj * HD + d], s);
            s *= scale;
            if (s > m) {
                const float corr = expf(m - s);
                l *= corr;
#pragma unroll
                for (int d = 0; d < HD; ++d) acc[d] *= corr;
                m = s;
            }
            const float p = expf(s - m);
            l += p;
            float pd = p;
            if (dr.thr != 0u) pd = tdm_keep(dr, ((unsigned long long)bh * L + i) * L + j0 + jj) ? p * dr.scale : 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = fmaf(pd, Vs[jj * HD + d], acc[d]);
        }
    }
    if (valid) {
        const float inv = 1.f / l;
        float* dst = o + ((long)(b * L + i)) * D + hh * HD;
#pragma unroll
        for (int d4 = 0; d4 < HD / 4; ++d4)
            *reinterpret_cast<float4*>(dst + d4 * 4) =
                make_float4(acc[d4 * 4] * inv, acc[d4 * 4 + 1] * inv, acc[d4 * 4 + 2] * inv, acc[d4 * 4 + 3] * inv);
        lse[(long)bh * L + i] = m + logf(l);
    }
}

// backward pass 1: dQ and D_i = dO_i . O_i   (thread per query row)
template <int HD>
__global__ __launch_bounds__(128) void attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                          const float* __restrict__ lse, const float* __restrict__ dO,
                                                          float* __restrict__ dqkv, float* __restrict__ Dvec, int L,
                                                          int D, int H, float scale, DropArgs dr) {
    extern __shared__ float4 sm4[];
    float* Ks = reinterpret_cast<float*>(sm4);
    float* Vs = Ks + 128 * HD;
    const int tid = threadIdx.x;
    const int bh = blockIdx.x, b = bh / H, hh = bh - b * H;
    const int i = blockIdx.y * 128 + tid;
    const bool valid = i < L;
    float q[HD], dov[HD], dq[HD];
    float Di = 0.f, lse_i = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) { q[d] = 0.f; dov[d] = 0.f; dq[d] = 0.f; }
    if (valid) {
        const float* qs = qkv + ((long)(b * L + i)) * 3 * D + hh * HD;
        const float* ds = dO + ((long)(b * L + i)) * D + hh * HD;
        const float* os = o + ((long)(b * L + i)) * D + hh * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            q[d] = qs[d];
            dov[d] = ds[d];
            Di = fmaf(dov[d], os[d], Di);
        }
        lse_i = lse[(long)bh * L + i];
    }
    for (int j0 = 0; j0 < L; j0 += 128) {
        __syncthreads();
        for (int e = tid; e < 128 * (HD / 4); e += 128) {
            const int jj = e / (HD / 4), d4 = e - jj * (HD / 4);
            const int jg = j0 + jj;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (jg < L) {
                const float* base = qkv + ((long)(b * L + jg)) * 3 * D + hh * HD + d4 * 4;
                kv = *reinterpret_cast<const float4*>(base + D);
                vv = *reinterpret_cast<const float4*>(base + 2 * D);
            }
            *reinterpret_cast<float4*>(Ks + jj * HD + d4 * 4) = kv;
            *reinterpret_cast<float4*>(Vs + jj * HD + d4 * 4) = vv;
        }
        __syncthreads();
        const int jmax = min(128, L - j0);
        for (int jj = 0; jj < jmax; ++jj) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                s = fmaf(q[d], Ks[jj * HD + d], s);
                dp = fmaf(dov[d], Vs[jj * HD + d], dp);
            }
            const float p = expf(s * scale - lse_i);
            if (dr.thr != 0u) dp = tdm_keep(dr, ((unsigned long long)bh * L + i) * L + j0 + jj) ? dp * dr.scale : 0.f;
            const float dsv = p * (dp - Di);
#pragma unroll
            for (int d = 0; d < HD; ++d) dq[d] = fmaf(dsv, Ks[jj * HD + d], dq[d]);
        }
    }
    if (valid) {
        float* dst = dqkv + ((long)(b * L + i)) * 3 * D + hh * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) dst[d] = dq[d] * scale;
        Dvec[(long)bh * L + i] = Di;
    }
}

// backward pass 2: dK and dV   (thread per key row; Q, dO, lse, D streamed through LDS)
template <int HD>
__global__ __launch_bounds__(128) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ lse,
                                                           const float* __restrict__ dO, const float* __restrict__ Dvec,
                                                           float* __restrict__ dqkv, int L, int D, int H, float scale,
                                                           DropArgs dr) {
    extern __shared__ float4 sm4[];
    float* Qs = reinterpret_cast<float*>(sm4);
    float* Os = Qs + 128 * HD;
    float* Ls = Os + 128 * HD;   // 128 lse
    float* Ds = Ls + 128;        // 128 D
    const int tid = threadIdx.x;
    const int bh = blockIdx.x, b = bh / H, hh = bh - b * H;
    const int jg = blockIdx.y * 128 + tid;
    const bool valid = jg < L;
    float k[HD], v[HD], dk[HD], dv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { k[d] = 0.f; v[d] = 0.f; dk[d] = 0.f; dv[d] = 0.f; }
    if (valid) {
        const float* base = qkv + ((long)(b * L + jg)) * 3 * D + hh * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) { k[d] = base[D + d]; v[d] = base[2 * D + d]; }
    }
    for (int i0 = 0; i0 < L; i0 += 128) {
        __syncthreads();
        for (int e = tid; e < 128 * (HD / 4); e += 128) {
            const int ii = e / (HD / 4), d4 = e - ii * (HD / 4);
            const int ig = i0 + ii;
            float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), ov = qv;
            if (ig < L) {
                qv = *reinterpret_cast<const float4*>(qkv + ((long)(b * L + ig)) * 3 * D + hh * HD + d4 * 4);
                ov = *reinterpret_cast<const float4*>(dO + ((long)(b * L + ig)) * D + hh * HD + d4 * 4);
            }
            *reinterpret_cast<float4*>(Qs + ii * HD + d4 * 4) = qv;
            *reinterpret_cast<float4*>(Os + ii * HD + d4 * 4) = ov;
        }
        {
            const int ig = i0 + tid;
            Ls[tid] = (ig < L) ? lse[(long)bh * L + ig] : INFINITY;   // exp(s - inf) = 0 for padded rows
            Ds[tid] = (ig < L) ? Dvec[(long)bh * L + ig] : 0.f;
        }
        __syncthreads();
        const int imax = min(128, L - i0);
        for (int ii = 0; ii < imax; ++ii) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                s = fmaf(Qs[ii * HD + d], k[d], s);
                dp = fmaf(Os[ii * HD + d], v[d], dp);
            }
            const float p = expf(s * scale - Ls[ii]);
            float pd = p;
            if (dr.thr != 0u) {
                const bool keep = tdm_keep(dr, ((unsigned long long)bh * L + i0 + ii) * L + jg);
                pd = keep ? p * dr.scale : 0.f;
                dp = keep ? dp * dr.scale : 0.f;
            }
            const float dsv = p * (dp - Ds[ii]);
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                dv[d] = fmaf(pd, Os[ii * HD + d], dv[d]);
                dk[d] = fmaf(dsv, Qs[ii * HD + d], dk[d]);
            }
        }
    }
    if (valid) {
        float* dst = dqkv + ((long)(b * L + jg)) * 3 * D + hh * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) { dst[D + d] = dk[d] * scale; dst[2 * D + d] = dv[d]; }
    }
}

// ------------------------------- LayerNorm ---------------------------------------
// y = LN(x + r) * gamma + beta, one wavefront per row, row cached in registers (D <= 1024)
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ y, float* __restrict__ y16,
                                                     float* __restrict__ s_out, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, long M, int D, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= M) return;
    const int D4 = D >> 2;
    float4 buf[4];
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c4 = lane + 64 * q;
        buf[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 < D4) {
            const float4 a = reinterpret_cast<const float4*>(x + row * D)[c4];
            const float4 b = r != nullptr ? reinterpret_cast<const float4*>(r + row * D)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
            buf[q] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
            sum += (buf[q].x + buf[q].y) + (buf[q].z + buf[q].w);
        }
    }
    const float mean = wave_sum(sum) / (float)D;
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c4 = lane + 64 * q;
        if (c4 < D4) {
            const float dx = buf[q].x - mean, dy = buf[q].y - mean, dz = buf[q].z - mean, dw = buf[q].w - mean;
            var += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)D + eps);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c4 = lane + 64 * q;
        if (c4 < D4) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[c4];
            const float4 be = reinterpret_cast<const float4*>(beta)[c4];
            float4 o;
            o.x = (buf[q].x - mean) * rstd * g.x + be.x;
            o.y = (buf[q].y - mean) * rstd * g.y + be.y;
            o.z = (buf[q].z - mean) * rstd * g.z + be.z;
            o.w = (buf[q].w - mean) * rstd * g.w + be.w;
            reinterpret_cast<float4*>(y + row * D)[c4] = o;
            if (y16 != nullptr) tdm_store_s16_4(y16, row, D, c4 * 4, o);
            if (s_out != nullptr) reinterpret_cast<float4*>(s_out + row * D)[c4] = buf[q];
        }
    }
    if (lane == 0 && mean_out != nullptr) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// ds = rstd * (g - mean(g) - xhat*mean(g*xhat)), g = (dy + dy2) * gamma.
// s = x + a is the LayerNorm input, so ds is the gradient of both the residual x and
// the sub-layer output a; in train mode a = dropout(linear(...)) and the gradient the
// linear layer sees is ds_drop = mask * ds / (1 - p), written next to ds.
// Per-workgroup partials -> part[block][3][D]: dgamma = sum dy*xhat, dbeta = sum dy and
// the column sums of ds_drop (= the bias gradient of that linear layer).
// NQ float4 per lane cover a row (D <= 256*NQ); a wave keeps 4/NQ rows in flight.
template <int NQ>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dy2,
                                                     const float* __restrict__ s, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     float* __restrict__ ds, float* __restrict__ ds_drop,
                                                     float* __restrict__ op16, float* __restrict__ part, long M, int D,
                                                     DropArgs dr) {
    constexpr int R = 4 / NQ;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D4 = D >> 2;
    const float invD = 1.f / (float)D;
    float4 gacc[NQ], bacc[NQ], cacc[NQ], gm[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        gacc[q] = make_float4(0.f, 0.f, 0.f, 0.f); bacc[q] = gacc[q]; cacc[q] = gacc[q];
        const int c4 = lane + 64 * q;
        gm[q] = c4 < D4 ? reinterpret_cast<const float4*>(gamma)[c4] : gacc[q];
    }
    const long wid = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
    for (long row0 = wid * R; row0 < M; row0 += nw * R) {
        float4 d[R][NQ], xh[R][NQ];
        float mu[R], rs[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const long row = row0 + rr;
            const bool rv = row < M;
            mu[rr] = rv ? mean[row] : 0.f;
            rs[rr] = rv ? rstd[row] : 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c4 = lane + 64 * q;
                d[rr][q] = make_float4(0.f, 0.f, 0.f, 0.f);
                xh[rr][q] = d[rr][q];
                if (rv && c4 < D4) {
                    d[rr][q] = reinterpret_cast<const float4*>(dy + row * D)[c4];
                    if (dy2 != nullptr) {
                        const float4 d2 = reinterpret_cast<const float4*>(dy2 + row * D)[c4];
                        d[rr][q].x += d2.x; d[rr][q].y += d2.y; d[rr][q].z += d2.z; d[rr][q].w += d2.w;
                    }
                    xh[rr][q] = reinterpret_cast<const float4*>(s + row * D)[c4];
                }
            }
        }
        float c1[R], c2[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            c1[rr] = 0.f; c2[rr] = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c4 = lane + 64 * q;
                if (c4 < D4) {
                    const float4 sv = xh[rr][q], dd = d[rr][q];
                    const float4 x = make_float4((sv.x - mu[rr]) * rs[rr], (sv.y - mu[rr]) * rs[rr], (sv.z - mu[rr]) * rs[rr],
                                                 (sv.w - mu[rr]) * rs[rr]);
                    xh[rr][q] = x;
                    const float4 g = make_float4(dd.x * gm[q].x, dd.y * gm[q].y, dd.z * gm[q].z, dd.w * gm[q].w);
                    c1[rr] += (g.x + g.y) + (g.z + g.w);
                    c2[rr] += (g.x * x.x + g.y * x.y) + (g.z * x.z + g.w * x.w);
                    gacc[q].x += dd.x * x.x; gacc[q].y += dd.y * x.y; gacc[q].z += dd.z * x.z; gacc[q].w += dd.w * x.w;
                    bacc[q].x += dd.x; bacc[q].y += dd.y; bacc[q].z += dd.z; bacc[q].w += dd.w;
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < R; ++rr) { c1[rr] = wave_sum(c1[rr]) * invD; c2[rr] = wave_sum(c2[rr]) * invD; }
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const long row = row0 + rr;
            if (row >= M) continue;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c4 = lane + 64 * q;
                if (c4 < D4) {
                    const float4 dd = d[rr][q], x = xh[rr][q];
                    float4 o;
                    o.x = rs[rr] * (dd.x * gm[q].x - c1[rr] - x.x * c2[rr]);
                    o.y = rs[rr] * (dd.y * gm[q].y - c1[rr] - x.y * c2[rr]);
                    o.z = rs[rr] * (dd.z * gm[q].z - c1[rr] - x.z * c2[rr]);
                    o.w = rs[rr] * (dd.w * gm[q].w - c1[rr] - x.w * c2[rr]);
                    reinterpret_cast<float4*>(ds + row * D)[c4] = o;
                    if (dr.thr != 0u) {
                        const unsigned long long e = (unsigned long long)row * D + c4 * 4;
                        o.x = tdm_keep(dr, e) ? o.x * dr.scale : 0.f;
                        o.y = tdm_keep(dr, e + 1) ? o.y * dr.scale : 0.f;
                        o.z = tdm_keep(dr, e + 2) ? o.z * dr.scale : 0.f;
                        o.w = tdm_keep(dr, e + 3) ? o.w * dr.scale : 0.f;
                        if (ds_drop != nullptr) reinterpret_cast<float4*>(ds_drop + row * D)[c4] = o;
                    }
                    if (op16 != nullptr) tdm_store_s16_4(op16, row, D, c4 * 4, o);   // the linear layer's gradient operand
                    cacc[q].x += o.x; cacc[q].y += o.y; cacc[q].z += o.z; cacc[q].w += o.w;
                }
            }
        }
    }
    // the 4 waves' partial sums are added in fixed order through LDS: one partial per workgroup
    __shared__ float4 red[3][3][64 * NQ];
    float* dst = part + (long)blockIdx.x * 3 * D;
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            red[wave - 1][0][lane + 64 * q] = gacc[q];
            red[wave - 1][1][lane + 64 * q] = bacc[q];
            red[wave - 1][2][lane + 64 * q] = cacc[q];
        }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c4 = lane + 64 * q;
            if (c4 < D4) {
                float4 a = gacc[q], b = bacc[q], c = cacc[q];
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const float4 x = red[w][0][c4], y = red[w][1][c4], z = red[w][2][c4];
                    a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
                    b.x += y.x; b.y += y.y; b.z += y.z; b.w += y.w;
                    c.x += z.x; c.y += z.y; c.z += z.z; c.w += z.w;
                }
                reinterpret_cast<float4*>(dst)[c4] = a;
                reinterpret_cast<float4*>(dst + D)[c4] = b;
                reinterpret_cast<float4*>(dst + 2 * D)[c4] = c;
            }
        }
    }
}

// in-place dropout backward: g = mask * g / (1 - p)
__global__ __launch_bounds__(256) void drop_apply_kernel(float* __restrict__ g, long n4, DropArgs dr) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 v = reinterpret_cast<float4*>(g)[i];
        const unsigned long long e = (unsigned long long)i * 4;
        v.x = tdm_keep(dr, e) ? v.x * dr.scale : 0.f;
        v.y = tdm_keep(dr, e + 1) ? v.y * dr.scale : 0.f;
        v.z = tdm_keep(dr, e + 2) ? v.z * dr.scale : 0.f;
        v.w = tdm_keep(dr, e + 3) ? v.w * dr.scale : 0.f;
        reinterpret_cast<float4*>(g)[i] = v;
    }
}

// column sums over row blocks: part[blockIdx.x][n] = sum_{rows of block} a[row][n]   (N % 4 == 0, N <= 4096)
// 256 threads = (256 / C4) row lanes x C4 float4 columns (or 1 row x up to 4 column passes when C4 > 256)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, float* __restrict__ part, long M, int N) {
    __shared__ float4 sh[256];
    const int C4 = N >> 2;
    if (C4 <= 256) {
        const int rpp = 256 / C4;
        const int c4 = threadIdx.x % C4, rg = threadIdx.x / C4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rg < rpp) {
            for (long row = (long)blockIdx.x * rpp + rg; row < M; row += (long)gridDim.x * rpp) {
                const float4 v = reinterpret_cast<const float4*>(a + row * N)[c4];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        sh[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x < C4) {
            float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < rpp; ++k) {
                const float4 t = sh[k * C4 + threadIdx.x];
                sacc.x += t.x; sacc.y += t.y; sacc.z += t.z; sacc.w += t.w;
            }
            reinterpret_cast<float4*>(part + (long)blockIdx.x * N)[threadIdx.x] = sacc;
        }
    } else {
        float4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long row = blockIdx.x; row < M; row += gridDim.x) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c4 = threadIdx.x + 256 * q;
                if (c4 < C4) {
                    const float4 v = reinterpret_cast<const float4*>(a + row * N)[c4];
                    acc[q].x += v.x; acc[q].y += v.y; acc[q].z += v.z; acc[q].w += v.w;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c4 = threadIdx.x + 256 * q;
            if (c4 < C4) reinterpret_cast<float4*>(part + (long)blockIdx.x * N)[c4] = acc[q];
        }
    }
}

// Sb[b][d] = sum_l g[b][l][d]
// (one workgroup per sequence; D / 4 column quads x 256 / (D / 4) row groups, four independent row chains per thread, row groups
//  added in fixed order through LDS — one thread per column walking L dependent loads ran at 1 TB/s: 32 us for 34 MB)
__global__ __launch_bounds__(256) void seqsum_kernel(const float* __restrict__ g, float* __restrict__ Sb, int L, int D) {
    __shared__ float4 sh[256];
    const long b = blockIdx.x;
    const int D4 = D >> 2;
    if ((D & 3) != 0 || D4 > 256 || (256 % D4) != 0) {   // odd widths: the plain form
        for (int d = threadIdx.x; d < D; d += 256) {
            float acc = 0.f;
            for (int l = 0; l < L; ++l) acc += g[(b * L + l) * D + d];
            Sb[b * D + d] = acc;
        }
        return;
    }
    const int R = 256 / D4, c4 = threadIdx.x % D4, rg = threadIdx.x / D4;
    const float4* base = reinterpret_cast<const float4*>(g + b * L * D) + c4;
    float4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    int l = rg;
    for (; l + 3 * R < L; l += 4 * R) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 v = base[(long)(l + u * R) * D4];
            a[u].x += v.x; a[u].y += v.y; a[u].z += v.z; a[u].w += v.w;
        }
    }
    for (; l < L; l += R) {
        const float4 v = base[(long)l * D4];
        a[0].x += v.x; a[0].y += v.y; a[0].z += v.z; a[0].w += v.w;
    }
    float4 t;
    t.x = (a[0].x + a[1].x) + (a[2].x + a[3].x); t.y = (a[0].y + a[1].y) + (a[2].y + a[3].y);
    t.z = (a[0].z + a[1].z) + (a[2].z + a[3].z); t.w = (a[0].w + a[1].w) + (a[2].w + a[3].w);
    sh[threadIdx.x] = t;
    __syncthreads();
    if (rg == 0) {
        float4 r = sh[c4];
        for (int k = 1; k < R; ++k) {
            const float4 o = sh[k * D4 + c4];
            r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
        }
        reinterpret_cast<float4*>(Sb + b * D)[c4] = r;
    }
}

// ----------------------------------- launchers -----------------------------------
template <int HD>
int attn_launch(int which, const float* qkv, const float* o, const float* lse, const float* dO, float* out, float* aux,
                long B, int L, int D, int H, DropArgs dr, hipStream_t st) {
    const float scale = 1.0f / sqrtf((float)HD);
    dim3 grid((unsigned)(B * H), (L + 127) / 128);
    const size_t lds2 = (size_t)2 * 128 * HD * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {   // HD = 64 needs 64-65 KB of dynamic LDS
        const int want = (int)(lds2 + 256 * sizeof(float));
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<HD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, want);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<HD>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, want);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<HD>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, want);
        if (e != hipSuccess) {
            tdm_set_error("attention: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return 100 + (int)e;
        }
        attr_set = true;
    }
    if (which == 0) {
        hipLaunchKernelGGL((attn_fwd_kernel<HD>), grid, dim3(128), lds2, st, qkv, out, aux, L, D, H, scale, dr);
        TDM_CHECK_LAUNCH("attn_fwd");
    } else if (which == 1) {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<HD>), grid, dim3(128), lds2, st, qkv, o, lse, dO, out, aux, L, D, H, scale,
                           dr);
        TDM_CHECK_LAUNCH("attn_bwd_dq");
    } else {
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD>), grid, dim3(128), lds2 + 256 * sizeof(float), st, qkv, lse, dO, aux,
                           out, L, D, H, scale, dr);
        TDM_CHECK_LAUNCH("attn_bwd_dkv");
    }
    return 0;
}
int attn_dispatch(int which, int hd, const float* qkv, const float* o, const float* lse, const float* dO, float* out,
                  float* aux, long B, int L, int D, int H, DropArgs dr, hipStream_t st, float* out16 = nullptr) {
    if (g_attn_mode == 2)   // writes the S16 twin itself (and skips the fp32 gradient nobody reads)
        return tdm_launch_attn_bf16(which, hd, qkv, o, lse, dO, (which != 0 && out16 != nullptr) ? nullptr : out, out16, aux, B, L, D,
                                    H, dr, st);
    if (out16 != nullptr) {   // the fp32 cross-check kernels: fp32 result, then one split pass (after the dK/dV launch)
        TDM_TRY(attn_dispatch(which, hd, qkv, o, lse, dO, out, aux, B, L, D, H, dr, st, nullptr));
        if (which == 1) return 0;
        return tdm_launch_split_s16(out, out16, B * L * (long)(which == 0 ? D : 3 * D), st);
    }
    if (g_attn_mode == 1) return tdm_launch_attn_mfma(which, hd, qkv, o, lse, dO, out, aux, B, L, D, H, dr, st);
    switch (hd) {
        case 8: return attn_launch<8>(which, qkv, o, lse, dO, out, aux, B, L, D, H, dr, st);
        case 16: return attn_launch<16>(which, qkv, o, lse, dO, out, aux, B, L, D, H, dr, st);
        case 32: return attn_launch<32>(which, qkv, o, lse, dO, out, aux, B, L, D, H, dr, st);
        case 64: return attn_launch<64>(which, qkv, o, lse, dO, out, aux, B, L, D, H, dr, st);
    }
    tdm_set_error("attention: head_dim %d not supported (8, 16, 32, 64)", hd);
    return 1;
}

// Pre-split (S16) operands for the bf16 GEMMs: the producers of every GEMM operand (time-bias add, LayerNorm forward /
// backward, attention, the FFN GEMM epilogues, the weight transposes) write an S16 twin — or, where nothing else reads the
// fp32 tensor (the FFN hidden activation f1 and its gradient, d(qkv)), ONLY the S16 form — and the GEMM loaders copy 16-byte
// pieces into LDS instead of converting: a row panel is re-read by N / 128 column tiles and by the weight-gradient GEMM, the
// split happens once.  Needs D and the FFN width to be multiples of 16; otherwise (and in the fp32 GEMM mode) the fp32 path.
inline bool tt_use16(int D, int F) { return g_gemm_mode != 0 && (D % 16) == 0 && (F % 16) == 0; }
// linear1 -> ReLU -> dropout -> linear2 (and their data gradients) as ONE launch with the hidden tile in registers (ffn_chain.hip);
// TDM_FFN_FUSED=0 keeps the two GEMM launches (A/B timing)
inline bool tt_fused_ffn(long M, int D, int F) {
    static const bool on = !(getenv("TDM_FFN_FUSED") && atoi(getenv("TDM_FFN_FUSED")) == 0);
    return on && tt_use16(D, F) && tdm_ffn_chain_ok(M, D, F);
}

// Y[M][N] = dropout(relu(X[M][K] W[N][K]^T + bias (+res)));  s16: X and W are S16;  Y16: S16 twin of Y (Y may be nullptr)
int linear_fwd(const float* X, const float* W, const float* bias, const float* res, float* Y, float* Y16, bool s16, long M,
               int N, int K, int relu, DropArgs dr, hipStream_t st) {
    GemmArgs g{};
    g.A = X; g.a_rs = K; g.a_cs = 1;
    g.B = W; g.b_rs = 1; g.b_cs = K;
    g.C = Y; g.C16 = Y16; g.s16_in = s16 ? 1 : 0;
    g.c_rs = N; g.bias = bias; g.res = res; g.relu = relu; g.M = (int)M; g.N = N; g.K = K; g.splitk = 1;
    g.drop = dr;
    if (g_gemm_mode != 0) return tdm_launch_gemm_nt_bf16(g, g_gemm_mode == 1 ? 3 : 1, st);
    return tdm_launch_gemm(g, st);
}
// dX[M][K] = dY[M][N] W[N][K] (+res), then dX = gate > 0 ? dX * gate_scale : 0 (ReLU / FFN-dropout backward)
// s16: dY and the gate are S16, the transposed weight is built as S16;  dX16: S16 twin of dX (dX may be nullptr)
// (wT_ready: wT already holds W^T — the backward pass transposes all its weights in one launch)
// (dr: a dropout mask applied to dX in the epilogue — bf16 GEMM modes only; the fp32 mode's caller applies it afterwards)
int linear_dgrad(const float* dY, const float* W, float* wT, const float* res, const float* gate, float gate_scale,
                 float* dX, float* dX16, bool s16, long M, int N, int K, hipStream_t st, bool wT_ready = false,
                 DropArgs dr = DropArgs{}) {
    GemmArgs g{};
    g.gate = gate; g.gate_scale = gate_scale;
    if (g_gemm_mode != 0) g.drop = dr;
    if (g_gemm_mode != 0) {   // dX = dY . (W^T)^T as a K-contiguous (NT) product on the transposed weight
        if (wT_ready) {}
        else if (s16) TDM_TRY(tdm_launch_transpose_s16(W, wT, N, K, st));
        else TDM_TRY(tdm_launch_transpose(W, wT, N, K, st));
        g.A = dY; g.a_rs = N; g.a_cs = 1;
        g.B = wT; g.b_rs = 1; g.b_cs = N;
        g.C = dX; g.C16 = dX16; g.s16_in = s16 ? 1 : 0; g.gate_s16 = (s16 && gate != nullptr) ? 1 : 0;
        g.c_rs = K; g.res = res; g.M = (int)M; g.N = K; g.K = N; g.splitk = 1;
        return tdm_launch_gemm_nt_bf16(g, g_gemm_mode == 1 ? 3 : 1, st);
    }
    g.A = dY; g.a_rs = N; g.a_cs = 1;
    g.B = W; g.b_rs = K; g.b_cs = 1;
    g.C = dX; g.c_rs = K; g.res = res; g.M = (int)M; g.N = K; g.K = N; g.splitk = 1;
    return tdm_launch_gemm(g, st);
}
// dW[N][K] partials = dY[M][N]^T X[M][K], split over M into SPLITK slabs at slab_region; bias_region != nullptr
// (bf16 modes): the same kernel also writes db partials = per-split column sums of dY, [SPLITK][N];  s16: dY and X are S16
int linear_wgrad(const float* dY, const float* X, float* slab_region, float* bias_region, bool s16, long M, int N, int K,
                 hipStream_t st) {
    GemmArgs g{};
    g.A = dY; g.a_rs = 1; g.a_cs = N;
    g.B = X; g.b_rs = K; g.b_cs = 1;
    g.C = slab_region; g.c_rs = K; g.M = N; g.N = K; g.K = (int)M; g.splitk = wgrad_splitk(N, K);
    g.c_split_stride = (long)N * K;
    if (g_gemm_mode != 0) {
        g.colsum = bias_region; g.colsum_stride = (N + 63) & ~63;
        g.s16_in = s16 ? 1 : 0;
        return tdm_launch_gemm_tn_bf16(g, g_gemm_mode == 1 ? 3 : 1, st);
    }
    return tdm_launch_gemm(g, st);
}
// bias gradient: db[N] = colsum(dY) via partials + reduce
int bias_grad(const float* dY, float* part, float* db, long M, int N, hipStream_t st) {
    const int C4 = N / 4;
    const long rowblocks = (C4 <= 256) ? (M + (256 / C4) - 1) / (256 / C4) : M;
    const int nb = (int)(rowblocks < CS_SLABS ? rowblocks : CS_SLABS);
    hipLaunchKernelGGL(colsum_kernel, dim3(nb), dim3(256), 0, st, dY, part, M, N);
    TDM_CHECK_LAUNCH("colsum");
    ReduceArgs ra{};
    ra.nsec = 1; ra.sec[0].off = 0; ra.sec[0].len = N; ra.sec[0].nslab = nb;
    return tdm_launch_reduce(part, N, ra, db, st);
}
// LayerNorm backward; G = gradient buffer base: dgamma/dbeta go to G[gamma_off .. +2D), the column sums of the
// (dropped) input gradient to G[bias_off .. +D) (the bias gradient of the linear layer feeding the residual add)
inline int ln_bwd_nb(long M, int D) {   // partial rows a LayerNorm backward launch writes
    const int NQ = D <= 256 ? 1 : (D <= 512 ? 2 : 4);
    const int R = 4 / NQ;
    const long nb = (M + 4 * R - 1) / (4 * R);
    return (int)(nb > LN_SLABS ? LN_SLABS : nb);
}
// (defer: the partial rows in `part` are summed later — by the caller's slab reduction — not by a launch of their own)
int ln_bwd(const float* dy, const float* dy2, const float* s, const float* mean, const float* rstd, const float* gamma,
           float* ds, float* ds_drop, float* op16, bool dropping, DropArgs dr, float* part, float* G, long gamma_off,
           long bias_off, long M, int D, hipStream_t st, bool defer = false) {
    const int NQ = D <= 256 ? 1 : (D <= 512 ? 2 : 4);
    const long nb = ln_bwd_nb(M, D);
    if (!dropping) dr = DropArgs{};
    if (NQ == 1)
        hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3((unsigned)nb), dim3(256), 0, st, dy, dy2, s, mean, rstd, gamma, ds, ds_drop, op16, part, M, D, dr);
    else if (NQ == 2)
        hipLaunchKernelGGL(ln_bwd_kernel<2>, dim3((unsigned)nb), dim3(256), 0, st, dy, dy2, s, mean, rstd, gamma, ds, ds_drop, op16, part, M, D, dr);
    else
        hipLaunchKernelGGL(ln_bwd_kernel<4>, dim3((unsigned)nb), dim3(256), 0, st, dy, dy2, s, mean, rstd, gamma, ds, ds_drop, op16, part, M, D, dr);
    TDM_CHECK_LAUNCH("ln_bwd");
    if (defer) return 0;
    ReduceArgs ra{};
    ra.nsec = 2;
    ra.sec[0].off = (int)gamma_off; ra.sec[0].len = 2 * D; ra.sec[0].nslab = (int)nb; ra.sec[0].src_delta = -gamma_off;
    ra.sec[1].off = (int)bias_off; ra.sec[1].len = D; ra.sec[1].nslab = (int)nb; ra.sec[1].src_delta = 2L * D - bias_off;
    return tdm_launch_reduce(part, 3L * D, ra, G, st);
}
int ln_fwd(const float* x, const float* r, const float* gamma, const float* beta, float* y, float* y16, float* s, float* mean,
           float* rstd, long M, int D, hipStream_t st) {
    hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, x, r, gamma, beta, y, y16, s, mean,
                       rstd, M, D, 1e-5f);
    TDM_CHECK_LAUNCH("ln_fwd");
    return 0;
}

int tt_check(long B, int L, int D, int H, int depth, int F) {
    TDM_REQUIRE(B >= 1 && B <= 65536 && L >= 1 && L <= 4096, "transformer: B=%ld L=%d out of range", B, L);
    TDM_REQUIRE(depth >= 1 && depth <= 8, "transformer: depth %d (1..8)", depth);
    TDM_REQUIRE(D % 4 == 0 && D <= 1024 && D % H == 0, "transformer: D=%d H=%d", D, H);
    TDM_REQUIRE(F % 4 == 0 && F <= 2048, "transformer: ffn %d (<= 2048, multiple of 4)", F);
    TDM_REQUIRE(B * L * (long)(F > 3 * D ? F : 3 * D) < 2147483647L, "transformer: B*L*max(F,3D) must fit int32");
    const int hd = D / H;
    TDM_REQUIRE(hd == 8 || hd == 16 || hd == 32 || hd == 64, "transformer: head_dim %d not in {8,16,32,64}", hd);
    return 0;
}

int tt_forward(const float* P, const TTLayout& lay, const float* x, const int64_t* t, float* out, const TTWs& w, long B,
               int L, int D, int H, int depth, int F, Drop drop, hipStream_t st, bool save = true) {
    const long M = B * L;
    const DropArgs none{};
    const bool s16 = tt_use16(D, F);
    if (s16) TDM_TRY(tdm_launch_split_s16(P, w.P16, lay.total & ~15L, st));   // (the tail past the last weight matrix is biases)
    const float* PW = s16 ? w.P16 : P;   // weight matrices as GEMM operands
    {
        long n = M * D / 4;
        int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
        hipLaunchKernelGGL(add_timebias_kernel, dim3(grid), dim3(256), 0, st, x, t, P + lay.te_w, P + lay.te_b, w.that,
                           w.tb, w.L[0].hin, s16 ? w.L[0].hin16 : nullptr, B, L, D, drop.site(0));
        TDM_CHECK_LAUNCH("add_timebias");
    }
    for (int l = 0; l < depth; ++l) {
        const LayerOff& o = lay.L[l];
        const LayerWs& a = w.L[l];
        float* hout = (l + 1 < depth) ? w.L[l + 1].hin : out;
        float* hout16 = (s16 && l + 1 < depth) ? w.L[l + 1].hin16 : nullptr;
        TDM_TRY(linear_fwd(s16 ? a.hin16 : a.hin, PW + o.in_w, P + o.in_b, nullptr, a.qkv, nullptr, s16, M, 3 * D, D, 0, none, st));
        TDM_TRY(attn_dispatch(0, D / H, a.qkv, nullptr, nullptr, nullptr, a.o, a.lse, B, L, D, H, drop.site(1 + 4 * l), st,
                              s16 ? a.o16 : nullptr));
        TDM_TRY(linear_fwd(s16 ? a.o16 : a.o, PW + o.out_w, P + o.out_b, nullptr, w.abuf, nullptr, s16, M, D, D, 0,
                           drop.site(2 + 4 * l), st));
        TDM_TRY(ln_fwd(a.hin, w.abuf, P + o.n1_w, P + o.n1_b, a.h1, s16 ? a.h1_16 : nullptr, a.s1, a.mean1, a.rstd1, M, D, st));
        if (tt_fused_ffn(M, D, F)) {
            // the hidden activation stays in registers; training additionally writes it once as S16 (for linear2's weight
            // gradient) with its sign masks (the ReLU / dropout gate of the data gradient)
            const bool keep = save || drop.p > 0.f;
            TDM_TRY(tdm_launch_ffn_chain(keep ? 1 : 0, g_gemm_mode == 1 ? 3 : 1, a.h1_16, PW + o.l1_w, P + o.l1_b, PW + o.l2_w,
                                         P + o.l2_b, w.abuf, keep ? a.f1 : nullptr, keep ? a.fmask : nullptr, 1.f,
                                         drop.site(3 + 4 * l), drop.site(4 + 4 * l), M, D, F, st, w.ypart));
        } else {
        // the FFN hidden activation: S16 only in the bf16 GEMM modes (read by linear2, its weight gradient and the ReLU gate)
        TDM_TRY(linear_fwd(s16 ? a.h1_16 : a.h1, PW + o.l1_w, P + o.l1_b, nullptr, s16 ? nullptr : a.f1, s16 ? a.f1 : nullptr, s16,
                           M, F, D, 1, drop.site(3 + 4 * l), st));
        TDM_TRY(linear_fwd(a.f1, PW + o.l2_w, P + o.l2_b, nullptr, w.abuf, nullptr, s16, M, D, F, 0, drop.site(4 + 4 * l), st));
        }
        TDM_TRY(ln_fwd(a.h1, w.abuf, P + o.n2_w, P + o.n2_b, hout, hout16, a.s2, a.mean2, a.rstd2, M, D, st));
    }
    return 0;
}

int tt_backward(const float* P, const TTLayout& lay, const float* dout, float* G, float* dx, const TTWs& w, float* slabs,
                long B, int L, int D, int H, int depth, int F, Drop drop, hipStream_t st) {
    const long M = B * L;
    const bool dropping = drop.p > 0.f;
    const bool fused_bias = g_gemm_mode != 0;   // in_proj / linear1 bias gradients come out of the bf16 wgrad GEMM
    const bool s16 = tt_use16(D, F);            // GEMM operands pre-split (see tt_use16): S16 twins / S16-only tensors
    // weight gradients two per launch on the 256 x 256-tile kernel (S16 operands, widths >= 256, operands < 2 GiB)
    const bool pairs = s16 && fused_bias && wgrad_pairs_shape(D, F) && (M + 64) * (long)(3 * D > F ? 3 * D : F) * 4 < 2147483647L;
    const int nprod = g_gemm_mode == 1 ? 3 : 1;
    // {dY0, X0, region 0, bias 0, N0, K0} + {dY1, X1, ...}: dW[N][K] partials of both products in one launch
    struct WgradJob { const float* dY; const float* X; float* slab; float* bias; int N, K; };
    auto wgrad_jobs = [&](const WgradJob* jobs, int n, int sk, hipStream_t q) -> int {
        TnJobs js{};
        for (int k = 0; k < n; ++k) {
            GemmArgs g{};
            g.A = jobs[k].dY; g.a_rs = 1; g.a_cs = jobs[k].N;
            g.B = jobs[k].X; g.b_rs = jobs[k].K; g.b_cs = 1;
            g.C = jobs[k].slab; g.c_rs = jobs[k].K; g.M = jobs[k].N; g.N = jobs[k].K; g.K = (int)M; g.splitk = sk;
            g.c_split_stride = (long)jobs[k].N * jobs[k].K;
            g.colsum = jobs[k].bias; g.colsum_stride = (jobs[k].N + 63) & ~63;
            g.s16_in = 1;
            TDM_TRY(tdm_tn_ring_add_job(js, g));
        }
        return tdm_launch_gemm_tn_ring(js, nprod, q);
    };
    const float* gh = dout;  // gradient w.r.t. the current layer's output
    float* gout = nullptr;
    // transposed weights of all layers in ONE launch (12 launches of ~5 us otherwise: the weights do not change inside a step)
    const bool pre = s16 && g_gemm_mode != 0 && w.wT_all != nullptr && 4 * depth <= TDM_TRANSPOSE_BATCH;
    const long per_layer = 4L * D * D + 2L * F * D;
    // Weight-gradient GEMMs on the library's side queue (tdm_set_bwd_overlap; unet.hip describes the mechanism): each reads a
    // gradient tensor the main chain has just produced and a saved activation, and only the final slab reduction reads what it
    // writes.  Two forks per layer: {linear2, linear1} behind the FFN chain, {out-projection, in-projection} behind the
    // attention backward.  Unlike the UNet's, this backward REUSES its gradient buffers layer after layer, so before the main
    // chain overwrites one it waits for the side queue's readers of the layer above (`back` events); the two LayerNorm backwards
    // of a layer get an S16 buffer each (g16 / the otherwise unused g_d) so that those waits are a whole layer away.
    // Worth 4.5 / 5.7 / 6.1 % of the denoiser step at 32 / 64 / 128 sequences of 128 tokens (eager issue), 0.7 % at 256 and nothing
    // for the full text step there (the chain kernels fill their CUs alone): used up to 16,384 tokens.
    const bool lane = pre && fused_bias && tt_fused_ffn(M, D, F) && M <= 16384 && tdm_bwd_overlap(st) != 0;
    TdmSideLane& ln = tdm_side_lane();
    if (lane) TDM_REQUIRE(ln.init(st), "tt_backward: side stream / events could not be created");
    const hipStream_t ss = lane ? ln.side : st;
    TdmSideJoin sj{&ln, st};
    int nfork = 0;
    auto fork = [&]() -> int {
        if (!lane) return 0;
        sj.armed = true;
        hipEvent_t e = ln.ready[nfork++ & 7];
        TDM_HIP(hipEventRecord(e, st));
        TDM_HIP(hipStreamWaitEvent(ss, e, 0));
        return 0;
    };
    auto side_done = [&](int k) -> int { if (lane) TDM_HIP(hipEventRecord(ln.back[k], ss)); return 0; };
    auto wait_side = [&](int k, int l) -> int { if (lane && l + 1 < depth) TDM_HIP(hipStreamWaitEvent(st, ln.back[k], 0)); return 0; };
    // weight gradients: 0 one launch per product, 1 two launches per layer (on the side queue), 2 one launch per layer (its four
    // products' 20 tiles x 12 token splits = 240 workgroups: HALF the slab bytes of two launches — a launch writes one 256 KB
    // partial per workgroup whatever it computes, and those 64 MB cost 20-37 us per launch plus their share of the reduction)
    static const bool quad_off = getenv("TDM_TN_QUAD") && atoi(getenv("TDM_TN_QUAD")) == 0;   // A/B timing: two launches of the same splits
    const int wmode = !pairs ? 0 : ((lane || quad_off) ? 1 : 2);
    const SlabPlan sp = slab_plan(D, depth, F, wmode);
    float* const g16b = (lane || wmode == 2) ? w.g_d : w.g16;   // LayerNorm 1's S16 output (mode 2: g16 = LayerNorm 2's lives to the layer's end)
    if (pre) {
        TransposeBatch tb{};
        for (int l = 0; l < depth; ++l) {
            const LayerOff& o = lay.L[l];
            float* base = w.wT_all + l * per_layer;
            const float* src[4] = {P + o.in_w, P + o.out_w, P + o.l1_w, P + o.l2_w};
            float* dst[4] = {base, base + 3L * D * D, base + 4L * D * D, base + 4L * D * D + (long)F * D};
            const int Rv[4] = {3 * D, D, F, D}, Cv[4] = {D, D, D, F};     // W[R][Cn] -> W^T[Cn][R]
            for (int k = 0; k < 4; ++k) { tb.in[tb.n] = src[k]; tb.out[tb.n] = dst[k]; tb.R[tb.n] = Rv[k]; tb.Cn[tb.n] = Cv[k]; ++tb.n; }
        }
        TDM_TRY(tdm_launch_transpose_s16_batch(tb, st));
    }
    // sections of the slab reduction that finish layer l's gradient (its flat range [L[l].in_w, L[l].n2_b + D) is contiguous)
    auto layer_sections = [&](ReduceArgs& ra, int& n, int l) {
        const LayerOff& o = lay.L[l];
        const long offs[4] = {o.in_w, o.out_w, o.l1_w, o.l2_w};
        const long lens[4] = {3L * D * D, (long)D * D, (long)F * D, (long)D * F};
        for (int k = 0; k < 4; ++k) {
            ra.sec[n].off = (int)offs[k]; ra.sec[n].len = (int)lens[k]; ra.sec[n].nslab = sp.sk[k];
            ra.sec[n].src_delta = sp.base[l][k] - offs[k];
            ra.sec[n].stride_override = sp.len[k];
            ++n;
        }
        {   // the two LayerNorms' partial rows: d gamma | d beta (2 D at n*_w) and the column sums (D: out_proj / linear2 bias gradient)
            const int nbl = ln_bwd_nb(M, D);
            const long goff[2] = {o.n1_w, o.n2_w}, boff[2] = {o.out_b, o.l2_b};
            for (int k = 0; k < 2; ++k) {
                ra.sec[n].off = (int)goff[k]; ra.sec[n].len = 2 * D; ra.sec[n].nslab = nbl;
                ra.sec[n].src_delta = sp.ln_base[l][k] - goff[k]; ra.sec[n].stride_override = 3L * D; ++n;
                ra.sec[n].off = (int)boff[k]; ra.sec[n].len = D; ra.sec[n].nslab = nbl;
                ra.sec[n].src_delta = sp.ln_base[l][k] + 2L * D - boff[k]; ra.sec[n].stride_override = 3L * D; ++n;
            }
        }
        if (fused_bias) {
            const long boffs[2] = {o.in_b, o.l1_b};
            const int ks[2] = {0, 2};
            for (int q = 0; q < 2; ++q) {
                const int k = ks[q];
                ra.sec[n].off = (int)boffs[q]; ra.sec[n].len = sp.nout[k]; ra.sec[n].nslab = sp.sk[k];
                ra.sec[n].src_delta = sp.bias_base[l][k] - boffs[q];
                ra.sec[n].stride_override = (sp.nout[k] + 63) & ~63;
                ++n;
            }
        }
    };
    // tdm_set_early_grads (data-parallel training, eager issue): layer l's gradient is reduced as soon as its weight-gradient
    // launches have retired — on the queue they ran on — and an event marks it final (tdm_tt_wait_layer_grads): the host's collective
    // stream sums layer l over the ranks while layers l - 1 .. 0 are still in their backward.  Same slabs, same fixed order: the
    // same bits as the one reduction at the end.  (bias gradients of the fp32 GEMM mode come from their own reduction launches.)
    const bool early = tdm_cur_ctx().early_grads != 0 && fused_bias && [&] {   // (a captured call keeps the one reduction: a replay records no events)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        return hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone; }();
    if (early) TDM_REQUIRE(ln.init(st), "tt_backward: events could not be created");
    ln.part_mask = 0;
    for (int l = depth - 1; l >= 0; --l) {
        const LayerOff& o = lay.L[l];
        const LayerWs& a = w.L[l];
        float* const wT_in = pre ? w.wT_all + l * per_layer : w.wT;
        float* const wT_out = pre ? wT_in + 3L * D * D : w.wT;
        float* const wT_l1 = pre ? wT_in + 4L * D * D : w.wT2;              // W1^T: [D][F]
        float* const wT_l2 = pre ? wT_l1 + (long)F * D : w.wT;               // W2^T: [F][D]
        TDM_TRY(wait_side(0, l));   // (the layer above's linear2 / linear1 weight gradients have read g16 and g_f)
        // LayerNorm 2: hout = LN(h1 + dropout2(f2)); g_s = d(h1) residual part, g2 = d(f2); db2 = colsum(g2)
        TDM_TRY(ln_bwd(gh, nullptr, a.s2, a.mean2, a.rstd2, P + o.n2_w, w.g_s, (dropping && !s16) ? w.g_d : nullptr,
                       s16 ? w.g16 : nullptr, dropping, drop.site(4 + 4 * l), slabs + sp.ln_base[l][1], G, o.n2_w, o.l2_b, M, D, st, true));
        const float* g2 = s16 ? w.g16 : (dropping ? w.g_d : w.g_s);
        // f2 = f1 W2^T + b2, f1 = dropout(relu(z1)): d(z1) = (g2 W2) * [f1 > 0] / (1 - p) in the GEMM epilogue
        if (!lane && !pairs) TDM_TRY(linear_wgrad(g2, a.f1, slabs + sp.base[l][3], nullptr, s16, M, D, F, st));
        const bool chain = tt_fused_ffn(M, D, F);
        if (chain) {
            // d(z1) = (g2 W2) gated by the saved sign masks, d(h1) = d(z1) W1: one launch, d(z1) written once (S16) for W1's gradient
            if (!pre) {
                TDM_TRY(tdm_launch_transpose_s16(P + o.l2_w, wT_l2, D, F, st));    // W2^T: [F][D]
                TDM_TRY(tdm_launch_transpose_s16(P + o.l1_w, wT_l1, F, D, st));    // W1^T: [D][F]
            }
            TDM_TRY(tdm_launch_ffn_chain(2, g_gemm_mode == 1 ? 3 : 1, g2, wT_l2, nullptr, wT_l1, nullptr, w.g_h1, w.g_f, a.fmask,
                                         dropping ? drop.site(3 + 4 * l).scale : 1.f, DropArgs{}, DropArgs{}, M, D, F, st, w.ypart));
        } else
        TDM_TRY(linear_dgrad(g2, P + o.l2_w, wT_l2, nullptr, a.f1, dropping ? drop.site(3 + 4 * l).scale : 1.f, s16 ? nullptr : w.g_f,
                             s16 ? w.g_f : nullptr, s16, M, D, F, st, pre));
        // z1 = h1 W1^T + b1
        if (!fused_bias) TDM_TRY(bias_grad(w.g_f, w.part, G + o.l1_b, M, F, st));
        TDM_TRY(fork());
        if (wmode == 2) {
        } else if (pairs) {   // {linear1 (+ its bias gradient), linear2}: g2 and f1 are untouched until LayerNorm 1's backward below
            const WgradJob jb[2] = {{w.g_f, a.h1_16, slabs + sp.base[l][2], slabs + sp.bias_base[l][2], F, D},
                                    {g2, a.f1, slabs + sp.base[l][3], nullptr, D, F}};
            TDM_TRY(wgrad_jobs(jb, 2, sp.sk[2], ss));
        } else {
        if (lane) TDM_TRY(linear_wgrad(g2, a.f1, slabs + sp.base[l][3], nullptr, s16, M, D, F, ss));
        TDM_TRY(linear_wgrad(w.g_f, s16 ? a.h1_16 : a.h1, slabs + sp.base[l][2], fused_bias ? slabs + sp.bias_base[l][2] : nullptr,
                             s16, M, F, D, ss));
        }
        TDM_TRY(side_done(0));
        TDM_TRY(wait_side(1, l));   // (the layer above's projection weight gradients have read g16b and g_qkv16)
        if (!chain) TDM_TRY(linear_dgrad(w.g_f, P + o.l1_w, pre ? wT_l1 : w.wT, nullptr, nullptr, 1.f, w.g_h1, nullptr, s16, M, F, D, st, pre));
        // LayerNorm 1: h1 = LN(hin + dropout1(a)); d(h1) = g_h1 (FFN path) + g_s (residual)
        TDM_TRY(ln_bwd(w.g_h1, w.g_s, a.s1, a.mean1, a.rstd1, P + o.n1_w, w.g_s1, (dropping && !s16) ? w.g_d : nullptr,
                       s16 ? g16b : nullptr, dropping, drop.site(2 + 4 * l), slabs + sp.ln_base[l][0], G, o.n1_w, o.out_b, M, D, st, true));
        const float* g1 = s16 ? g16b : (dropping ? w.g_d : w.g_s1);
        // a = o Wout^T + bout
        if (!lane && !pairs) TDM_TRY(linear_wgrad(g1, s16 ? a.o16 : a.o, slabs + sp.base[l][1], nullptr, s16, M, D, D, st));
        TDM_TRY(linear_dgrad(g1, P + o.out_w, wT_out, nullptr, nullptr, 1.f, w.g_o, nullptr, s16, M, D, D, st, pre));
        // attention
        const DropArgs da = drop.site(1 + 4 * l);
        TDM_TRY(attn_dispatch(1, D / H, a.qkv, a.o, a.lse, w.g_o, w.g_qkv, w.Dvec, B, L, D, H, da, st, s16 ? w.g_qkv16 : nullptr));
        TDM_TRY(attn_dispatch(2, D / H, a.qkv, nullptr, a.lse, w.g_o, w.g_qkv, w.Dvec, B, L, D, H, da, st, s16 ? w.g_qkv16 : nullptr));
        // qkv = hin Win^T + bin ; d(hin) = g_qkv Win + g_s1 (residual)
        if (!fused_bias) TDM_TRY(bias_grad(w.g_qkv, w.part, G + o.in_b, M, 3 * D, st));
        const float* gq = s16 ? w.g_qkv16 : w.g_qkv;
        TDM_TRY(fork());
        if (wmode == 2) {   // all four: d(z1) = g_f, g2 (g16) and g1 (g_d) are not rewritten before the next layer's backward
            const WgradJob jb[4] = {{w.g_f, a.h1_16, slabs + sp.base[l][2], slabs + sp.bias_base[l][2], F, D},
                                    {g2, a.f1, slabs + sp.base[l][3], nullptr, D, F},
                                    {gq, a.hin16, slabs + sp.base[l][0], slabs + sp.bias_base[l][0], 3 * D, D},
                                    {g1, a.o16, slabs + sp.base[l][1], nullptr, D, D}};
            TDM_TRY(wgrad_jobs(jb, 4, sp.sk[0], ss));
        } else if (pairs) {   // {in_proj (+ its bias gradient), out_proj}: g1 lives until the next layer's LayerNorm 2 backward
            const WgradJob jb[2] = {{gq, a.hin16, slabs + sp.base[l][0], slabs + sp.bias_base[l][0], 3 * D, D},
                                    {g1, a.o16, slabs + sp.base[l][1], nullptr, D, D}};
            TDM_TRY(wgrad_jobs(jb, 2, sp.sk[0], ss));
        } else {
        if (lane) TDM_TRY(linear_wgrad(g1, s16 ? a.o16 : a.o, slabs + sp.base[l][1], nullptr, s16, M, D, D, ss));
        TDM_TRY(linear_wgrad(gq, s16 ? a.hin16 : a.hin, slabs + sp.base[l][0], fused_bias ? slabs + sp.bias_base[l][0] : nullptr, s16,
                             M, 3 * D, D, ss));
        }
        TDM_TRY(side_done(1));
        if (early) {   // every slab and partial row of layer l is written behind this point of `ss`
            ReduceArgs ra{};
            int n = 0;
            layer_sections(ra, n, l);
            ra.nsec = n;
            TDM_TRY(tdm_launch_reduce(slabs, 0, ra, G, ss));
            TDM_HIP(hipEventRecord(ln.part[l], ss));
            ln.part_mask |= 1u << l;
        }
        gout = (l == 0 && dx != nullptr) ? dx : w.g_h;   // layer 0: this is d(loss)/d(dropout0(x + time bias))
        // (layer 0 in the bf16 GEMM modes: the input dropout's mask — d(x + time bias) = mask * g / (1 - p) — in this epilogue)
        const bool drop_here = dropping && l == 0 && g_gemm_mode != 0;
        TDM_TRY(linear_dgrad(gq, P + o.in_w, wT_in, w.g_s1, nullptr, 1.f, gout, nullptr, s16, M, 3 * D, D, st, pre,
                             drop_here ? drop.site(0) : DropArgs{}));
        gh = gout;
    }
    if (dropping && g_gemm_mode == 0) {   // input dropout: d(x + time bias) = mask * g / (1 - p)
        const long n4 = M * D / 4;
        int grid = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
        hipLaunchKernelGGL(drop_apply_kernel, dim3(grid), dim3(256), 0, st, gout, n4, drop.site(0));
        TDM_CHECK_LAUNCH("drop_apply");
    }
    // time embedding: hin0 = x + (w*that + b)
    hipLaunchKernelGGL(seqsum_kernel, dim3((unsigned)B), dim3(256), 0, st, gh, w.Sb, L, D);
    TDM_CHECK_LAUNCH("seqsum");
    TDM_TRY(tdm_launch_time_grad(w.Sb, w.that, G + lay.te_w, G + lay.te_b, (int)B, D, st));
    if (lane) TDM_TRY(sj.join());   // the reduction reads every slab
    if (early) return 0;   // (every layer was reduced behind its own launches; the time embedding's gradient is written directly)
    // sum the split-K weight-gradient slabs
    ReduceArgs ra{};
    int n = 0;
    for (int l = 0; l < depth; ++l) {
        if (n + 10 > TDM_MAX_SECS) {   // (deep models need more sections than one launch's table holds)
            ra.nsec = n;
            TDM_TRY(tdm_launch_reduce(slabs, 0, ra, G, st));
            ra = ReduceArgs{};
            n = 0;
        }
        layer_sections(ra, n, l);
    }
    ra.nsec = n;
    return tdm_launch_reduce(slabs, 0, ra, G, st);
}

}  // namespace

extern "C" {

// (size queries return -1 for shapes the layouts are not defined for: the tables hold at most 8 layers)
int64_t tdm_tt_param_count(int D, int depth, int ffn) {
    if (depth < 1 || depth > 8 || D <= 0 || ffn <= 0 || D > 8192 || ffn > 65536) return -1;
    return tt_layout(D, depth, ffn).total;
}

int tdm_tt_param_offsets(int D, int depth, int ffn, int64_t* offs) {
    TDM_REQUIRE(depth >= 1 && depth <= 8, "tt_param_offsets: depth %d", depth);
    TDM_REQUIRE(offs != nullptr && D > 0 && ffn > 0, "tt_param_offsets: bad arguments (D=%d ffn=%d)", D, ffn);
    const TTLayout t = tt_layout(D, depth, ffn);
    for (int i = 0; i <= t.ntensor; ++i) offs[i] = t.tensor_off[i];
    return 0;
}

int64_t tdm_tt_workspace_floats(int64_t B, int L, int D, int H, int depth, int ffn, int training) {
    if (depth < 1 || depth > 8 || B < 0 || L <= 0 || D <= 0 || H <= 0 || D % H != 0 || ffn <= 0) return -1;
    if (B > (1 << 24) || L > (1 << 20) || D > 8192 || ffn > 65536 || B * (int64_t)L > ((int64_t)1 << 31)) return -1;   // (no overflow below)
    return tt_carve(nullptr, B, L, D, H, depth, ffn, training).total;
}

int64_t tdm_tt_slab_floats(int D, int depth, int ffn) {
    if (depth < 1 || depth > 8 || D <= 0 || ffn <= 0 || D > 8192 || ffn > 65536) return -1;
    const long a = slab_plan(D, depth, ffn, 0).total;   // (the arithmetic mode of the later call decides which plan runs)
    const long b = wgrad_pairs_shape(D, ffn) ? slab_plan(D, depth, ffn, 2).total : 0;
    return a > b ? a : b;
}

int tdm_tt_fwd_f32(const float* params, const float* x, const int64_t* t, float* out, float* ws, int64_t B, int L, int D,
                   int H, int depth, int ffn, int save, float p_drop, uint64_t seed, void* stream) {
    TDM_TRY(tt_check(B, L, D, H, depth, ffn));
    TDM_REQUIRE(params && x && t && out && ws, "tt_fwd: NULL pointer");
    TDM_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "tt_fwd: dropout probability %g outside [0, 1)", (double)p_drop);
    const TTLayout lay = tt_layout(D, depth, ffn);
    const TTWs w = tt_carve(ws, B, L, D, H, depth, ffn, save);
    return tt_forward(params, lay, x, t, out, w, B, L, D, H, depth, ffn, Drop{p_drop, seed}, (hipStream_t)stream, save != 0);
}

int tdm_tt_bwd_f32(const float* params, const float* dout, float* grads, float* dx, float* ws, float* slabs, int64_t B,
                   int L, int D, int H, int depth, int ffn, float p_drop, uint64_t seed, void* stream) {
    TDM_TRY(tt_check(B, L, D, H, depth, ffn));
    TDM_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "tt_bwd: dropout probability %g outside [0, 1)", (double)p_drop);
    TDM_REQUIRE(params && dout && grads && ws && slabs, "tt_bwd: NULL pointer");
    const TTLayout lay = tt_layout(D, depth, ffn);
    const TTWs w = tt_carve(ws, B, L, D, H, depth, ffn, 1);
    return tt_backward(params, lay, dout, grads, dx, w, slabs, B, L, D, H, depth, ffn, Drop{p_drop, seed},
                       (hipStream_t)stream);
}

int tdm_tt_loss_grad_f32(const float* params, const float* x0, const float* noise, const int64_t* t,
                         const float* sqrt_acp, const float* sqrt_1m_acp, float* x_noisy, float* pred, float* dpred,
                         float* loss_out, float* grads, float* ws, float* slabs, int64_t B, int L, int D, int H,
                         int depth, int ffn, float p_drop, uint64_t seed, void* stream) {
    TDM_TRY(tt_check(B, L, D, H, depth, ffn));
    TDM_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "tt_loss_grad: dropout probability %g outside [0, 1)", (double)p_drop);
    const Drop drop{p_drop, seed};
    TDM_REQUIRE(params && x0 && noise && t && x_noisy && pred && dpred && loss_out && grads && ws && slabs,
                "tt_loss_grad: NULL pointer");
    const TTLayout lay = tt_layout(D, depth, ffn);
    const TTWs w = tt_carve(ws, B, L, D, H, depth, ffn, 1);
    hipStream_t st = (hipStream_t)stream;
    TDM_TRY(tdm_q_sample_f32(x0, noise, t, sqrt_acp, sqrt_1m_acp, x_noisy, B, (int64_t)L * D, stream));
    TDM_TRY(tt_forward(params, lay, x_noisy, t, pred, w, B, L, D, H, depth, ffn, drop, st));
    TDM_TRY(tdm_mse_fwd_bwd_f32(pred, noise, loss_out, dpred, w.part, B * L * D, stream));
    return tt_backward(params, lay, dpred, grads, nullptr, w, slabs, B, L, D, H, depth, ffn, drop, st);
}

// The same step with its randomness on the device (src/shakespeare.py:221-236 incl. the draws of :228-229): t and noise from
// the Philox stream (seed, rng_state[0]) into t_buf / noise, and the dropout masks of this step = masks(drop_seed) salted with
// the low word of the (already advanced) stream offset — nothing the host writes per step, so the call is hipGraph-replayable
// with fresh draws and fresh masks on every replay.
int tdm_tt_loss_grad_philox_f32(const float* params, const float* x0, const float* sqrt_acp, const float* sqrt_1m_acp,
                                uint64_t seed, int64_t* rng_state, int64_t* t_buf, float* noise, float* x_noisy, float* pred,
                                float* dpred, float* loss_out, float* grads, float* ws, float* slabs, int64_t B, int L, int D,
                                int H, int depth, int ffn, float p_drop, uint64_t drop_seed, void* stream) {
    return tdm_tt_loss_grad_philox_dx_f32(params, x0, sqrt_acp, sqrt_1m_acp, seed, rng_state, t_buf, noise, x_noisy, pred, dpred,
                                          loss_out, grads, nullptr, ws, slabs, B, L, D, H, depth, ffn, p_drop, drop_seed, stream);
}

// ... with dx_noisy (nullable) = d loss / d x_noisy, the gradient that reaches LEARNED embeddings through q_sample
// (src/shakespeare.py:225-233: x0 = embedding_fn(token_ids) carries a gradient)
int tdm_tt_loss_grad_philox_dx_f32(const float* params, const float* x0, const float* sqrt_acp, const float* sqrt_1m_acp,
                                   uint64_t seed, int64_t* rng_state, int64_t* t_buf, float* noise, float* x_noisy, float* pred,
                                   float* dpred, float* loss_out, float* grads, float* dx_noisy, float* ws, float* slabs,
                                   int64_t B, int L, int D, int H, int depth, int ffn, float p_drop, uint64_t drop_seed,
                                   void* stream) {
    TDM_TRY(tt_check(B, L, D, H, depth, ffn));
    TDM_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "tt_loss_grad_philox: dropout probability %g outside [0, 1)", (double)p_drop);
    TDM_REQUIRE(params && x0 && sqrt_acp && sqrt_1m_acp && rng_state && t_buf && noise && x_noisy && pred && dpred && loss_out &&
                    grads && ws && slabs, "tt_loss_grad_philox: NULL pointer");
    TDM_REQUIRE(((long)L * D) % 4 == 0, "tt_loss_grad_philox: L * D must be a multiple of 4");
    Drop drop{p_drop, drop_seed};
    drop.salt = reinterpret_cast<const uint32_t*>(rng_state);    // low word of the offset (little endian), read AFTER the bump below
    const TTLayout lay = tt_layout(D, depth, ffn);
    const TTWs w = tt_carve(ws, B, L, D, H, depth, ffn, 1);
    hipStream_t st = (hipStream_t)stream;
    TDM_TRY(tdm_launch_draw_q_sample(x0, sqrt_acp, sqrt_1m_acp, seed, rng_state, t_buf, noise, x_noisy, B, (int64_t)L * D, true, st));
    TDM_TRY(tt_forward(params, lay, x_noisy, t_buf, pred, w, B, L, D, H, depth, ffn, drop, st));
    TDM_TRY(tdm_mse_fwd_bwd_f32(pred, noise, loss_out, dpred, w.part, B * L * D, stream));
    return tt_backward(params, lay, dpred, grads, dx_noisy, w, slabs, B, L, D, H, depth, ffn, drop, st);
}

int tdm_tt_p_sample_step_f32(const float* params, const float* x, const int64_t* t, const float* noise,
                             const float* tab_recip, const float* tab_eps, const float* tab_sigma, int t_index,
                             float* eps, float* x_out, float* ws, int64_t B, int L, int D, int H, int depth, int ffn,
                             void* stream) {
    TDM_TRY(tt_check(B, L, D, H, depth, ffn));
    TDM_REQUIRE(params && x && t && eps && x_out && ws, "tt_p_sample_step: NULL pointer");
    const TTLayout lay = tt_layout(D, depth, ffn);
    const TTWs w = tt_carve(ws, B, L, D, H, depth, ffn, 0);
    TDM_TRY(tt_forward(params, lay, x, t, eps, w, B, L, D, H, depth, ffn, Drop{0.f, 0}, (hipStream_t)stream, false));
    return tdm_p_sample_update_f32(x, eps, t_index == 0 ? nullptr : noise, tab_recip, tab_eps, tab_sigma, t_index, x_out,
                                   B * L * D, stream);
}

// One reverse step with device-resident step index and device-drawn noise (src/shakespeare.py:382-385, :343-352):
// eps = TinyTransformer(x, t_dev); x_out = update(x, eps, z ~ Philox); t_dev -= 1 (floor 0).  tab_sigma0[0] must be 0.
int tdm_tt_p_sample_step_philox_f32(const float* params, const float* x, int64_t* t_dev, const float* tab_recip,
                                    const float* tab_eps, const float* tab_sigma0, uint64_t seed, int64_t* rng_state,
                                    float* eps, float* x_out, float* ws, int64_t B, int L, int D, int H, int depth, int ffn,
                                    void* stream) {
    TDM_TRY(tt_check(B, L, D, H, depth, ffn));
    TDM_REQUIRE(params && x && t_dev && eps && x_out && ws && rng_state, "tt_p_sample_step_philox: NULL pointer");
    const TTLayout lay = tt_layout(D, depth, ffn);
    const TTWs w = tt_carve(ws, B, L, D, H, depth, ffn, 0);
    TDM_TRY(tt_forward(params, lay, x, t_dev, eps, w, B, L, D, H, depth, ffn, Drop{0.f, 0}, (hipStream_t)stream, false));
    return tdm_p_sample_update_philox_f32(x, eps, tab_recip, tab_eps, tab_sigma0, t_dev, seed, rng_state, x_out, B,
                                          (int64_t)L * D, stream);
}

int tdm_set_gemm_mode(int mode) {
    TDM_REQUIRE(mode >= 0 && mode <= 2, "gemm mode %d (0 = fp32 MFMA, 1 = bf16x3, 2 = bf16)", mode);
    g_gemm_mode = mode;
    return 0;
}
int tdm_get_gemm_mode(void) { return g_gemm_mode; }

// Data-parallel denoiser training (tdm_set_early_grads(1), eager issue, bf16 GEMM modes): layer l's gradient — floats
// [begin, end) of the flat gradient, tdm_tt_layer_grad_range — is final when the backward has retired that layer's weight-gradient
// launches; tdm_tt_wait_layer_grads orders `stream` behind that point of the calling thread's LAST backward (once per layer and call).
// Returns 1 if there was such an event, 0 if not (selector off, captured call, fp32 GEMM mode: order behind the call's stream), < 0 on error.
int tdm_tt_wait_layer_grads(void* stream, int layer) {
    TDM_REQUIRE(layer >= 0 && layer < 8, "tt_wait_layer_grads: layer %d", layer);
    TdmSideLane& ln = tdm_side_lane();
    if (!ln.ok || !(ln.part_mask & (1u << layer))) return 0;
    if (hipStreamWaitEvent((hipStream_t)stream, ln.part[layer], 0) != hipSuccess) {
        tdm_set_error("tt_wait_layer_grads: hipStreamWaitEvent failed");
        (void)hipGetLastError();
        return -1;
    }
    ln.part_mask &= ~(1u << layer);
    return 1;
}
int tdm_tt_layer_grad_range(int D, int depth, int ffn, int layer, int64_t* begin, int64_t* end) {
    TDM_REQUIRE(depth >= 1 && depth <= 8 && layer >= 0 && layer < depth && D > 0 && ffn > 0 && begin != nullptr && end != nullptr,
                "tt_layer_grad_range: bad arguments (D=%d depth=%d ffn=%d layer=%d)", D, depth, ffn, layer);
    const TTLayout t = tt_layout(D, depth, ffn);
    *begin = t.L[layer].in_w;
    *end = layer + 1 < depth ? t.L[layer + 1].in_w : t.te_w;
    return 0;
}

int tdm_set_attn_mode(int mode) {
    TDM_REQUIRE(mode >= 0 && mode <= 2, "attention mode %d (0 = scalar fp32, 1 = fp32 MFMA, 2 = bf16x3 MFMA)", mode);
    g_attn_mode = mode;
    return 0;
}
int tdm_get_attn_mode(void) { return g_attn_mode; }

// Per-op attention (nn.MultiheadAttention's core, src/shakespeare.py:108-111) in the selected attention mode, on the packed
// projection qkv[B][L][3 D] (q | k | v, heads = consecutive head_dim slices): softmax(q k^T / sqrt(head_dim)) (dropout) v.
int tdm_attention_fwd_f32(const float* qkv, float* o, float* lse, int64_t B, int L, int D, int H, float p_drop, uint64_t seed,
                          int site, void* stream) {
    TDM_REQUIRE(qkv != nullptr && o != nullptr && lse != nullptr && B >= 0 && L > 0 && H > 0 && D > 0 && D % H == 0,
                "attention_fwd: bad arguments (B=%lld L=%d D=%d H=%d)", (long long)B, L, D, H);
    TDM_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "attention_fwd: p_drop %g", (double)p_drop);
    if (B == 0) return 0;
    return attn_dispatch(0, D / H, qkv, nullptr, nullptr, nullptr, o, lse, B, L, D, H, tdm_drop_site(p_drop, seed, site),
                         (hipStream_t)stream);
}
// gradients wrt the packed projection: dqkv[B][L][3 D] (every element written); Dvec[B H][L] is scratch (dO . O per row)
int tdm_attention_bwd_f32(const float* qkv, const float* o, const float* lse, const float* dO, float* dqkv, float* Dvec,
                          int64_t B, int L, int D, int H, float p_drop, uint64_t seed, int site, void* stream) {
    TDM_REQUIRE(qkv != nullptr && o != nullptr && lse != nullptr && dO != nullptr && dqkv != nullptr && Dvec != nullptr &&
                    B >= 0 && L > 0 && H > 0 && D > 0 && D % H == 0,
                "attention_bwd: bad arguments (B=%lld L=%d D=%d H=%d)", (long long)B, L, D, H);
    TDM_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "attention_bwd: p_drop %g", (double)p_drop);
    if (B == 0) return 0;
    const DropArgs da = tdm_drop_site(p_drop, seed, site);
    TDM_TRY(attn_dispatch(1, D / H, qkv, o, lse, dO, dqkv, Dvec, B, L, D, H, da, (hipStream_t)stream));
    return attn_dispatch(2, D / H, qkv, nullptr, lse, dO, dqkv, Dvec, B, L, D, H, da, (hipStream_t)stream);
}

// The forms the train step launches (tt_forward / tt_backward), one kernel per call: which = 0 forward (out = O fp32, out16 =
// its S16 twin, aux = lse), 1 dQ (aux = Dvec written), 2 dK/dV (aux = Dvec read); with out16 != NULL the backward writes
// d(qkv) as S16 only.  For per-kernel timing and parity (tools/time_attn.py, tests); o / dO may be NULL where unused.
int tdm_attention_step_form_f32(int which, const float* qkv, const float* o, const float* lse, const float* dO, float* out,
                                float* out16, float* aux, int64_t B, int L, int D, int H, float p_drop, uint64_t seed, int site,
                                void* stream) {
    TDM_REQUIRE(which >= 0 && which <= 2 && qkv != nullptr && aux != nullptr && (out != nullptr || out16 != nullptr) && B >= 0 &&
                    L > 0 && H > 0 && D > 0 && D % H == 0,
                "attention_step_form: bad arguments (which=%d B=%lld L=%d D=%d H=%d)", which, (long long)B, L, D, H);
    TDM_REQUIRE(which == 0 ? out != nullptr : (lse != nullptr && dO != nullptr && (which == 2 || o != nullptr)),
                "attention_step_form: missing operand for which=%d", which);
    TDM_REQUIRE(out != nullptr || g_attn_mode == 2,
                "attention_step_form: the fp32 attention modes write the fp32 result first (out must not be NULL; only the bf16 "
                "kernels skip it)");
    TDM_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "attention_step_form: p_drop %g", (double)p_drop);
    if (B == 0) return 0;
    return attn_dispatch(which, D / H, qkv, o, which == 0 ? nullptr : lse, dO, out, aux, B, L, D, H, tdm_drop_site(p_drop, seed, site),
                         (hipStream_t)stream, out16);
}

// Per-op post-LN residual LayerNorm of nn.TransformerEncoderLayer (norm1 / norm2, src/shakespeare.py:108-111):
//   s = x + r (r may be NULL); y = (s - mean(s)) / sqrt(var(s) + 1e-5) * gamma + beta, statistics over the last dim (biased).
// s / mean / rstd (optional, all or none) are what the backward twin reads.  Dropout-free: the train-mode mask of the
// sub-layer output is applied by the producer of r.
int tdm_layernorm_residual_fwd_f32(const float* x, const float* r, const float* gamma, const float* beta, float* y, float* s,
                                   float* mean, float* rstd, int64_t M, int D, void* stream) {
    TDM_REQUIRE(x && gamma && beta && y && M >= 0 && D > 0 && D % 4 == 0 && D <= 1024,
                "layernorm_residual_fwd: bad arguments (M=%lld D=%d; D %% 4 == 0, D <= 1024)", (long long)M, D);
    TDM_REQUIRE((s != nullptr) == (mean != nullptr) && (mean != nullptr) == (rstd != nullptr),
                "layernorm_residual_fwd: s, mean and rstd are saved together or not at all");
    if (M == 0) return 0;
    return ln_fwd(x, r, gamma, beta, y, nullptr, s, mean, rstd, M, D, (hipStream_t)stream);
}
// Backward twin: given dy and the saved (s, mean, rstd): ds[M][D] = d(loss)/d(s) — the gradient of BOTH x and r — and
// dgamma_dbeta[2][D] = (sum_m dy * xhat, sum_m dy).  scratch: tdm_layernorm_scratch_floats(D) floats (per-workgroup
// partials, reduced in fixed order: deterministic).
int64_t tdm_layernorm_scratch_floats(int D) { return D < 1 ? -1 : (int64_t)LN_SLABS * 3 * D + 3 * (int64_t)D + 64; }
int tdm_layernorm_residual_bwd_f32(const float* dy, const float* s, const float* mean, const float* rstd, const float* gamma,
                                   float* ds, float* dgamma_dbeta, float* scratch, int64_t M, int D, void* stream) {
    TDM_REQUIRE(dy && s && mean && rstd && gamma && ds && dgamma_dbeta && scratch && M >= 1 && D > 0 && D % 4 == 0 && D <= 1024,
                "layernorm_residual_bwd: bad arguments (M=%lld D=%d; D %% 4 == 0, D <= 1024)", (long long)M, D);
    // ln_bwd reduces into G[gamma_off .. + 2 D) and G[bias_off .. + D): lay them out behind the partials
    float* G = scratch + (long)LN_SLABS * 3 * D;
    TDM_TRY(ln_bwd(dy, nullptr, s, mean, rstd, gamma, ds, nullptr, nullptr, false, DropArgs{}, scratch, G, 0, 2L * D, M, D,
                   (hipStream_t)stream));
    hipError_t e = hipMemcpyAsync(dgamma_dbeta, G, (size_t)2 * D * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) {
        tdm_set_error("layernorm_residual_bwd: copy failed: %s", hipGetErrorString(e));
        return 100 + (int)e;
    }
    return 0;
}

// keep[i] = 1 if element idx0 + i of dropout site `site` survives (the mask the kernels regenerate in registers)
static int dropout_keep_host(float p_drop, uint64_t seed, bool salted, uint32_t salt, int site, int64_t idx0, int64_t n, uint8_t* keep_host) {
    TDM_REQUIRE(p_drop >= 0.f && p_drop < 1.f && keep_host != nullptr && n >= 0, "dropout_keep: bad arguments");
    DropArgs d = tdm_drop_site(p_drop, seed, site);
    if (salted) d.key = tdm_salted_key(d.key, salt);
    for (int64_t i = 0; i < n; ++i) keep_host[i] = (d.thr == 0u || tdm_keep(d, (unsigned long long)(idx0 + i))) ? 1 : 0;
    return 0;
}
int tdm_dropout_keep_u8(float p_drop, uint64_t seed, int site, int64_t idx0, int64_t n, uint8_t* keep_host) {
    return dropout_keep_host(p_drop, seed, false, 0u, site, idx0, n, keep_host);
}
// ... with the site key salted THROUGH the hash (tdm_salted_key; DropArgs::salt: the device-drawn train step salts with the low word of its Philox offset)
int tdm_dropout_keep_salted_u8(float p_drop, uint64_t seed, uint32_t salt, int site, int64_t idx0, int64_t n, uint8_t* keep_host) {
    return dropout_keep_host(p_drop, seed, true, salt, site, idx0, n, keep_host);
}

// out (S16, tdm_s16.h) = split of in, n % 16 == 0 elements: every 16 consecutive floats become hi[16] | lo[16] bf16
int tdm_split_s16_f32(const float* in, float* out, int64_t n, void* stream) {
    TDM_REQUIRE(in != nullptr && out != nullptr && n >= 0, "split_s16: bad arguments");
    return tdm_launch_split_s16(in, out, (long)n, (hipStream_t)stream);
}

int tdm_gemm_f32(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs, float* C,
                 int64_t c_rs, const float* bias, const float* res, int M, int N, int K, int relu, int splitk,
                 int64_t c_split_stride, void* stream) {
    GemmArgs g{};
    g.A = A; g.a_rs = a_rs; g.a_cs = a_cs; g.B = B; g.b_rs = b_rs; g.b_cs = b_cs; g.C = C; g.c_rs = c_rs;
    g.bias = bias; g.res = res; g.M = M; g.N = N; g.K = K; g.relu = relu & 1; g.splitk = splitk;
    g.ablate = (relu >> 8) & 31;   // timing diagnostics only
    g.c_split_stride = c_split_stride;
    // relu & 2: both operands are S16 tensors (tdm_split_s16_f32); relu & 4: C is written as an S16 tensor (NT form).
    // bf16 GEMM modes only.
    g.s16_in = (relu >> 1) & 1;
    if (relu & 4) { g.C16 = C; g.C = nullptr; }
    TDM_REQUIRE((relu & 6) == 0 || g_gemm_mode != 0, "gemm: S16 operands / output exist in the bf16 GEMM modes only");
    if (g_gemm_mode != 0) {   // route the K-contiguous (NT) and token-major (TN) forms through the bf16 kernels
        const int nprod = g_gemm_mode == 1 ? 3 : 1;
        if (a_cs == 1 && b_rs == 1 && splitk <= 1 && (c_rs % 4) == 0 && (K % 4) == 0)
            return tdm_launch_gemm_nt_bf16(g, nprod, (hipStream_t)stream);
        if ((relu & 16) && a_rs == 1 && b_cs == 1 && bias == nullptr && res == nullptr && !(relu & 1)) {   // relu & 16: the 256 x 256-tile ring kernel
            TnJobs js{};
            TDM_TRY(tdm_tn_ring_add_job(js, g));
            return tdm_launch_gemm_tn_ring(js, nprod, (hipStream_t)stream);
        }
        if (a_rs == 1 && b_cs == 1 && bias == nullptr && res == nullptr && !(relu & 1) && (M % 4) == 0 && (N % 4) == 0)
            return tdm_launch_gemm_tn_bf16(g, nprod, (hipStream_t)stream);
        TDM_REQUIRE((relu & 6) == 0, "gemm: S16 operands / output need the NT or TN form");
    }
    return tdm_launch_gemm(g, (hipStream_t)stream);
}

}  // extern "C"
