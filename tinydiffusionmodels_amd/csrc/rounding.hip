// Row N1 of SURVEY.md §8: the learned embedding table and the rounding head of the
// text train step (src/shakespeare.py:46-102, :225-243):
//     x0     = Embedding(V, D)(token_ids)
//     logits = Linear(D, V)(x0);  rounding_loss = cross_entropy(logits, token_ids)
// and the argmax decode of sampling (:387-390).
//
// First native version, sized for 288 GB of HBM rather than for a 16 GB card: the
// (tokens x V) logits are materialised once in the caller's workspace (6.6 GB at 32,768
// tokens x 50,257 entries) and overwritten in place by their gradient, so the head is three
// MFMA GEMMs on the transformer's kernels (logits = x W^T + b; dx = g W; dW = g^T x with the
// bias gradient from the same pass) plus one row-wise kernel (online log-sum-exp, then
// softmax - onehot in place).  Fusing the row work into the GEMM epilogue / loaders is the follow-up.
#include <math.h>
#include "tdm_common.h"
#include "tdm_transformer.h"

namespace {

// out[m][:] = table[ids[m]][:]                                     (src/shakespeare.py:67)
__global__ __launch_bounds__(256) void embed_gather_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids,
                                                           float* __restrict__ out, long M, int V, int D4) {
    const long total = M * D4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / D4;
        const int c = (int)(i - m * D4);
        long id = ids[m];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);   // ids are validated on the host; never read out of bounds
        reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(table + id * (long)D4 * 4)[c];
    }
}

// dtable[ids[m]][:] += scale * g[m][:]   (gradient of the gather; float atomics like torch's GPU embedding backward,
// so the summation order over repeated ids is not fixed)
__global__ __launch_bounds__(256) void embed_scatter_add_kernel(const float* __restrict__ g, const int64_t* __restrict__ ids,
                                                                float* __restrict__ dtable, long M, int V, int D,
                                                                float scale) {
    const long total = M * D;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / D;
        const int d = (int)(i - m * D);
        const long id = ids[m];
        if (id >= 0 && id < V) atomicAdd(dtable + id * D + d, scale * g[i]);
    }
}

// One workgroup per token row, two sweeps over the row in the same kernel:
//   1. online (max, sum of exp) -> lse[m]; rowloss[m] = lse - logits[m][ids[m]]
//   2. in place  logits[m][v] <- scale * (softmax(logits[m])[v] - [v == ids[m]]),  padding columns [V, ld) <- 0
// The second sweep re-reads a 200 KB row the workgroup has just streamed (L2 hit), so HBM sees one read and one write
// of the logits instead of the three reads + one write of separate statistics / gradient passes.
constexpr int CE_BLOCK = 1024;   // 16 waves stream one 200 KB row: the kernel is latency-bound with fewer
__global__ __launch_bounds__(CE_BLOCK) void ce_softmax_grad_kernel(float* __restrict__ logits, const int64_t* __restrict__ ids,
                                                              float* __restrict__ lse, float* __restrict__ rowloss, long M,
                                                              int V, long ld, float scale) {
    __shared__ float redm[CE_BLOCK / 64], reds[CE_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ld4 = (int)(ld >> 2);
    for (long m = blockIdx.x; m < M; m += gridDim.x) {
        float* row = logits + m * ld;
        long tgtl = ids[m];
        tgtl = tgtl < 0 ? 0 : (tgtl >= V ? V - 1 : tgtl);   // never index out of bounds on a bad id
        const int tgt = (int)tgtl;
        float mx = -INFINITY, sm = 0.f;
        for (int q = threadIdx.x; q < ld4; q += CE_BLOCK) {
            const float4 x = reinterpret_cast<const float4*>(row)[q];
            const float r[4] = {x.x, x.y, x.z, x.w};
            float lm = -INFINITY;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (q * 4 + e < V) lm = fmaxf(lm, r[e]);
            if (lm > mx) { sm *= expf(mx - lm); mx = lm; }
#pragma unroll
            for (int e = 0; e < 4; ++e) if (q * 4 + e < V) sm += expf(r[e] - mx);
        }
        // combine the 256 (max, sum) pairs: wave shuffles, then the 4 waves through LDS
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float om = __shfl_xor(mx, o), os = __shfl_xor(sm, o);
            const float nm = fmaxf(mx, om);
            sm = (mx == -INFINITY ? 0.f : sm * expf(mx - nm)) + (om == -INFINITY ? 0.f : os * expf(om - nm));
            mx = nm;
        }
        __syncthreads();
        if (lane == 0) { redm[wave] = mx; reds[wave] = sm; }
        __syncthreads();
        float gm = redm[0];
#pragma unroll
        for (int w = 1; w < CE_BLOCK / 64; ++w) gm = fmaxf(gm, redm[w]);
        float gs = 0.f;
#pragma unroll
        for (int w = 0; w < CE_BLOCK / 64; ++w) gs += redm[w] == -INFINITY ? 0.f : reds[w] * expf(redm[w] - gm);
        const float l = gm + logf(gs);
        if (threadIdx.x == 0) { lse[m] = l; rowloss[m] = l - row[tgt]; }
        __syncthreads();   // row[tgt] is read before the row is overwritten
        for (int q = threadIdx.x; q < ld4; q += CE_BLOCK) {
            const float4 x = reinterpret_cast<const float4*>(row)[q];
            float r[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int v = q * 4 + e;
                r[e] = v < V ? scale * (expf(r[e] - l) - (v == tgt ? 1.f : 0.f)) : 0.f;
            }
            reinterpret_cast<float4*>(row)[q] = make_float4(r[0], r[1], r[2], r[3]);
        }
    }
}

// loss_out[0] = mean(rowloss) in fixed order
__global__ __launch_bounds__(256) void ce_mean_kernel(const float* __restrict__ rowloss, float* __restrict__ loss_out, long M) {
    __shared__ float sh[256];
    float a = 0.f;
    for (long m = threadIdx.x; m < M; m += 256) a += rowloss[m];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[0] = sh[0] / (float)M;
}

// wT[d][v] = W[v][d] for v < V, 0 for the padding columns v in [V, ldv)
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ W, float* __restrict__ wT, int V, int D,
                                                            int ldv) {
    __shared__ float t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;   // bx over d, by over v
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int v = by + ty + 8 * k, d = bx + tx;
        t[ty + 8 * k][tx] = (v < V && d < D) ? W[(long)v * D + d] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int d = bx + ty + 8 * k, v = by + tx;
        if (d < D && v < ldv) wT[(long)d * ldv + v] = t[tx][ty + 8 * k];
    }
}

// out_ids[m] = argmax_v logits[m][v] (first maximum, like torch.argmax on ties of exact equality)
__global__ __launch_bounds__(256) void row_argmax_kernel(const float* __restrict__ logits, int64_t* __restrict__ out_ids,
                                                         long M, int V, long ld) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    for (long m = blockIdx.x; m < M; m += gridDim.x) {
        const float* row = logits + m * ld;
        float best = -INFINITY;
        int idx = 0x7fffffff;
        for (int v = threadIdx.x; v < V; v += 256) {
            const float x = row[v];
            if (x > best) { best = x; idx = v; }
        }
        __syncthreads();
        bv[threadIdx.x] = best; bi[threadIdx.x] = idx;
        __syncthreads();
        for (int o = 128; o >= 1; o >>= 1) {
            if ((int)threadIdx.x < o) {
                const float ob = bv[threadIdx.x + o];
                const int oi = bi[threadIdx.x + o];
                if (ob > bv[threadIdx.x] || (ob == bv[threadIdx.x] && oi < bi[threadIdx.x])) { bv[threadIdx.x] = ob; bi[threadIdx.x] = oi; }
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) out_ids[m] = bi[0];
    }
}

inline int grid_for(long n) { long g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }
inline long pad4(long v) { return (v + 3) & ~3L; }
inline int round_splitk(int Vp, int D) {
    const int tiles = ((Vp + 127) / 128) * ((D + 127) / 128);
    int sk = (1024 + tiles - 1) / tiles;
    return sk < 1 ? 1 : (sk > 64 ? 64 : sk);
}

// workspace carve (floats): logits [M][Vp] | lse [M] | rowloss [M] | wT [D][Vp] | dW slabs [sk][Vp][D] | db slabs [sk][Vp]
struct RoundWs { float *logits, *lse, *rowloss, *wT, *wslab, *bslab; long total; int Vp, sk; };
RoundWs round_carve(float* base, long M, int V, int D) {
    RoundWs w{};
    long off = 0;
    auto take = [&](long n) { float* p = base ? base + off : nullptr; off += (n + 63) & ~63L; return p; };
    w.Vp = (int)pad4(V);
    w.sk = round_splitk(w.Vp, D);
    w.logits = take(M * w.Vp); w.lse = take(M); w.rowloss = take(M);
    w.wT = take((long)D * w.Vp);
    w.wslab = take((long)w.sk * w.Vp * D);
    w.bslab = take((long)w.sk * ((w.Vp + 63) & ~63));
    w.total = off;
    return w;
}

int round_check(long M, int V, int D) {
    TDM_REQUIRE(M >= 1 && V >= 2 && D >= 4 && (D % 4) == 0, "rounding head: M=%ld V=%d D=%d (D must be a multiple of 4)", M, V, D);
    TDM_REQUIRE(M * pad4(V) < (1L << 40) && M < 2147483647L && (long)pad4(V) * D < 2147483647L, "rounding head: problem too large");
    return 0;
}

int logits_gemm(const float* x, const float* W, const float* b, float* logits, long M, int V, int Vp, int D, hipStream_t st) {
    GemmArgs g{};
    g.A = x; g.a_rs = D; g.a_cs = 1;
    g.B = W; g.b_rs = 1; g.b_cs = D;
    g.C = logits; g.c_rs = Vp; g.bias = b; g.M = (int)M; g.N = V; g.K = D; g.splitk = 1;
    return tdm_launch_gemm_nt_bf16(g, 3, st);
}

}  // namespace

extern "C" {

int tdm_embed_gather_f32(const float* table, const int64_t* ids, float* out, int64_t M, int V, int D, void* stream) {
    TDM_REQUIRE(table && ids && out && M >= 1 && V >= 1 && D >= 4 && (D % 4) == 0, "embed_gather: bad arguments (D %% 4 == 0)");
    hipLaunchKernelGGL(embed_gather_kernel, dim3(grid_for(M * (D / 4))), dim3(256), 0, (hipStream_t)stream, table, ids, out,
                       (long)M, V, D / 4);
    TDM_CHECK_LAUNCH("embed_gather");
    return 0;
}

int tdm_embed_scatter_add_f32(const float* g, const int64_t* ids, float* dtable, int64_t M, int V, int D, float scale,
                              void* stream) {
    TDM_REQUIRE(g && ids && dtable && M >= 1 && V >= 1 && D >= 1, "embed_scatter_add: bad arguments");
    hipLaunchKernelGGL(embed_scatter_add_kernel, dim3(grid_for(M * D)), dim3(256), 0, (hipStream_t)stream, g, ids, dtable,
                       (long)M, V, D, scale);
    TDM_CHECK_LAUNCH("embed_scatter_add");
    return 0;
}

int64_t tdm_round_workspace_floats(int64_t M, int V, int D) {
    if (M < 1 || V < 2 || D < 4) return -1;
    return round_carve(nullptr, M, V, D).total;
}

int tdm_round_ce_loss_grad_f32(const float* x, const float* W, const float* b, const int64_t* ids, float grad_scale,
                               float* loss_out, float* dx, float* dW, float* db, float* ws, int64_t M, int V, int D,
                               void* stream) {
    TDM_TRY(round_check(M, V, D));
    TDM_REQUIRE(x && W && b && ids && loss_out && dW && db && ws, "round_ce_loss_grad: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    const RoundWs w = round_carve(ws, M, V, D);
    const int Vp = w.Vp;
    // logits = x W^T + b                                            (src/shakespeare.py:239)
    TDM_TRY(logits_gemm(x, W, b, w.logits, M, V, Vp, D, st));
    // cross-entropy, mean over tokens                               (src/shakespeare.py:240)
    // g = grad_scale * (softmax - onehot) / M, in place over the logits; lse and the per-row loss on the way
    hipLaunchKernelGGL(ce_softmax_grad_kernel, dim3((unsigned)(M < 16384 ? M : 16384)), dim3(CE_BLOCK), 0, st, w.logits, ids, w.lse,
                       w.rowloss, (long)M, V, (long)Vp, grad_scale / (float)M);
    TDM_CHECK_LAUNCH("ce_softmax_grad");
    hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, st, w.rowloss, loss_out, (long)M);
    TDM_CHECK_LAUNCH("ce_mean");
    // dx = g W  (K-contiguous product on the transposed, zero-padded weight)
    if (dx != nullptr) {
        dim3 tg((D + 31) / 32, (Vp + 31) / 32);
        hipLaunchKernelGGL(transpose_pad_kernel, tg, dim3(256), 0, st, W, w.wT, V, D, Vp);
        TDM_CHECK_LAUNCH("transpose_pad");
        GemmArgs g{};
        g.A = w.logits; g.a_rs = Vp; g.a_cs = 1;
        g.B = w.wT; g.b_rs = 1; g.b_cs = Vp;
        g.C = dx; g.c_rs = D; g.M = (int)M; g.N = D; g.K = Vp; g.splitk = 1;
        TDM_TRY(tdm_launch_gemm_nt_bf16(g, 3, st));
    }
    // dW = g^T x (split over tokens) and db = column sums of g from the same pass
    {
        GemmArgs g{};
        g.A = w.logits; g.a_rs = 1; g.a_cs = Vp;
        g.B = x; g.b_rs = D; g.b_cs = 1;
        g.C = w.wslab; g.c_rs = D; g.M = Vp; g.N = D; g.K = (int)M; g.splitk = w.sk;
        g.c_split_stride = (long)Vp * D;
        g.colsum = w.bslab; g.colsum_stride = (Vp + 63) & ~63;
        TDM_TRY(tdm_launch_gemm_tn_bf16(g, 3, st));
        ReduceArgs ra{};
        ra.nsec = 1;
        ra.sec[0].off = 0; ra.sec[0].len = (int)((long)V * D); ra.sec[0].nslab = w.sk; ra.sec[0].stride_override = (long)Vp * D;
        TDM_TRY(tdm_launch_reduce(w.wslab, 0, ra, dW, st));
        ra.sec[0].len = V; ra.sec[0].stride_override = (Vp + 63) & ~63;
        TDM_TRY(tdm_launch_reduce(w.bslab, 0, ra, db, st));
    }
    return 0;
}

int tdm_round_logits_f32(const float* x, const float* W, const float* b, float* logits, int64_t ld, int64_t M, int V, int D,
                         void* stream) {
    TDM_TRY(round_check(M, V, D));
    TDM_REQUIRE(x && W && b && logits && ld >= V && (ld % 4) == 0, "round_logits: bad arguments (ld >= V, ld %% 4 == 0)");
    return logits_gemm(x, W, b, logits, M, V, (int)ld, D, (hipStream_t)stream);
}

int tdm_round_argmax_f32(const float* x, const float* W, const float* b, int64_t* out_ids, float* ws, int64_t M, int V, int D,
                         void* stream) {
    TDM_TRY(round_check(M, V, D));
    TDM_REQUIRE(x && W && b && out_ids && ws, "round_argmax: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    const RoundWs w = round_carve(ws, M, V, D);
    TDM_TRY(logits_gemm(x, W, b, w.logits, M, V, w.Vp, D, st));
    hipLaunchKernelGGL(row_argmax_kernel, dim3((unsigned)(M < 8192 ? M : 8192)), dim3(256), 0, st, w.logits, out_ids, (long)M, V,
                       (long)w.Vp);
    TDM_CHECK_LAUNCH("row_argmax");
    return 0;
}

}  // extern "C"
