// Row N1 of SURVEY.md §8: the learned embedding table and the rounding head of the
// text train step (src/shakespeare.py:46-102, :225-243):
//     x0     = Embedding(V, D)(token_ids)
//     logits = Linear(D, V)(x0);  rounding_loss = cross_entropy(logits, token_ids)
// and the argmax decode of sampling (:387-390).
//
// Second version.  The (tokens x V) logits are written ONCE (6.6 GB at 32,768 tokens x 50,257 entries, in the
// caller's workspace) and read twice; nothing else of that size exists:
//   * logits = x W^T + b on the NT MFMA GEMM, whose epilogue also emits per (row, 64-column block) the pair
//     (max, sum exp) and the target logit -> a tiny row kernel folds 786 pairs per row into lse and the loss;
//   * dx = g W and dW = g^T x (+ db = column sums of g) read the stored LOGITS and regenerate
//     g = scale * (softmax - onehot) = scale * (exp(l - lse) - [v == id]) in their loaders (GemmArgs::ce_*),
//     so the gradient tensor is never written.
// (The first version overwrote the logits with g in a separate row pass: written twice, read four times.)
// Recomputing the logits inside the gradient GEMMs instead of storing them would cost two more bf16x3 GEMMs
// (5 x 2.5 TFLOP of MFMA work instead of 3 x): slower than 20 GB of HBM traffic on this machine.
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

// lse[m] = log sum_v exp(logits[m][v]) from the (max, sum exp) pairs the logits GEMM wrote per 64-column block, and
// rowloss[m] = lse[m] - logits[m][ids[m]]  (F.cross_entropy per token, src/shakespeare.py:240).  One wave per row.
__global__ __launch_bounds__(256) void ce_lse_kernel(const float* __restrict__ part, const float* __restrict__ tgt,
                                                     float* __restrict__ lse, float* __restrict__ rowloss, long M, int nblk) {
    const int lane = threadIdx.x & 63;
    for (long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (long)gridDim.x * 4) {
        const float2* row = reinterpret_cast<const float2*>(part) + m * nblk;
        float mx = -INFINITY, sm = 0.f;
        for (int k = lane; k < nblk; k += 64) {
            const float2 p = row[k];
            if (p.x > mx) { sm = sm * expf(mx - p.x); mx = p.x; }
            if (p.x != -INFINITY) sm += p.y * expf(p.x - mx);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float om = __shfl_xor(mx, o), os = __shfl_xor(sm, o);
            const float nm = fmaxf(mx, om);
            sm = (mx == -INFINITY ? 0.f : sm * expf(mx - nm)) + (om == -INFINITY ? 0.f : os * expf(om - nm));
            mx = nm;
        }
        if (lane == 0) {
            const float l = mx + logf(sm);
            lse[m] = l;
            rowloss[m] = l - tgt[m];
        }
    }
}

// y[r][:] = x[r][:] / max(||x[r]||_2, eps)   (F.normalize(x, dim=-1), eps 1e-12; src/shakespeare.py:398-399). Wave per row.
__global__ __launch_bounds__(256) void l2_normalize_kernel(const float* __restrict__ x, float* __restrict__ y, long R, int D) {
    const int lane = threadIdx.x & 63;
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < R; r += (long)gridDim.x * 4) {
        const float* xr = x + r * D;
        float ss = 0.f;
        for (int d = lane; d < D; d += 64) ss += xr[d] * xr[d];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        for (int d = lane; d < D; d += 64) y[r * D + d] = xr[d] * inv;
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

// workspace carve (floats): logits [M][Vp] | lse [M] | rowloss [M] | target logit [M] | (max, sum exp) pairs [M][nblk][2] |
// wT [D][Vp] | dW slabs [sk][Vp][D] | db slabs [sk][Vp] | xn [M][D] + [Vp]   (cosine decode: wT holds the normalised
// table, xn the normalised rows and a zero bias)
// The target logit of a row is written by the ONE tile that holds column ids[row]; an id outside [0, V) (F.cross_entropy
// raises on it, src/shakespeare.py:240) would leave the slot as stale workspace and a plausible-looking loss.  Pre-filling
// with NaN (all-ones bytes) makes such a batch show up as a NaN loss instead.
static int ce_poison_targets(float* tgt, long M, hipStream_t st) {
    hipError_t e = hipMemsetAsync(tgt, 0xFF, (size_t)M * sizeof(float), st);
    if (e != hipSuccess) {
        tdm_set_error("round_ce: memset failed: %s", hipGetErrorString(e));
        return 100 + (int)e;
    }
    return 0;
}

struct RoundWs { float *logits, *lse, *rowloss, *tgt, *part, *wT, *wslab, *bslab, *xn; long total; int Vp, sk, nblk; };
RoundWs round_carve(float* base, long M, int V, int D) {
    RoundWs w{};
    long off = 0;
    auto take = [&](long n) { float* p = base ? base + off : nullptr; off += (n + 63) & ~63L; return p; };
    w.Vp = (int)pad4(V);
    w.sk = round_splitk(w.Vp, D);
    w.nblk = ((V + 127) / 128) * 2;          // 64-column blocks the NT GEMM's 128-wide tiles cover
    w.logits = take(M * w.Vp); w.lse = take(M); w.rowloss = take(M); w.tgt = take(M); w.part = take(M * w.nblk * 2);
    w.wT = take((long)D * w.Vp);
    w.wslab = take((long)w.sk * w.Vp * D);
    w.bslab = take((long)w.sk * ((w.Vp + 63) & ~63));
    w.xn = take(M * D + w.Vp);               // cosine decode: normalised rows + a zero bias
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
    if (M < 1 || V < 2 || D < 4 || M > ((int64_t)1 << 31) || D > 8192) return -1;
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
    const float gscale = grad_scale / (float)M;
    TDM_TRY(ce_poison_targets(w.tgt, (long)M, st));
    // logits = x W^T + b, with the cross-entropy partials in the epilogue            (src/shakespeare.py:239)
    {
        GemmArgs g{};
        g.A = x; g.a_rs = D; g.a_cs = 1;
        g.B = W; g.b_rs = 1; g.b_cs = D;
        g.C = w.logits; g.c_rs = Vp; g.bias = b; g.M = (int)M; g.N = V; g.K = D; g.splitk = 1;
        g.ce_part = w.part; g.ce_nblk = w.nblk; g.ce_tgt = w.tgt; g.ce_ids = ids;
        TDM_TRY(tdm_launch_gemm_nt_bf16(g, 3, st));
    }
    // lse per token and the mean cross-entropy                                       (src/shakespeare.py:240)
    hipLaunchKernelGGL(ce_lse_kernel, dim3((unsigned)((M + 3) / 4 < 4096 ? (M + 3) / 4 : 4096)), dim3(256), 0, st, w.part, w.tgt,
                       w.lse, w.rowloss, (long)M, w.nblk);
    TDM_CHECK_LAUNCH("ce_lse");
    hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, st, w.rowloss, loss_out, (long)M);
    TDM_CHECK_LAUNCH("ce_mean");
    // dx = g W with g = grad_scale / M * (softmax - onehot) regenerated from the logits in the loader
    // (K-contiguous product on the transposed, zero-padded weight)
    if (dx != nullptr) {
        dim3 tg((D + 31) / 32, (Vp + 31) / 32);
        hipLaunchKernelGGL(transpose_pad_kernel, tg, dim3(256), 0, st, W, w.wT, V, D, Vp);
        TDM_CHECK_LAUNCH("transpose_pad");
        GemmArgs g{};
        g.A = w.logits; g.a_rs = Vp; g.a_cs = 1;
        g.B = w.wT; g.b_rs = 1; g.b_cs = Vp;
        g.C = dx; g.c_rs = D; g.M = (int)M; g.N = D; g.K = Vp; g.splitk = 1;
        g.ce_lse = w.lse; g.ce_ids = ids; g.ce_scale = gscale; g.ce_V = V;
        TDM_TRY(tdm_launch_gemm_nt_bf16(g, 3, st));
    }
    // dW = g^T x (split over tokens) and db = column sums of g from the same pass, g regenerated likewise
    {
        GemmArgs g{};
        g.A = w.logits; g.a_rs = 1; g.a_cs = Vp;
        g.B = x; g.b_rs = D; g.b_cs = 1;
        g.C = w.wslab; g.c_rs = D; g.M = Vp; g.N = D; g.K = (int)M; g.splitk = w.sk;
        g.c_split_stride = (long)Vp * D;
        g.colsum = w.bslab; g.colsum_stride = (Vp + 63) & ~63;
        g.ce_lse = w.lse; g.ce_ids = ids; g.ce_scale = gscale; g.ce_V = V;
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

// ---- the same loss and gradients WITHOUT ever holding the (M, V) logits: one statistics pass over the vocabulary (the logits
// GEMM with its cross-entropy epilogue and no store), then per vocabulary chunk [v0, v0 + Vc): recompute the chunk's logits into
// a (M, Vc) scratch and feed it to the two gradient GEMMs (softmax - onehot regenerated in their loaders, dx accumulated through
// the residual input).  One GEMM pass more than the stored-logits form, 1 / (V / Vc) of its workspace.
struct RoundChunkWs { float *clog, *lse, *rowloss, *tgt, *part, *wT, *wslab, *bslab; long total; int Vcp, sk, nblk; };
RoundChunkWs round_chunk_carve(float* base, long M, int V, int D, int Vc) {
    RoundChunkWs w{};
    long off = 0;
    auto take = [&](long n) { float* p = base ? base + off : nullptr; off += (n + 63) & ~63L; return p; };
    w.Vcp = (int)pad4(Vc);
    w.sk = round_splitk(w.Vcp, D);
    w.nblk = ((V + 127) / 128) * 2;
    w.clog = take(M * w.Vcp); w.lse = take(M); w.rowloss = take(M); w.tgt = take(M); w.part = take(M * w.nblk * 2);
    w.wT = take((long)D * w.Vcp);
    w.wslab = take((long)w.sk * w.Vcp * D);
    w.bslab = take((long)w.sk * ((w.Vcp + 63) & ~63));
    w.total = off;
    return w;
}

int64_t tdm_round_workspace_chunked_floats(int64_t M, int V, int D, int Vc) {
    if (M < 1 || V < 2 || D < 4 || Vc < 128 || (Vc % 128) != 0 || M > ((int64_t)1 << 31) || D > 8192) return -1;
    return round_chunk_carve(nullptr, M, V, D, Vc).total;
}

int tdm_round_ce_loss_grad_chunked_f32(const float* x, const float* W, const float* b, const int64_t* ids, float grad_scale,
                                       float* loss_out, float* dx, float* dW, float* db, float* ws, int64_t M, int V, int D, int Vc,
                                       void* stream) {
    TDM_TRY(round_check(M, V, D));
    TDM_REQUIRE(x && W && b && ids && loss_out && dW && db && ws, "round_ce_loss_grad_chunked: NULL pointer");
    TDM_REQUIRE(Vc >= 128 && (Vc % 128) == 0, "round_ce_loss_grad_chunked: chunk of %d vocabulary entries (a multiple of 128)", Vc);
    hipStream_t st = (hipStream_t)stream;
    const RoundChunkWs w = round_chunk_carve(ws, M, V, D, Vc);
    const float gscale = grad_scale / (float)M;
    TDM_TRY(ce_poison_targets(w.tgt, (long)M, st));
    {   // statistics pass: (max, sum exp) per 64-column block and the target logit of every row; nothing else is stored
        GemmArgs g{};
        g.A = x; g.a_rs = D; g.a_cs = 1;
        g.B = W; g.b_rs = 1; g.b_cs = D;
        g.C = nullptr; g.c_rs = pad4(V); g.bias = b; g.M = (int)M; g.N = V; g.K = D; g.splitk = 1;
        g.ce_part = w.part; g.ce_nblk = w.nblk; g.ce_tgt = w.tgt; g.ce_ids = ids;
        TDM_TRY(tdm_launch_gemm_nt_bf16(g, 3, st));
    }
    hipLaunchKernelGGL(ce_lse_kernel, dim3((unsigned)((M + 3) / 4 < 4096 ? (M + 3) / 4 : 4096)), dim3(256), 0, st, w.part, w.tgt,
                       w.lse, w.rowloss, (long)M, w.nblk);
    TDM_CHECK_LAUNCH("ce_lse");
    hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, st, w.rowloss, loss_out, (long)M);
    TDM_CHECK_LAUNCH("ce_mean");
    for (int v0 = 0, ci = 0; v0 < V; v0 += Vc, ++ci) {
        const int vn = V - v0 < Vc ? V - v0 : Vc;     // vocabulary entries of this chunk
        const int vp = (int)pad4(vn);                 // row length of its logits (zero-weight padding columns)
        TDM_TRY(logits_gemm(x, W + (long)v0 * D, b + v0, w.clog, M, vn, vp, D, st));
        if (dx != nullptr) {
            dim3 tg((D + 31) / 32, (vp + 31) / 32);
            hipLaunchKernelGGL(transpose_pad_kernel, tg, dim3(256), 0, st, W + (long)v0 * D, w.wT, vn, D, vp);
            TDM_CHECK_LAUNCH("transpose_pad");
            GemmArgs g{};
            g.A = w.clog; g.a_rs = vp; g.a_cs = 1;
            g.B = w.wT; g.b_rs = 1; g.b_cs = vp;
            g.C = dx; g.c_rs = D; g.M = (int)M; g.N = D; g.K = vp; g.splitk = 1;
            g.res = ci > 0 ? dx : nullptr;            // accumulate over the chunks
            g.ce_lse = w.lse; g.ce_ids = ids; g.ce_scale = gscale; g.ce_V = vn; g.ce_voff = v0;
            TDM_TRY(tdm_launch_gemm_nt_bf16(g, 3, st));
        }
        {
            GemmArgs g{};
            g.A = w.clog; g.a_rs = 1; g.a_cs = vp;
            g.B = x; g.b_rs = D; g.b_cs = 1;
            g.C = w.wslab; g.c_rs = D; g.M = vp; g.N = D; g.K = (int)M; g.splitk = w.sk;
            g.c_split_stride = (long)w.Vcp * D;
            g.colsum = w.bslab; g.colsum_stride = (w.Vcp + 63) & ~63;
            g.ce_lse = w.lse; g.ce_ids = ids; g.ce_scale = gscale; g.ce_V = vn; g.ce_voff = v0;
            TDM_TRY(tdm_launch_gemm_tn_bf16(g, 3, st));
            ReduceArgs ra{};
            ra.nsec = 1;
            ra.sec[0].off = 0; ra.sec[0].len = (int)((long)vn * D); ra.sec[0].nslab = w.sk; ra.sec[0].stride_override = (long)w.Vcp * D;
            TDM_TRY(tdm_launch_reduce(w.wslab, 0, ra, dW + (long)v0 * D, st));
            ra.sec[0].len = vn; ra.sec[0].stride_override = (w.Vcp + 63) & ~63;
            TDM_TRY(tdm_launch_reduce(w.bslab, 0, ra, db + v0, st));
        }
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

// Cosine-similarity decode (the reference's fallback when no rounding head is trained, src/shakespeare.py:393-401):
// out_ids[m] = argmax_v <x[m] / max(|x[m]|, eps), E[v] / max(|E[v]|, eps)>: two row-normalisation passes, the logits GEMM
// with a zero bias, the row argmax.  ws: tdm_round_workspace_floats(M, V, D).
int tdm_cosine_argmax_f32(const float* x, const float* E, int64_t* out_ids, float* ws, int64_t M, int V, int D, void* stream) {
    TDM_TRY(round_check(M, V, D));
    TDM_REQUIRE(x && E && out_ids && ws, "cosine_argmax: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    const RoundWs w = round_carve(ws, M, V, D);
    float* En = w.wT;                      // [V][D]  (fits: the carve reserves D * Vp floats)
    float* xn = w.xn;                      // [M][D] then the zero bias [Vp]
    float* zb = xn + M * D;
    hipLaunchKernelGGL(l2_normalize_kernel, dim3((unsigned)((V + 3) / 4 < 4096 ? (V + 3) / 4 : 4096)), dim3(256), 0, st, E, En, (long)V, D);
    TDM_CHECK_LAUNCH("l2_normalize(table)");
    hipLaunchKernelGGL(l2_normalize_kernel, dim3((unsigned)((M + 3) / 4 < 4096 ? (M + 3) / 4 : 4096)), dim3(256), 0, st, x, xn, (long)M, D);
    TDM_CHECK_LAUNCH("l2_normalize(x)");
    hipError_t e = hipMemsetAsync(zb, 0, (size_t)w.Vp * sizeof(float), st);
    if (e != hipSuccess) { tdm_set_error("cosine_argmax: memset failed: %s", hipGetErrorString(e)); return 100 + (int)e; }
    TDM_TRY(logits_gemm(xn, En, zb, w.logits, M, V, w.Vp, D, st));
    hipLaunchKernelGGL(row_argmax_kernel, dim3((unsigned)(M < 8192 ? M : 8192)), dim3(256), 0, st, w.logits, out_ids, (long)M, V,
                       (long)w.Vp);
    TDM_CHECK_LAUNCH("row_argmax");
    return 0;
}

}  // extern "C"
