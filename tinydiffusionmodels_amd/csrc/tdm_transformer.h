// Internal declarations for the transformer-denoiser kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tdm_dropout.h"

struct GemmArgs {
    const float* A; long a_rs, a_cs;   // A(i,k) = A[i*a_rs + k*a_cs]
    const float* B; long b_rs, b_cs;   // B(k,j) = B[k*b_rs + j*b_cs]
    float* C; long c_rs;               // C[i*c_rs + j]
    const float* bias;                 // [N] or nullptr
    int M, N, K;
    const float* res;                  // [M][c_rs] residual added before relu, may alias C; or nullptr
    int relu;
    int splitk; long c_split_stride;   // splitk > 1: partial z goes to C + z*c_split_stride (raw)
    // epilogue extras, applied after bias / residual / relu, in this order:
    const float* gate; float gate_scale;   // v = gate[i*c_rs + j] > 0 ? v * gate_scale : 0   (ReLU (+dropout) backward)
    DropArgs drop;                         // dropout over the flat index i*N + j
    // TN form only: colsum[z*colsum_stride + i] = sum over this split's k of A(i,k) (exact fp32; the bias gradient
    // when A = dY), written by the workgroups of tile column 0
    float* colsum; long colsum_stride;
    // Cross-entropy fusion of the rounding head (rounding.hip, src/shakespeare.py:239-240), bf16 kernels only:
    //  * ce_part != nullptr (NT, logits GEMM): besides C = logits, the epilogue writes per (row, 64-column block) the pair
    //    (max, sum exp(v - max)) to ce_part[(row * ce_nblk + block) * 2] and the logit of the row's target id to ce_tgt[row];
    //  * ce_lse != nullptr (NT as the A operand over k = vocabulary; TN as A(i = vocabulary, k = token)): A is read as
    //    ce_scale * (exp(A - ce_lse[token]) - [vocabulary index == ce_ids[token]]) for vocabulary indices < ce_V, else 0 —
    //    softmax - onehot is regenerated from the stored logits in the loader instead of being written back and re-read.
    float* ce_part; int ce_nblk; float* ce_tgt;
    const float* ce_lse; const int64_t* ce_ids; float ce_scale; int ce_V;
    int ce_voff;   // the stored logits are the vocabulary slice [ce_voff, ce_voff + ce_V): targets are compared as ce_ids - ce_voff
    // Pre-split ("S16", tdm_s16.h) operands and outputs, bf16 kernels only.  An S16 tensor has the shape and byte size of
    // its fp32 counterpart; every 64-byte group of 16 consecutive elements of a row holds hi[16] | lo[16] as bf16.
    //  * s16_in: BOTH operands are S16 (row lengths multiples of 16): the loaders copy 16-byte pieces straight into the
    //    hi / lo LDS planes — no conversion (the in-loader split was 28 % of the N = 2048 layer's time, repeated by every
    //    column tile that re-reads a row panel);
    //  * C16 != nullptr (NT): the epilogue also writes the result as S16 to C16[M][c_rs] (C may then be nullptr);
    //  * gate_s16: `gate` is an S16 tensor (its elements are >= 0: the test is "nonzero").
    int s16_in; float* C16; int gate_s16;
    int ablate;   // timing diagnostics (NT bf16 kernel; results are wrong when set): 1 no global loads after the first
                  // chunk, 2 no MFMA, 4 no epilogue stores, 8 no split / LDS stores after the first chunk
};
int tdm_launch_gemm(const GemmArgs& g, hipStream_t st);

// bf16 MFMA GEMMs (gemm_bf16.hip); nprod = 3 (hi/lo split operands, ~1e-5) or 1 (plain bf16 operands)
int tdm_launch_gemm_nt_bf16(const GemmArgs& g, int nprod, hipStream_t st);
int tdm_launch_gemm_tn_bf16(const GemmArgs& g, int nprod, hipStream_t st);
// S16-operand NT GEMM on the LDS-DMA ring kernel (gemm_ring.hip): bit-identical to tdm_launch_gemm_nt_bf16's kernels;
// serves K % 32 == 0, N % 8 == 0, operands < 2 GiB, problems of at least 32 tiles of 256 x 128
bool tdm_gemm_nt_ring_ok(const GemmArgs& g);
int tdm_launch_gemm_nt_ring(const GemmArgs& g, int nprod, hipStream_t st);
// Token-major (TN, weight-gradient) products over S16 operands on 256 x 256 tiles with an LDS-DMA operand ring
// (gemm_tn_ring.hip).  Up to TDM_TN_JOBS products of the same token count and split count run as ONE launch: build the table
// with tdm_tn_ring_add_job (GemmArgs as for tdm_launch_gemm_tn_bf16: A(i,k) = A[k*a_cs + i], B(k,j) = B[k*b_rs + j], raw
// split-K slabs at C + z*c_split_stride, optional column sums of A), then launch.  Serves M, N >= 256 (multiples of 16),
// operands < 2 GiB.  Same arithmetic as the 128 x 128-tile kernel; not the same bits (a K step's tokens sit in another order).
// Workgroup -> (tile, split): the launcher fills `map` (tile | split << 8; up to TDM_TN_MAP workgroups, <= 256 tiles and splits) so
// that the tiles of one (product, split) — which share an operand's token range — run on ONE XCD's L2 as far as the XCDs' equal
// shares allow; larger launches fall back to whole splits per XCD (split count a multiple of 8) or plain split-major order.
#define TDM_TN_JOBS 16
#define TDM_TN_MAP 512
struct TnJob { const float* A; const float* B; float* C; float* colsum; long a_cs, b_rs, c_rs, c_split_stride, colsum_stride; int M, N, tn, tile0; };
struct TnJobs { TnJob j[TDM_TN_JOBS]; int njobs, ntiles, K, splitk, ablate, use_map; unsigned short map[TDM_TN_MAP]; };
bool tdm_gemm_tn_ring_ok(const GemmArgs& g);
int tdm_tn_ring_add_job(TnJobs& js, const GemmArgs& g);
int tdm_launch_gemm_tn_ring(TnJobs& js, int nprod, hipStream_t st);
int tdm_launch_transpose(const float* in, float* out, int R, int Cn, hipStream_t st);
// out[c][r] = in[r][c] written as S16 (R % 16 == 0); out = S16 of in, elementwise over n (n % 16 == 0) floats
int tdm_launch_transpose_s16(const float* in, float* out, int R, int Cn, hipStream_t st);
// several matrices in one launch: out[j] (S16, [Cn][R]) = transpose of in[j] ([R][Cn] fp32); blk0 is filled by the launcher
constexpr int TDM_TRANSPOSE_BATCH = 32;
struct TransposeBatch { int n; const float* in[TDM_TRANSPOSE_BATCH]; float* out[TDM_TRANSPOSE_BATCH]; int R[TDM_TRANSPOSE_BATCH], Cn[TDM_TRANSPOSE_BATCH], blk0[TDM_TRANSPOSE_BATCH + 1]; };
int tdm_launch_transpose_s16_batch(TransposeBatch& tb, hipStream_t st);
int tdm_launch_split_s16(const float* in, float* out, long n, hipStream_t st);

// fp32-MFMA attention (attn_mfma.hip). which: 0 forward (out = O, aux = lse), 1 dQ (out = dqkv, aux = D written),
// 2 dK/dV (out = dqkv, aux = D read)
int tdm_launch_attn_mfma(int which, int hd, const float* qkv, const float* o, const float* lse, const float* dO, float* out,
                         float* aux, long B, int L, int D, int H, DropArgs dr, hipStream_t st);
// the same three kernels on the bf16 matrix cores with split operands (attn_bf16.hip, ~1e-5 relative): the default
// out16 != nullptr: the S16 twin of `out` is written too (backward: `out` may then be nullptr)
int tdm_launch_attn_bf16(int which, int hd, const float* qkv, const float* o, const float* lse, const float* dO, float* out,
                         float* out16, float* aux, long B, int L, int D, int H, DropArgs dr, hipStream_t st);

// Fused FFN chain (ffn_chain.hip): Y = mid(X Wa^T) Wb^T with the hidden tile in registers.  D must be 256, F % 32 == 0.
// mode 0 forward (inference), 1 forward + hidden S16 / sign masks saved, 2 data gradient (gate by the saved masks)
bool tdm_ffn_chain_ok(long M, int D, int F);
int64_t tdm_ffn_chain_mask_elems(int64_t M, int F);   // 32-bit words of the sign-mask buffer
// hidden-range segments of the chain for M tokens / floats of the partial-sum buffer they need (ffn_chain.hip)
int tdm_ffn_chain_segments(long M, int F);
long tdm_ffn_chain_part_floats(long M, int F);
// ypart: partial-sum buffer of tdm_ffn_chain_part_floats(M, F) floats, or nullptr (then one segment: small batches use M / 128 CUs)
int tdm_launch_ffn_chain(int mode, int nprod, const float* X16, const float* Wa16, const float* bias_a, const float* Wb16,
                         const float* bias_b, float* Y, float* mid16, unsigned* mask, float gate_scale, DropArgs drop_mid,
                         DropArgs drop_out, long M, int D, int F, hipStream_t st, float* ypart = nullptr);
