// Internal declarations shared by the HIP translation units of libtdm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "tdm_hip.h"

void tdm_set_error(const char* fmt, ...);

#define TDM_CHECK_LAUNCH(name)                                                     \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            tdm_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));  \
            return 100 + (int)e__;                                                 \
        }                                                                          \
    } while (0)

#define TDM_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            tdm_set_error(__VA_ARGS__);        \
            return 1;                          \
        }                                      \
    } while (0)

// Timing diagnostics (runtime `ablate` bits, shader-clock phase probes) exist in DIAGNOSTIC builds only
// (TDM_BUILD_DEFINES=-DTDM_DIAG python -m tinydiffusionmodels_amd.build; tools/ablate_*.py, tools/phase_probe.py, tools/time_ring.py
// --ablate): in the product build TDM_ABLATE(x) is the constant 0, so the hot kernels carry no diagnostic branch or register.
#ifdef TDM_DIAG
#define TDM_ABLATE(x) (x)
#define TDM_DIAG_BUILD 1
#else
#define TDM_ABLATE(x) 0
#define TDM_DIAG_BUILD 0
#endif

#define TDM_HIP(expr)                                                              \
    do {                                                                           \
        hipError_t e__ = (expr);                                                   \
        if (e__ != hipSuccess) {                                                   \
            tdm_set_error("%s: %s", #expr, hipGetErrorString(e__));                \
            return 100 + (int)e__;                                                 \
        }                                                                          \
    } while (0)

// The backward passes' second launch queue (tdm_set_bwd_overlap; defined and described in unet.hip): one per CONTEXT (TdmCtx below),
// created on first use.  `ready`: main -> side dependencies (forks); `back`: side -> main (a buffer the side queue's launches read is
// about to be overwritten); `done`: the final join.
// Lifetime: the lane belongs to its context AND to one device — init(st) (re)creates stream and events on the device of the caller's
// stream when that differs from the one they were made on (a context that moves between GPUs); tdm_ctx_destroy (a thread's default
// context: the thread's exit) destroys them.  One lane per context: a context serves one thread, which drives one backward at a time.
struct TdmSideLane {
    hipStream_t side = nullptr;
    hipEvent_t ready[8] = {}, back[2] = {}, done = nullptr;
    hipEvent_t early = nullptr;      // "the early part of the flat gradient is final" (tdm_set_early_grads / tdm_unet_wait_early_grads)
    bool early_recorded = false;     // ... recorded by the calling thread's last backward
    hipEvent_t part[8] = {};         // denoiser: "layer l's gradient is final" (tdm_tt_wait_layer_grads)
    unsigned part_mask = 0;          // ... bit l: recorded by the last denoiser backward and not yet consumed
    bool ok = false;
    int device = -1;
    bool init(hipStream_t st);
    void destroy();
    ~TdmSideLane() { destroy(); }
};
// All selector state of the library lives in an explicit context object (tdm_hip.h: tdm_ctx_create / _destroy / _make_current): the
// arithmetic of the conv / GEMM / attention kernels, the two-queue and early-gradient switches, and the side queue with its events
// (created lazily by the first backward that forks, destroyed with the context).  A thread that never binds one works on its own
// default context (thread-exit destroys it).  There is no process-global mutable state; the only thread-local is the BINDING.
struct TdmCtx {
    int conv_mode = 2;     // 0 exact fp32 MFMA, 2 bf16x3 over S16 tensors (default); 1 was round 1's in-loader split (removed)
    int gemm_mode = 1;     // 0 fp32 MFMA, 1 bf16x3 (default), 2 plain bf16 operands
    int attn_mode = 2;     // 0 scalar fp32, 1 fp32 MFMA, 2 bf16x3 MFMA (default)
    int bwd_overlap = 1;   // weight-gradient launches on the context's side queue
    int early_grads = 0;   // gradients final in parts behind events (data parallel): the UNet in two, the denoiser per layer
    TdmSideLane lane;
    int bound = 0;         // 1 while some thread has it current (a context serves one thread at a time)
};
// the C ABI's context object (tdm_hip.h): the process's RCCL communicator (comm.hip; void* here: an ncclComm_t) + the selector state
struct tdm_ctx { int device = 0; int rank = 0, world = 1; void* comm = nullptr; TdmCtx c; };
TdmCtx& tdm_cur_ctx();
int tdm_ctx_unbind_for_destroy(tdm_ctx* ctx);   // ctx.hip: unbinds from the calling thread; fails if another thread has it current
inline TdmSideLane& tdm_side_lane() { return tdm_cur_ctx().lane; }
// Joins the side queue into the caller's stream on EVERY exit path of a backward that has forked — an error return between the
// first fork and the regular join included — so that "every effect of a call is ordered on the stream the caller passed"
// (tdm_hip.h) also holds for a failed call: the caller may free or reuse its buffers behind its own stream.
struct TdmSideJoin {
    TdmSideLane* ln;
    hipStream_t st;
    bool armed = false;
    int join() {   // the regular join; disarms the guard
        armed = false;
        TDM_HIP(hipEventRecord(ln->done, ln->side));
        TDM_HIP(hipStreamWaitEvent(st, ln->done, 0));
        return 0;
    }
    ~TdmSideJoin() {
        if (armed) { (void)hipEventRecord(ln->done, ln->side); (void)hipStreamWaitEvent(st, ln->done, 0); }
    }
};
// 1 if the calling thread's selector is on AND `st` is not being captured: a forked step replayed as a hipGraph is slower than
// the one-queue graph (ROCm's graph executor pays more per cross-branch edge than the overlap returns), so captures get one queue
int tdm_bwd_overlap(hipStream_t st);

#define TDM_TRY(expr)               \
    do {                            \
        int rc__ = (expr);          \
        if (rc__ != 0) return rc__; \
    } while (0)

// ---------------------------------------------------------------------------
// implicit-GEMM convolution on fp32 MFMA (conv_mfma.hip)
// ---------------------------------------------------------------------------
// One K-source of a convolution: a channel range of an NHWC tensor, optionally
// nearest-upsampled x2 (virtual upsample+concat of src/mnist.py:83-84) and
// optionally with a per-(sample, channel) bias added to in-image pixels (the
// timestep bias of src/mnist.py:58-59, applied while staging).
struct ConvSrc {
    const float* ptr;   // [B][H>>up][W>>up][C]
    const float* tb;    // [B][tb_stride] or nullptr
    const float* w;     // weight tensor of this source (HWIO)
    int C;              // channels in ptr
    int c0;             // first channel used
    int nch;            // channels used (multiple of 16)
    int up;             // 1: ptr is half resolution
    int taps;           // 9 (3x3, pad 1) or 1 (1x1)
    int tb_stride;
    int w_rows;         // rows per tap of w  (fwd: Cin_total; dgrad: Cin_total = N)
    int w_r0;           // fwd: first row of this source inside w
    int w_cols;         // columns of w (fwd: N; dgrad: Cout of the forward conv = K)
    // bf16x3 path (conv_pack.hip / conv_s16.hip): weights pre-packed in MFMA B-fragment order
    const unsigned short* wp;   // packed tensor for this direction (hi/lo planes)
    int wchunk0;                // first 16-channel K chunk of this source inside wp
};

struct ConvArgs {
    ConvSrc src[2];
    int nsrc;
    const float* bias;  // [N] or nullptr
    const float* res;   // [M][N] or nullptr, added after relu
    float* out;         // [M][N]
    float* aux;         // [M][N] or nullptr: value after relu, before residual
    int relu;
    int B;
    int ablate;         // diagnostics: bit0 skip input loads (s16: all prefetches after chunk 0), bit1 skip weight staging
                        // (bf16x3 kernel), bit2 skip MFMA, bit3 skip the epilogue (s16 kernel)
    // S16 pipeline (conv_s16.hip): sources are S16 tensors; optional extra pre-split output
    float* out_s16;           // [M][N] S16 copy of the result (+ tb_out), or nullptr
    const float* tb_out;      // [B][tb_out_stride]: per-(sample, channel) bias folded into out_s16 only
    int tb_out_stride;
    // [M][N/4] bytes or nullptr: bit e of byte (m, c/4) = (value after ReLU of channel c + e > 0).  What backward needs of
    // a post-ReLU tensor is its sign; a byte per 4 channels replaces a 16-byte fp32 quad in both directions.
    unsigned char* mask_out;
    // ReLU backward fused into a data-gradient launch (S16 kernel): when relu_mask_in != nullptr the result v of the
    // transposed conv is the gradient w.r.t. a post-ReLU tensor (+ time bias); `out` / `out_s16` then receive
    // v * (mask bit) and `sums` the per-32-pixel-group partial sums the time-bias / bias gradients need:
    // sums[((group*2 + slot)*2 + kind)*N + c], kind 0 = sum v (unmasked), 1 = sum of the masked value; slot 0 = pixels
    // of the image the group's first pixel belongs to, slot 1 = pixels of the next image (a group may straddle two).
    const unsigned char* relu_mask_in;
    float* sums;
    // ... and (optional) the S16 twin of v BEFORE the mask is applied: a tensor that is both the output gradient of a ReLU'd conv
    // (masked: out_s16) and, unmasked, the output gradient of the block's skip path (rb2: dout2) leaves one launch in both forms
    float* out_s16_pre;
    // Fused 1x1 "skip" conv of a residual block (src/mnist.py:52,61) over the SAME sources (N = 32 kernels, 3x3 sources):
    // a second accumulator fed by the centre-tap pixel fragments the 3x3 conv has already read.  skip_wp: packed 1x1
    // weights with the same chunk numbering as the sources' 3x3 weights; skip_out[M][N] = skip conv + skip_bias (fp32).
    const unsigned short* skip_wp;
    const float* skip_bias;
    float* skip_out;
    // Rank-1 residual (S16 kernel): rb1's skip conv has ONE input channel, so its output s[m][c] = fma(x[m], w[c], b[c]) is
    // recomputed in the epilogue from 4 bytes per pixel instead of being written and re-read as a [M][N] fp32 tensor
    // (51 MB each way at B = 512).  r1_x: [M] or nullptr; r1_w, r1_b: [N].  Added after ReLU, like `res` (exclusive with it).
    const float* r1_x;
    const float* r1_w;
    const float* r1_b;
    // rb4.conv1's data gradient (S16 kernel, 28x28, N = 96 = the concat's channels): when dc_pair != nullptr the epilogue
    //  * adds the skip path's share rk1_d[m] * rk1_u[c] (rank one: out_bwd_s16_kernel's header; rk1_d: [M], rk1_u: [96]),
    //  * writes channels 0..63 (the up-sampled h3 part) with horizontally adjacent pixels already ADDED — dc_pair is
    //    (B, 28, 14, 64): half of the 2x2 sum of the upsample backward, taken where the pair sits in one transpose block —
    //  * and channels 64..95 (the h1 part) to dc_h1, a (M, 32) tensor.  `out` is not written.
    const float* rk1_d;
    const float* rk1_u;
    float* dc_pair;
    float* dc_h1;
    // Fused 1x1 output conv N -> 1 (src/mnist.py:87, same kernel instantiation): o1_out[m] = sum_c value[m][c] * o1_w[c]
    // + o1_b[0] over the value that goes to `out` (after ReLU and residual), with conv_out_kernel's association, so the
    // [M][32] tensor need not be written when nothing else reads it (sampling).  o1_out: [M] or nullptr.
    const float* o1_w;
    const float* o1_b;
    float* o1_out;
    // F.mse_loss forward / backward at the source (training, src/mnist.py:158-159 on the output of :87): with o1_tgt != nullptr
    // (the noise target, [M]) the same epilogue also writes o1_deps[m] = (o1_out[m] - o1_tgt[m]) * o1_dscale = d loss / d eps
    // and, per 32-pixel group g, the partial row o1_sums[g * 40 + ...]: [0..31] = sum_m d[m] * value[m][c] (the output conv's
    // weight gradient), [32] = sum_m d[m] (its bias gradient), [33] = sum_m (eps - tgt)^2 — so the [M][32] tensor is not
    // written for the backward pass either (`out` may be nullptr).
    const float* o1_tgt;
    float* o1_deps;
    float* o1_sums;
    float o1_dscale;
    // rb4.conv1's forward in the PHASE form (S16 kernel; conv_s16.hip "PH"): src[0] is the half-resolution tensor the reference
    // up-samples x2 (src/mnist.py:83), given with up = 1 and taps = 16 — its packed weights are the 4 phases x 4 taps of
    // PackDesc::phase = 1 — and src[1] the full-resolution tensor with the ordinary nine taps.
    int up_phase;
    // The transpose (conv_s16.hip "S2D"): rb4.conv1's data gradient w.r.t. the up-sampled tensor, computed at ITS resolution.  hw = 14,
    // N = 64; src[0] = the 28x28 output gradient (S16, up = 0, taps = 4, packed weights of PackDesc::phase = 2); out = the 14x14
    // gradient (fp32); rk1_d / rk1_u (optional): + (rk1_d summed over the source pixel's four output pixels) * rk1_u[c].
    int s2d;
};

// hw in {28,14}; N in {32,64,96}; dgrad: transposed convolution with the forward weights
int tdm_launch_conv(const ConvArgs& a, int hw, int N, bool dgrad, hipStream_t st);

// weight pre-pack of the bf16x3 kernels (conv_pack.hip; direction-agnostic kernels: the pre-pack encodes fwd / dgrad)
// phase = 1: forward weights of an up-sampled source in the phase form (conv_s16.hip "PH"): rows [0, kuse) of the cin input channels,
// 16 packed taps per chunk, packed tap (2 py + px) * 4 + 2 a + b = sum of the 3x3 taps (ky, kx) that fall on source pixel
// (i + py - 1 + a, j + px - 1 + b) for output pixel (2 i + py, 2 j + px): ky in {0} / {1, 2} for (py, a) = (0, 0) / (0, 1),
// {0, 1} / {2} for (1, 0) / (1, 1); kx likewise (summed in fp32, then split hi / lo).
// phase = 2: the TRANSPOSED weights of the same source for its data gradient at source resolution (conv_s16.hip "S2D"): d h[i][j] is a
// 4x4, stride-2 gather over the output gradient g, g[2i-1+u][2j-1+v], with weights sum_{ky in T(u)} sum_{kx in T(v)} W[ky][kx]^T,
// T(0) = {2}, T(1) = {1, 2}, T(2) = {0, 1}, T(3) = {0}.  Packed per parity group (p, q) of g's rows / columns (g[2i'+p][2j'+q]):
// chunk (2p + q) * (cout / 16) + kc holds g channels 16 kc .., 4 taps each: tap 2 a + b with u = 2 dy + p + 1 for
// dy = a (p = 0) or a - 1 (p = 1), v likewise from (q, b); n = input channel < kuse, k = output-gradient channel.
// n0 / nuse (dgrad = 1, phase = 0): only output channels [n0, n0 + nuse) of the transposed conv (a sub-block of the concat).
struct PackDesc { int src_off, cin, cout, taps, dgrad; long dst_off; int phase, kuse, n0, nuse; };
#define TDM_MAX_PACK 24
struct PackArgs { PackDesc d[TDM_MAX_PACK]; int n; };
int tdm_launch_pack(const float* params, const PackArgs& pa, unsigned short* out, hipStream_t st);
int tdm_launch_pack_timebias(const float* params, const PackArgs& pa, unsigned short* out, const int64_t* t, const int* te_w_off,
                             const int* te_b_off, float* that, float* tb, int B, int64_t* bump, float* u96, int skw4_off,
                             int outw_off, hipStream_t st);

// S16 pipeline: pre-split sources, float4 epilogue (conv_s16.hip)
// pixel-count limit of its 32-bit addressing (__mul24 on the pixel index): B * H * W < 2^23, i.e. B <= 10,699 at 28x28
#define TDM_S16_MAX_PIXELS (1L << 23)
#define TDM_S16_MAX_BATCH 10699
int tdm_launch_conv_s16(const ConvArgs& a, int hw, int N, hipStream_t st);
int tdm_launch_to_s16(const float* in, const float* tb, int tb_stride, float* out, long M, int HWpix, int C,
                      hipStream_t st);

struct WgradArgs {
    ConvSrc a;          // activation source (nch = channels covered, multiple of 32)
    const float* g;     // [M][Cout] output gradient
    int Cout;
    float* slab;        // slab 0 base
    long slab_stride;   // floats between slabs
    int w_off;          // offset of the weight tensor inside a slab
    int b_off;          // offset of the bias gradient inside a slab, or -1
    int B;
    int ntiles;
    int nci;            // number of 32-channel ci tiles
    // S16 producer / consumer kernel only: the 1x1 skip conv of the same residual block reads the same activation tensor;
    // its weight gradient dWskip[ci][co] = sum_p A[p][ci] * G2[p][co] rides on the 3x3 launch (the centre-tap fragments are
    // staged anyway; tap group 1 has a free fifth accumulator).  g2: [M][Cout] S16 or nullptr; w_off2: slab offset of the
    // 1x1 weight tensor (rows a.w_r0 .. as for the 3x3 tensor).
    const float* g2;
    int w_off2;
    // rb4.conv1's up-sampled source in the parity form (conv_s16.hip: wgrad_s2d_kernel): a = the HALF-resolution activation (hw = 14,
    // up = 0), g = the full-resolution (28x28) output gradient; the nine 3x3 slots of the slab are written as usual.
    int s2d;
};
int tdm_launch_wgrad(const WgradArgs& a, int hw, int nslab, hipStream_t st);
int tdm_launch_wgrad_s16(const WgradArgs& a, int hw, int nslab, hipStream_t st);    // a.a.ptr and a.g are S16; no bias

// slab reduction: out[off+i] = sum_s slab[s*stride + off + i]
struct ReduceSec {
    int off, len, nslab;
    long src_delta;        // slab 0 of this section starts at slabs + off + src_delta (default 0)
    long stride_override;  // distance between this section's slabs; 0 = the launch's common stride
    float* dst;            // where the section's sums go; nullptr = out + off
    float scale;           // factor applied to the sums; 0 = 1
    // outer_w != nullptr: the section's sums r[i] are one factor of a rank-one gradient: element i writes the outer_n
    // values r[i] * outer_w[j] to (dst ? dst : out + off)[i * outer_n + j] (rb4.skip's weight gradient, out_bwd_s16_kernel)
    const float* outer_w;
    int outer_n;
    int vec4;              // set by tdm_launch_reduce: the section is walked with 16-byte loads (lengths / offsets / stride multiples of 4)
};
#define TDM_MAX_SECS 48
struct ReduceArgs {
    ReduceSec sec[TDM_MAX_SECS];
    int nsec;
    int blk0[TDM_MAX_SECS + 1];   // filled by tdm_launch_reduce: first workgroup of each section in the 1-D grid
};
int tdm_launch_reduce(const float* slabs, long stride, const ReduceArgs& ra, float* out, hipStream_t st);

// ---------------------------------------------------------------------------
// elementwise / small kernels (elementwise.hip)
// ---------------------------------------------------------------------------
// bump != nullptr: also bump[0] += 1 (the Philox offset of the fused train step, consumed by the preceding draw kernel)
// u96 != nullptr: also u96[ci] = sum_co params[skw4_off + ci * 32 + co] * params[outw_off + co] (96 values; ConvArgs::rk1_u)
int tdm_launch_timebias(const int64_t* t, const float* params, const int* te_w_off, const int* te_b_off,
                        float* that, float* tb, int B, hipStream_t st, int64_t* bump = nullptr, float* u96 = nullptr,
                        int skw4_off = 0, int outw_off = 0);
int tdm_launch_timebias_float(const float* that, const float* w, const float* bias, float* tb, int B, int C, hipStream_t st);
int tdm_launch_draw_q_sample(const float* x0, const float* sqrt_acp, const float* sqrt_1m_acp, uint64_t seed, int64_t* rng_state,
                             int64_t* t_out, float* noise_out, float* x_noisy_out, int64_t B, int64_t inner, bool bump,
                             hipStream_t st, const int64_t* perm = nullptr, const int64_t* steps = nullptr,
                             const int64_t* base = nullptr, int64_t n_rows = 0, int64_t stride = 0, int64_t offset = 0);
int tdm_launch_conv_first(const float* x, const float* w1, const float* b1, const float* ws, const float* bs,
                          float* a1, float* s, int B, hipStream_t st);
int tdm_launch_avgpool(const float* in, float* out, int B, int Hout, int C, hipStream_t st);
int tdm_launch_conv_out(const float* h, const float* w, const float* b, float* eps, int64_t M, hipStream_t st);
int tdm_launch_out_bwd(const float* deps, const float* h4, const float* w, const float* a2,
                       float* dout, float* dc2, float* slab, long slab_stride, int w_off, int b_off,
                       int64_t M, int nslab, hipStream_t st);
int tdm_launch_relu_mask(const float* dout, const float* a, float* dc, int64_t n, hipStream_t st);
int tdm_launch_relu_bwd_tb(float* dh, const float* a1, float* S, int B, int HWpix, int C, hipStream_t st);
// S[b][c] / S2[b][c] = per-image sums of the group partials written by a conv launch with relu_mask_in (ConvArgs::sums)
int tdm_launch_image_sums(const float* sums, float* S, float* S2, int B, int HWpix, int C, hipStream_t st);
int tdm_launch_time_grad(const float* S, const float* that, float* d_tw, float* d_tb, int B, int C, hipStream_t st);
int tdm_launch_time_grad_multi(const float* const* S, float* const* d_tw, float* const* d_tb, const int* C, int n,
                               const float* that, int B, hipStream_t st);
int tdm_launch_split_dcat(const float* dcat, float* dout3, int B, hipStream_t st);
int tdm_launch_combine_dh1(const float* dcat, const float* dp1, float* dout1, int B, hipStream_t st);
int tdm_launch_first_wgrad(const float* x, const float* dc1, const float* dout1, float* slab, long slab_stride,
                           int w1_off, int b1_off, int ws_off, int bs_off, int B, int nslab, hipStream_t st);
// S16-pipeline producers (optional extra outputs; nullptr = not written)
int tdm_launch_conv_first_s16(const float* x, const float* w1, const float* b1, const float* ws, const float* bs,
                              const float* tb, int tb_stride, float* a1, unsigned char* a1m, float* a1_s16, float* s, int B,
                              hipStream_t st);
// p1_s16 = split(avg_pool2d(h1, 2)), s2 = rb2.skip(p1) + bias (1x1, 32 -> 64, exact fp32 on the vector units)
int tdm_launch_pool_skip_s16(const float* h1, const float* wsk, const float* bsk, float* p1_s16, float* s2, int B,
                             hipStream_t st);
int tdm_launch_s16_to_nchw(const float* in_s16, float* out, int B, int HWpix, int C, hipStream_t st);   // accessor: hi + lo -> fp32 NCHW
int tdm_launch_avgpool_s16(const float* in, float* out, float* out_s16, int B, int Hout, int C, hipStream_t st);
// deps == nullptr: the MSE backward is fused — d = (eps - noise) * 2/M is computed here (and written to deps_out when
// given), and the slab partial of sum (eps - noise)^2 goes to slab offset loss_off (F.mse_loss, src/mnist.py:158).
// dout4 = d x w_out is rank one and never written: the kernel emits rb4.skip's gradients in factored form (partial rows:
// vsk_off: the 96-vector sum_m cat[m] d[m] over cat = [up2(h3s) | h1s]; skb_off: w_out * sum d) — elementwise.hip
int tdm_launch_skip4_factored(const float* deps, const float* h1, const float* h3, const float* w_out, float* slab, long slab_stride,
                              int skw_off, int skb_off, int B, int nslab, hipStream_t st);
int tdm_launch_out_bwd_s16(const float* deps, const float* h4, const float* w, const unsigned char* a2m, const float* h1s,
                           const float* h3s, float* dc2_s16, float* slab, long slab_stride, int w_off, int b_off,
                           int c2b_off, int skb_off, int vsk_off, int64_t M, int nslab, hipStream_t st,
                           const float* eps = nullptr, const float* noise = nullptr, float* deps_out = nullptr,
                           int loss_off = -1, const float* o1_sums = nullptr);
// time_emb weight / bias gradients of the four blocks and the conv1 bias gradients of rb2..rb4 as slab partials, straight
// from the per-32-pixel-group sums the data-gradient launches wrote (ConvArgs::sums; one buffer per block)
// jobs job0 .. job0 + njobs - 1 (njobs 0: to the 4th).  tew[i] < 0: no time-embedding outputs (job 4: the masked sums of a tensor
// that is NOT a conv1 pre-activation — rb3.conv2's output gradient, whose bias row the S2D data-gradient launch feeds)
// Jobs 4 / 5 (phase form of the backward): rb3.conv2's and rb2.conv2's output gradients; job 5 also emits the UNMASKED sum (ub[i] >= 0:
// rb2.skip's bias gradient).  Jobs 0..3: ub = -1 (their unmasked sum is the time-embedding bias row at tew + C).
#define TDM_GS_JOBS 6
struct GroupSumJobs { const float* gs[TDM_GS_JOBS]; int C[TDM_GS_JOBS]; int HWpix[TDM_GS_JOBS]; int tew[TDM_GS_JOBS]; int c1b[TDM_GS_JOBS]; int ub[TDM_GS_JOBS]; int job0; int njobs; };
int tdm_launch_group_sums(const GroupSumJobs& jb, const float* that, int B, float* slab, long slab_stride, int nslab,
                          hipStream_t st);
// dc_s16 = split(dout * (a > 0)); slab partial sums of the masked (and optionally unmasked) gradient per channel
int tdm_launch_split_dcat_mask_s16(const float* dcat, const unsigned char* a2m, float* dout3, float* dc_s16, float* slab,
                                   long slab_stride, int b_masked_off, int B, int nslab, hipStream_t st);
// x != nullptr: also the slab partials of rb1.skip's gradients (1 input channel: dW[c] = sum x[m] * dout1[m][c],
// db[c] = sum dout1[m][c]) at ws_off / bs_off, so dout1 itself (nullable) need not be written for first_wgrad
int tdm_launch_combine_dh1_mask_s16(const float* dcat, const float* dp1, const unsigned char* a2m, float* dout1, float* dc_s16,
                                    float* slab, long slab_stride, int b_masked_off, int B, int nslab, hipStream_t st,
                                    const float* x = nullptr, int ws_off = -1, int bs_off = -1);
int tdm_launch_relu_mask_s16(const float* dout, const unsigned char* am, float* dc_s16, float* slab, long slab_stride,
                             int b_masked_off, int b_unmasked_off, int64_t M, int C, int nslab, hipStream_t st);
// dh <- dh * (a1 > 0) in place (fp32) + S16 copy; S[b][c] = sum dh (unmasked), S2[b][c] = sum of the masked values
int tdm_launch_time_grad_multi2(const float* const* S, const float* const* S2, float* const* d_tw, float* const* d_tb,
                                float* const* d_b, const int* C, int n, const float* that, int B, hipStream_t st);
int tdm_launch_mask_io(unsigned char* packed, unsigned char* nchw, int B, int HWpix, int C, int write, hipStream_t st);
int tdm_launch_nhwc_to_nchw(const float* in, float* out, int B, int HWpix, int C, hipStream_t st);
