/* libtdm_hip.so — C ABI of the MI355X-native DDPM hot path.
 *
 * The reference (LiamConnell/TinyDiffusionModels) has no FFI: its hot path is
 * plain Python over ATen ops.  Each entry point below replaces the ATen work
 * behind one reference function; the citation names the reference lines it
 * stands in for.  Python host code (the modules of tinydiffusionmodels_amd) re-creates
 * the reference's Python surface on top of these.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm
 *    allocates); the library never allocates, frees or synchronises device
 *    memory or the caller's stream, so every call is hipGraph-capturable; the
 *    only device-side objects it owns are, per CONTEXT (tdm_ctx below) and
 *    created by the first backward that forks, ONE side stream and a handful of
 *    events (tdm_set_bwd_overlap): work it puts there is forked from and joined
 *    back into the caller's stream by events inside the same call, so for the
 *    caller every effect of a call is ordered on the stream it passed — also
 *    when the call FAILS after a fork (the join runs on every exit path).  They
 *    live on the device of the stream the caller passed (rebuilt there if the
 *    context moves to another GPU) and are destroyed with the context
 *    (tdm_ctx_destroy; a thread's default context: when the thread exits);
 *  - selector state (arithmetic, launch overlap) lives in an explicit tdm_ctx;
 *    the library has no process-global mutable state, and its thread-local state
 *    is which context a thread has bound, the last-error text (tdm_last_error)
 *    and the profiling marks of tdm_unet_mark_launch;
 *  - `stream` is a hipStream_t passed as void*;
 *  - return 0 on success, non-zero on error; tdm_last_error() gives the text;
 *  - activations inside the library are NHWC fp32; the UNet input/output
 *    (C = 1) is identical in NCHW and NHWC;
 *  - UNet parameters live in ONE flat fp32 buffer in the layout given by
 *    tdm_unet_param_offsets(): conv weights are HWIO ([ky][kx][ci][co]),
 *    i.e. the reference's OIHW `state_dict` tensors permuted (2,3,1,0).
 */
#ifndef TDM_HIP_H
#define TDM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDM_VERSION 402
#define TDM_TIMESTEPS 1000
#define TDM_UNET_NPARAM 181473      /* SimpleUNet(), src/mnist.py:64-74 */
#define TDM_UNET_NTENSOR 32         /* number of state_dict entries      */
#define TDM_UNET_MAX_SLABS 512      /* wgrad partial slabs               */

int tdm_version(void);
const char* tdm_last_error(void);

/* ---- a2: q_sample  (src/mnist.py:36-42, src/shakespeare.py:37-44) -------
 * out[b,i] = sqrt_acp[t[b]] * x0[b,i] + sqrt_1m_acp[t[b]] * noise[b,i]
 * (mul, mul, add — no FMA contraction: bit-exact with the reference).      */
int tdm_q_sample_f32(const float* x0, const float* noise, const int64_t* t,
                     const float* sqrt_acp, const float* sqrt_1m_acp,
                     float* out, int64_t B, int64_t inner, void* stream);

/* gradient of q_sample w.r.t. x0: out[b,i] = tab[t[b]] * x[b,i] with tab = sqrt_acp (learned embeddings get
 * their diffusion-loss gradient through it, src/shakespeare.py:225-232)                                      */
int tdm_scale_by_table_f32(const float* x, const int64_t* t, const float* tab, float* out, int64_t B,
                           int64_t inner, void* stream);

/* ---- a6: p_sample arithmetic  (src/mnist.py:169-180, shakespeare.py:343-352)
 * out = c_recip * (x - c_eps * eps) [+ c_sigma * noise]; the three scalars are
 * read on the device from 1000-entry tables at index t_index (uniform-t fast
 * path of the reverse loops, src/mnist.py:191-193); noise == NULL -> t == 0
 * branch.  Op order as the reference: mul, sub, mul, (mul, add).            */
int tdm_p_sample_update_f32(const float* x, const float* eps, const float* noise,
                            const float* tab_recip, const float* tab_eps, const float* tab_sigma,
                            int t_index, float* out, int64_t n, void* stream);
/* general per-sample t (API parity with p_sample(model, x, t)): tables gathered by t[b] */
int tdm_p_sample_update_pert_f32(const float* x, const float* eps, const float* noise,
                                 const float* tab_recip, const float* tab_eps, const float* tab_sigma,
                                 const int64_t* t, int add_noise, float* out, int64_t B, int64_t inner,
                                 void* stream);

/* (x.clamp(-1,1)+1)/2 -> [0,1] (src/mnist.py:194) and the uint8 quantisation
 * save_image applies (mul 255, add 0.5, clamp, truncate).                    */
int tdm_to_unit_u8_f32(const float* x, float* x01, uint8_t* u8, int64_t n, void* stream);

/* ---- a3/a4: SimpleUNet  (src/mnist.py:45-87) ------------------------------ */
/* offsets (in floats) of the 32 parameter tensors inside the flat buffer, in
 * the reference's state_dict order (rb1.conv1.weight, rb1.conv1.bias, ...,
 * out.weight, out.bias); offs has TDM_UNET_NTENSOR+1 entries (last = total). */
int tdm_unet_param_offsets(int32_t* offs);

/* workspace size in floats for batch B (activations, saved tensors, backward
 * temporaries).  `training` != 0 also reserves the backward temporaries.     */
int64_t tdm_unet_workspace_floats(int64_t B, int training);
/* slab buffer (wgrad partials) size in floats                                */
int64_t tdm_unet_slab_floats(void);

/* eps = UNet(x, t).  x (B,1,28,28) fp32, t (B,) int64 raw step index,
 * eps (B,1,28,28).  save != 0 keeps what backward needs in `ws`.            */
int tdm_unet_fwd_f32(const float* params, const float* x, const int64_t* t, float* eps,
                     float* ws, int64_t B, int save, void* stream);

/* gradients of all parameters given d(loss)/d(eps); needs ws from a
 * save != 0 forward of the same (x, t).  grads: flat, same layout as params. */
int tdm_unet_bwd_f32(const float* params, const float* x, const float* deps, float* grads,
                     float* ws, float* slabs, int64_t B, void* stream);

/* copy one saved activation out of the workspace as NCHW (tests):
 * which: 0=h1 (32,28,28) 1=h2 (64,14,14) 2=h3 (64,14,14) 3=h4 (32,28,28)     */
int tdm_unet_get_activation(const float* ws, int64_t B, int which, float* out_nchw, void* stream);

/* ReLU sign masks kept by a save != 0 forward (default arithmetic: one bit per element instead of the
 * fp32 post-ReLU tensors), exchanged as one 0/1 byte per element, NCHW (B, C, H, W) of block 0..3's
 * conv1 (which = 1) / conv2 (which = 2) output.  write = 0 reads them; write != 0 installs masks
 * (tests: the reference's masks teacher-forced into tdm_unet_bwd_f32).                                */
int tdm_unet_relu_mask_io(float* ws, int64_t B, int block, int which, uint8_t* mask_nchw, int write, void* stream);

/* ---- a5: MSE + AdamW  (src/mnist.py:158, :148) ---------------------------- */
/* loss = mean((pred-target)^2) -> loss_out[0]; dpred = 2*(pred-target)/n.
 * scratch: >= 1024 floats.                                                   */
int tdm_mse_fwd_bwd_f32(const float* pred, const float* target, float* loss_out, float* dpred,
                        float* scratch, int64_t n, void* stream);

/* torch.optim.AdamW single-tensor update over a flat buffer; step is 1-based;
 * g is multiplied by grad_scale first (1/world after an all-reduce SUM).     */
int tdm_adamw_flat_f32(float* p, const float* g, float* m, float* v, int64_t n,
                       float lr, float beta1, float beta2, float eps, float weight_decay,
                       int64_t step, float grad_scale, void* stream);

/* whole fused train step on one GPU (q_sample -> fwd -> MSE -> bwd), leaving
 * the flat gradient in `grads` and the loss in loss_out[0]; the caller then
 * all-reduces `grads` (RCCL, via torch.distributed) and calls
 * tdm_adamw_flat_f32.  x_noisy: (B,784) scratch owned by the caller.        */
int tdm_unet_loss_grad_f32(const float* params, const float* x0, const float* noise, const int64_t* t,
                           const float* sqrt_acp, const float* sqrt_1m_acp,
                           float* x_noisy, float* eps, float* deps, float* loss_out, float* grads,
                           float* ws, float* slabs, int64_t B, void* stream);

/* one reverse step x <- p_sample(model, x, t=t_index) for the whole batch
 * (src/mnist.py:191-193): UNet forward + update; noise NULL at t_index==0.  */
int tdm_unet_p_sample_step_f32(const float* params, const float* x, const int64_t* t, const float* noise,
                               const float* tab_recip, const float* tab_eps, const float* tab_sigma,
                               int t_index, float* eps, float* x_out, float* ws, int64_t B, void* stream);

/* ---- device-side randomness and step counting (hipGraph-replayable loops) ----
 * The reference draws `t = torch.randint(0, timesteps, (B,))`, `noise = torch.randn_like(x0)`
 * (src/mnist.py:154-155) and `torch.randn_like(x)` (:178) from the host-seeded generator.  The
 * variants below draw the same distributions from Philox4x32-10 as a pure function of
 * (seed, stream offset, element index) — csrc/tdm_philox.h — inside the kernel that consumes
 * them.  rng_state: DEVICE int64[2] = {stream offset, reserved}; every call advances the offset
 * by one on the device, in a kernel enqueued behind the consumer (so a captured graph draws fresh
 * numbers on every replay).  The teacher-forced entry points above stay the parity path.       */
/* out[i] = N(0,1) draw i of stream (seed, offset); n % 4 == 0 (tests / utilities)             */
int tdm_philox_normal_f32(uint64_t seed, uint64_t offset, float* out, int64_t n, void* stream);
/* raw generator words, 4 per counter (kind 0 = noise stream, 1 = step-index stream), device / host */
int tdm_philox_u32(uint64_t seed, uint64_t offset, int kind, uint32_t* out, int64_t n, void* stream);
int tdm_philox_u32_host(uint64_t seed, uint64_t offset, int kind, uint64_t idx, uint32_t* out4);
/* t_out[b] ~ U{0..999}; noise_out ~ N(0,1); x_noisy_out = q_sample(x0, t, noise)  (src/mnist.py:154-156) */
int tdm_ddpm_draw_q_sample_f32(const float* x0, const float* sqrt_acp, const float* sqrt_1m_acp, uint64_t seed,
                               int64_t* rng_state, int64_t* t_out, float* noise_out, float* x_noisy_out,
                               int64_t B, int64_t inner, void* stream);
/* p_sample update (src/mnist.py:173-180) with z drawn in registers and the per-sample step index in
 * DEVICE memory: out = recip[t] * (x - ceps[t] * eps) + sigma0[t] * z, then t_dev[b] = max(t_dev[b] - 1, 0).
 * tab_sigma0 = sqrt(betas) with entry 0 set to 0 (the reference's `if t[0] == 0: return mean`).           */
int tdm_p_sample_update_philox_f32(const float* x, const float* eps, const float* tab_recip, const float* tab_eps,
                                   const float* tab_sigma0, int64_t* t_dev, uint64_t seed, int64_t* rng_state,
                                   float* out, int64_t B, int64_t inner, void* stream);
/* tdm_adamw_flat_f32 with the step count in DEVICE memory: step_state int64[4] = {steps taken, 0,
 * beta1^steps, beta2^steps (doubles, maintained by the kernel)}, all zero before the first step; the call
 * performs step steps_taken + 1 and stores it (src/mnist.py:148,159).  beta1 / beta2 must stay the same
 * for the life of a state.                                                                              */
int tdm_adamw_flat_devstep_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int64_t* step_state, float grad_scale,
                               void* stream);
/* tdm_unet_loss_grad_f32 with t and noise drawn on the device into t_buf (B) / noise (B,784)          */
int tdm_unet_loss_grad_philox_f32(const float* params, const float* x0, const float* sqrt_acp,
                                  const float* sqrt_1m_acp, uint64_t seed, int64_t* rng_state, int64_t* t_buf,
                                  float* noise, float* x_noisy, float* eps, float* deps, float* loss_out,
                                  float* grads, float* ws, float* slabs, int64_t B, void* stream);
/* ... with the BATCH gathered on the fly from a device-resident dataset (src/mnist.py:150-152): image b = row
 * perm[(step_state[0] - epoch_base[0]) * stride + offset + b] of data (n_rows, 784); step_state is AdamW's device-side
 * step count (tdm_adamw_flat_devstep_f32), epoch_base its value at the start of the epoch (one device write per epoch),
 * stride = batch x world, offset = rank x batch.  Whole batches only ((steps - base + 1) * stride <= n_rows).  A train loop
 * is then ONE hipGraph replay per batch: no gather launch, no host-written index.                                         */
int tdm_unet_loss_grad_philox_epoch_f32(const float* params, const float* data, const int64_t* perm, const int64_t* step_state,
                                        const int64_t* epoch_base, int64_t n_rows, int64_t stride, int64_t offset,
                                        const float* sqrt_acp, const float* sqrt_1m_acp, uint64_t seed, int64_t* rng_state,
                                        int64_t* t_buf, float* noise, float* x_noisy, float* eps, float* deps, float* loss_out,
                                        float* grads, float* ws, float* slabs, int64_t B, void* stream);
/* tdm_unet_p_sample_step_f32 with device-resident t and device-drawn noise: one reverse step of
 * src/mnist.py:190-193 with no host-written scalar                                                   */
int tdm_unet_p_sample_step_philox_f32(const float* params, const float* x, int64_t* t_dev, const float* tab_recip,
                                      const float* tab_eps, const float* tab_sigma0, uint64_t seed,
                                      int64_t* rng_state, float* eps, float* x_out, float* ws, int64_t B,
                                      void* stream);

/* ---- e: data-parallel collective (no reference counterpart: the reference trains
 *      on one device, deployment/configs/mnist-training.yaml:5-6) ---------------------------------
 * One process per GPU.  A context owns the RCCL communicator of its process; the unique id is made
 * on rank 0 (tdm_comm_unique_id) and handed to the other ranks by the host's bootstrap channel
 * (torchrun's store / any out-of-band byte transport).  Collectives are enqueued on the caller's
 * stream and never synchronise it.  RCCL itself is loaded at the first comm call (dlopen).          */
typedef struct tdm_ctx tdm_ctx;
int tdm_ctx_create(int device, tdm_ctx** out);
int tdm_ctx_destroy(tdm_ctx* ctx);
int tdm_comm_unique_id_bytes(void);                 /* sizeof(ncclUniqueId) = 128 */
int tdm_comm_unique_id(void* out_bytes);
int tdm_comm_init(tdm_ctx* ctx, const void* unique_id, int rank, int world);
int tdm_comm_rank(const tdm_ctx* ctx);
int tdm_comm_world(const tdm_ctx* ctx);
int tdm_comm_rccl_version(void);                    /* NCCL_VERSION_CODE of the loaded RCCL, -1 if unavailable */
/* buf <- sum over ranks of buf (in place): the ONE collective of a train step, 725,892 B for the UNet */
int tdm_allreduce_sum_f32(tdm_ctx* ctx, float* buf, int64_t n, void* stream);
/* buf <- root's buf (identical replicas before the first step)                                       */
int tdm_broadcast_f32(tdm_ctx* ctx, float* buf, int64_t n, int root, void* stream);

/* ---- profiling: one launch of the train step at a time ------------------------
 * The default train step's forward + backward are 33 launches; each has an id and a name.  After a full
 * tdm_unet_loss_grad_f32 call, tdm_unet_replay_launch_f32 re-issues launch `id` alone with the
 * arguments it had inside the step (bench.py times every launch with events; PMC per kernel).   */
int tdm_unet_launch_count(void);
const char* tdm_unet_launch_name(int id);
/* noise != NULL: the train step's launches (the MSE backward sits in the epilogue of the forward's last
 * launch and WRITES deps); NULL: those of tdm_unet_fwd_f32(save) + tdm_unet_bwd_f32 over the given deps. */
int tdm_unet_replay_launch_f32(const float* params, const float* x_noisy, const int64_t* t, float* eps,
                               float* deps, const float* noise, float* grads, float* ws, float* slabs,
                               int64_t B, int id, void* stream);
/* In-step timing: after tdm_unet_mark_launch(id, capacity) every EAGERLY issued train step (never during a stream capture)
 * records a HIP-event pair on the launch stream around launch `id`; tdm_unet_mark_collect waits for the pairs recorded
 * since the last collect, writes their elapsed microseconds (at most cap) and returns the count (-1 on error).
 * id < 0 switches the marks off and frees the events.  State is per host thread.                          */
int tdm_unet_mark_launch(int id, int capacity);
int tdm_unet_mark_collect(float* us_out, int cap);

/* ---- per-layer entry points (tests / profiling) --------------------------- */
/* generic NHWC 3x3 (pad 1) or 1x1 convolution as implicit GEMM on fp32 MFMA.
 * in: (B,H,W,Cin), w: HWIO, out: (B,H,W,Cout); flags: bit0 relu, bit1 dgrad
 * (w is the FORWARD HWIO weight [k][Cout_of_this_call][Cin_of_this_call];
 * the call computes the transposed convolution).  H=W in {28,14}.
 * bias/res/tb may be NULL.  tb: (B,Cin) added to in-image input pixels.     */
int tdm_conv_nhwc_f32(const float* in, const float* w, const float* bias, const float* res,
                      const float* tb, float* out, float* aux_relu_out,
                      int64_t B, int HW, int Cin, int Cout, int ksize, int flags, void* stream);
/* weight gradient of the same convolution: dw (HWIO) and db from in and dout;
 * slabs: scratch of nslab*(k*k*Cin*Cout + Cout) floats.                      */
int tdm_conv_wgrad_nhwc_f32(const float* in, const float* tb, const float* dout, float* dw, float* db,
                            float* slabs, int64_t B, int HW, int Cin, int Cout, int ksize, void* stream);

/* ---- a3: ResidualBlock(in_ch, out_ch).forward(x, t)  (src/mnist.py:45-61) as ONE call in the active conv arithmetic:
 *   h = relu(conv1(x)); h = h + time_emb(that)[b][c]; h = relu(conv2(h)); out = h + (skip(x) if skw else x)
 * x (B,HW,HW,Cin), out (B,HW,HW,Cout): NHWC fp32; c1w (3,3,Cin,Cout), c2w (3,3,Cout,Cout), skw (1,1,Cin,Cout): HWIO;
 * tew, teb (Cout): time_emb.weight[:, 0] / .bias; that (B): the float `t` of the reference's (B,1,1,1) argument.
 * skw == NULL: identity skip (Cin == Cout).  Geometries: HW in {14, 28}, Cin in {32, 64, 96}, Cout in {32, 64}, and the
 * first block's HW 28, Cin 1, Cout 32.  scratch: tdm_resblock_scratch_floats(B, HW, Cin, Cout) floats.                 */
int64_t tdm_resblock_scratch_floats(int64_t B, int HW, int Cin, int Cout);
int tdm_resblock_fwd_f32(const float* x, const float* that, const float* c1w, const float* c1b, const float* c2w,
                         const float* c2b, const float* tew, const float* teb, const float* skw, const float* skb,
                         float* out, float* scratch, int64_t B, int HW, int Cin, int Cout, void* stream);

/* Contexts: selector state (replaces: no reference counterpart — the reference is single-threaded PyTorch, src/mnist.py:129-166).
 * A tdm_ctx owns every piece of state a call consults besides its arguments: the arithmetic of the convolutions / linear
 * layers / attention, the launch-overlap switches, and the side stream + events of the two-queue backward.  Entry points use
 * the calling thread's CURRENT context: the one bound with tdm_ctx_make_current, else a default context the thread owns
 * (default arithmetic; destroyed at thread exit).  A context is current on at most one thread at a time (make_current on a
 * second thread fails); two threads with two contexts share nothing.  tdm_set_conv_mode / _gemm_mode / _attn_mode /
 * _bwd_overlap / _early_grads below are shorthands that set the field of the CURRENT context.
 *   tdm_ctx_create        (section e above; the same object also owns the process's RCCL communicator) a context in the default
 *                         arithmetic (conv 2, gemm 1, attention 2), overlap on, early gradients off
 *   tdm_ctx_make_current  bind to the calling thread (NULL: back to the thread's default); tdm_ctx_current: the bound one or NULL
 *   tdm_ctx_destroy       frees it and the side stream / events it created; unbinds it from the calling thread first; fails if
 *                         another thread has it current.  Every backward joins its side queue before it returns, so a context
 *                         is idle between calls and may be destroyed as soon as the caller's stream work is enqueued.       */
int tdm_ctx_make_current(tdm_ctx* ctx);
tdm_ctx* tdm_ctx_current(void);
int tdm_ctx_set_arithmetic(tdm_ctx* ctx, int conv_mode, int gemm_mode, int attn_mode);
int tdm_ctx_get_arithmetic(const tdm_ctx* ctx, int* conv_mode, int* gemm_mode, int* attn_mode);
int tdm_ctx_set_overlap(tdm_ctx* ctx, int bwd_overlap, int early_grads);

/* Arithmetic of the UNet's MFMA convolutions (forward, data and weight gradient):
 *   0  exact fp32 (v_mfma_f32_32x32x2_f32, bitwise an fp32 fmaf chain)
 *   2  bf16x3 split operands on v_mfma_f32_32x32x16_bf16 (hi*hi + hi*lo + lo*hi,
 *      fp32 accumulate; ~1e-5 relative, 3/16 of the fp32 MFMA cycles) over
 *      pre-split "S16" tensors (bf16 hi/lo per 16-channel group, same
 *      4 B/element) that the producing kernels write — loaders are plain
 *      16-byte copies — default
 *   (1, the same arithmetic with fp32 tensors split while staging, was superseded
 *    by 2 in round 1 and is no longer built: tdm_set_conv_mode(1) fails)
 * The three arithmetic selectors (conv / gemm / attention mode) are fields of the calling thread's CURRENT context (above):
 * they configure the calls that thread makes afterwards; every context starts in the default arithmetic.            */
int tdm_set_conv_mode(int mode);
int tdm_get_conv_mode(void);
/* Launch overlap inside the backward passes (a field of the current context like the selectors above; default 1).  UNet: the eight weight-gradient
 * launches (transformer, up to 16,384 tokens per batch: the weight-gradient GEMMs of every layer) are issued on a side stream the context owns (created on first use), each behind an event
 * recorded after the launch that produced its gradient operand; the data-gradient chain continues on the caller's stream and
 * waits for the side stream before the final slab reduction.  Same kernels on the same buffers: results are bit-identical
 * to 0 (everything on the caller's stream in program order).  A call whose stream is being CAPTURED always takes one queue
 * (the forked step replays slower as a hipGraph than the plain one).                                              */
int tdm_set_bwd_overlap(int on);
int tdm_get_bwd_overlap(void);
/* Data-parallel training (a field of the current context, default 0; none in the reference, deployment/configs/mnist-training.yaml:5-6 is one GPU):
 * with 1 the default-arithmetic backward finishes the flat gradient in TWO parts.  Floats [tdm_unet_early_grad_offset(),
 * TDM_UNET_NPARAM) — every tensor of rb2, rb3, rb4 and the output conv, 95 % of the bytes — are final as soon as rb2's weight-gradient
 * launches have retired, and an event marks that point; rb1's 9,760 floats follow with the last launch.
 * tdm_unet_wait_early_grads(stream) makes `stream` (the caller's collective stream) wait for that event of the calling thread's LAST
 * backward call (once per call), so the all-reduce of the early part runs under the rest of the backward.  Returns 1 if an event was recorded,
 * 0 if not (selector off, or the call was being captured: order the collective behind the call's stream as usual), < 0 on error.
 * Same kernels, same fixed-order sums: the gradient is bit-identical to the one-part form.                            */
int tdm_set_early_grads(int on);
int tdm_get_early_grads(void);
int64_t tdm_unet_early_grad_offset(void);
int tdm_unet_wait_early_grads(void* stream);
/* The same for the transformer denoiser (src/shakespeare.py:105-120 backward; bf16 GEMM modes, eager issue): with early gradients on,
 * tdm_tt_loss_grad_* / tdm_tt_bwd_f32 reduce layer l's slabs right behind that layer's weight-gradient launches — last layer first —
 * and record an event per layer.  Floats [begin, end) of the flat gradient (tdm_tt_layer_grad_range: a layer's twelve tensors are
 * contiguous) are final at layer l's event; tdm_tt_wait_layer_grads(stream, l) orders the caller's collective stream behind it
 * (1 = ordered, 0 = no such event: order behind the call's stream, < 0 error).  The time embedding's 2 D floats at the end of the
 * vector are final with the call's stream.  Same slabs, same fixed summation order: the gradient is bit-identical.         */
int tdm_tt_wait_layer_grads(void* stream, int layer);
int tdm_tt_layer_grad_range(int D, int depth, int ffn, int layer, int64_t* begin, int64_t* end);
/* per-layer entry points of the S16 pipeline (tests / profiling): the fp32 input is
 * pre-split into scratch first.  conv: scratch >= k*k*Cin*Cout + B*HW*HW*Cin + 64 floats;
 * out_s16 (optional, S16 layout: every 16-channel group = 16 bf16 hi then 16 bf16 lo)
 * receives split(result + tb_out[b][c]).  wgrad: dw only;
 * scratch >= B*HW*HW*(Cin+Cout) + 65*k*k*Cin*Cout + 128 floats.                         */
int tdm_conv_nhwc_s16_f32(const float* in, const float* w, const float* bias, const float* res,
                          const float* tb, float* out, float* aux_relu_out, float* out_s16,
                          const float* tb_out, float* scratch, int64_t B, int HW, int Cin, int Cout,
                          int ksize, int flags, void* stream);
int tdm_conv_wgrad_nhwc_s16_f32(const float* in, const float* tb, const float* dout, float* dw,
                                float* scratch, int64_t B, int HW, int Cin, int Cout, int ksize,
                                void* stream);

/* ---- a8-a10: TinyTransformer embedding-space denoiser (src/shakespeare.py:105-120) ----
 * x (B,L,D) fp32, t (B,) int64; post-LN encoder layers (packed in_proj, H heads,
 * ReLU FFN of width ffn, LayerNorm eps 1e-5), no mask, no positional encoding.
 * p_drop = 0: eval mode.  p_drop > 0: train mode of the reference (model.train(),
 * src/shakespeare.py:210) with its 1 + 4*depth dropout sites; the masks are a pure
 * function of (seed, site, element index) — tdm_dropout_keep_u8 — regenerated in
 * registers by forward and backward, so a backward call must receive the p_drop and
 * seed of its forward.  Parameters: ONE flat fp32 buffer in the reference's
 * state_dict order and native layouts (tdm_tt_param_offsets: 12 tensors per
 * layer, then time_emb.weight, time_emb.bias; last entry = total).  params, x, out and
 * ws must be 64-byte aligned (the bf16 GEMM modes keep pre-split copies of the GEMM
 * operands — "S16", see tdm_split_s16_f32 — in ws when D and ffn are multiples of 16). */
int64_t tdm_tt_param_count(int D, int depth, int ffn);
int tdm_tt_param_offsets(int D, int depth, int ffn, int64_t* offs);
int64_t tdm_tt_workspace_floats(int64_t B, int L, int D, int H, int depth, int ffn, int training);
int64_t tdm_tt_slab_floats(int D, int depth, int ffn);
/* out = TinyTransformer(x, t)  (src/shakespeare.py:115-120) */
int tdm_tt_fwd_f32(const float* params, const float* x, const int64_t* t, float* out, float* ws,
                   int64_t B, int L, int D, int H, int depth, int ffn, int save, float p_drop,
                   uint64_t seed, void* stream);
/* parameter gradients given d(loss)/d(out); needs ws of a save != 0 forward;
 * dx (B,L,D), may be NULL: d(loss)/d(x) (needed by learned embeddings,
 * src/shakespeare.py:225-226)                                                 */
int tdm_tt_bwd_f32(const float* params, const float* dout, float* grads, float* dx, float* ws, float* slabs,
                   int64_t B, int L, int D, int H, int depth, int ffn, float p_drop, uint64_t seed,
                   void* stream);
/* denoiser part of the text train step (src/shakespeare.py:230-236): q_sample ->
 * forward -> MSE -> backward; flat gradient in grads, loss in loss_out[0]     */
int tdm_tt_loss_grad_f32(const float* params, const float* x0, const float* noise, const int64_t* t,
                         const float* sqrt_acp, const float* sqrt_1m_acp, float* x_noisy, float* pred,
                         float* dpred, float* loss_out, float* grads, float* ws, float* slabs,
                         int64_t B, int L, int D, int H, int depth, int ffn, float p_drop, uint64_t seed,
                         void* stream);
/* tdm_tt_loss_grad_f32 with its randomness on the device: t ~ U{0..999} and noise ~ N(0,1) from the Philox stream (seed,
 * rng_state[0]) into t_buf (B) / noise (B,L,D) — the draws of src/shakespeare.py:228-229 — and this step's dropout masks =
 * masks(drop_seed) with every site key XORed with the low 32 bits of the advanced stream offset (rng_state[0] after the call
 * = before + 1): fresh draws and fresh masks on every hipGraph replay, no host-written scalar.  L * D % 4 == 0.            */
int tdm_tt_loss_grad_philox_f32(const float* params, const float* x0, const float* sqrt_acp, const float* sqrt_1m_acp,
                                uint64_t seed, int64_t* rng_state, int64_t* t_buf, float* noise, float* x_noisy,
                                float* pred, float* dpred, float* loss_out, float* grads, float* ws, float* slabs,
                                int64_t B, int L, int D, int H, int depth, int ffn, float p_drop, uint64_t drop_seed,
                                void* stream);
/* one reverse step of src/shakespeare.py:382-385 / :343-352 for a uniform t_index */
int tdm_tt_p_sample_step_f32(const float* params, const float* x, const int64_t* t, const float* noise,
                             const float* tab_recip, const float* tab_eps, const float* tab_sigma,
                             int t_index, float* eps, float* x_out, float* ws,
                             int64_t B, int L, int D, int H, int depth, int ffn, void* stream);
/* tdm_tt_p_sample_step_f32 with device-resident t and device-drawn noise (see tdm_unet_p_sample_step_philox_f32) */
int tdm_tt_p_sample_step_philox_f32(const float* params, const float* x, int64_t* t_dev, const float* tab_recip,
                                    const float* tab_eps, const float* tab_sigma0, uint64_t seed, int64_t* rng_state,
                                    float* eps, float* x_out, float* ws, int64_t B, int L, int D, int H, int depth,
                                    int ffn, void* stream);
/* Arithmetic of the transformer's linear layers and their gradients:
 *   0 exact fp32 MFMA; 1 bf16x3 split operands, fp32 accumulate (default; ~1e-5 rel,
 *   meets the 1e-3 parity bound); 2 plain bf16 operands (throughput mode, ~3e-3 rel) */
int tdm_set_gemm_mode(int mode);
int tdm_get_gemm_mode(void);
/* Attention kernels: 0 scalar fp32 (one thread per row), 1 fp32 MFMA, 2 bf16x3 MFMA (default:
 * split bf16 operands, fp32 accumulate, ~1e-5 rel like gemm mode 1).  Modes 0 and 1 are exact
 * fp32 arithmetic, the cross-checks for mode 2.                                           */
int tdm_set_attn_mode(int mode);
int tdm_get_attn_mode(void);
/* Per-op self-attention core of nn.TransformerEncoderLayer (reference: src/shakespeare.py:108-111, the
 * nn.MultiheadAttention inside the encoder layer) on the packed projection qkv[B][L][3 D] = q | k | v with
 * heads as consecutive D/H slices: o[B][L][D] = softmax(q k^T / sqrt(D/H)) (dropout on the probabilities
 * when p_drop > 0, mask site `site` of tdm_dropout_keep_u8) v;  lse[B H][L] = log-sum-exp of every score
 * row (saved for the backward).  D/H in {8, 16, 32, 64}; runs in the selected attention mode.            */
int tdm_attention_fwd_f32(const float* qkv, float* o, float* lse, int64_t B, int L, int D, int H, float p_drop,
                          uint64_t seed, int site, void* stream);
/* dqkv[B][L][3 D] = gradient of the above wrt qkv given dO[B][L][D] (every element written); Dvec[B H][L]
 * is scratch.  Same p_drop / seed / site as the forward call.                                          */
int tdm_attention_bwd_f32(const float* qkv, const float* o, const float* lse, const float* dO, float* dqkv,
                          float* Dvec, int64_t B, int L, int D, int H, float p_drop, uint64_t seed, int site,
                          void* stream);
/* One attention kernel in the form the train step launches it (S16 twin outputs): which = 0 forward (out = O, out16 = its
 * S16 twin or NULL, aux = lse), 1 dQ (aux = Dvec, written), 2 dK/dV (aux = Dvec, read); backward with out16 != NULL writes
 * d(qkv) as S16 only (out may be NULL).  Per-kernel timing / parity (tools/time_attn.py). */
int tdm_attention_step_form_f32(int which, const float* qkv, const float* o, const float* lse, const float* dO, float* out,
                                float* out16, float* aux, int64_t B, int L, int D, int H, float p_drop, uint64_t seed, int site,
                                void* stream);
/* Per-op residual LayerNorm of the post-LN encoder layer (norm1 / norm2 of nn.TransformerEncoderLayer,
 * src/shakespeare.py:108-111): s = x + r (r may be NULL); y = (s - mean) * rstd * gamma + beta over the last dim
 * (biased variance, eps 1e-5).  x, r, y, s: (M, D) fp32; mean, rstd: (M).  s / mean / rstd are optional (all or none):
 * what the backward twin reads.  D % 4 == 0, D <= 1024.                                                            */
int tdm_layernorm_residual_fwd_f32(const float* x, const float* r, const float* gamma, const float* beta, float* y,
                                   float* s, float* mean, float* rstd, int64_t M, int D, void* stream);
/* ds (M, D) = d(loss)/d(x) = d(loss)/d(r); dgamma_dbeta (2, D) = (dgamma, dbeta); deterministic (fixed-order partials).
 * scratch: tdm_layernorm_scratch_floats(D) floats.                                                                 */
int64_t tdm_layernorm_scratch_floats(int D);
int tdm_layernorm_residual_bwd_f32(const float* dy, const float* s, const float* mean, const float* rstd,
                                   const float* gamma, float* ds, float* dgamma_dbeta, float* scratch, int64_t M, int D,
                                   void* stream);
/* Host-side evaluation of the dropout mask (tests, oracle cross-check): keep_host[i] = 1 iff
 * flat element idx0 + i of dropout site `site` survives.  Sites in the order the reference's
 * forward reaches them: 0 input dropout (src/shakespeare.py:119); layer l: 1+4l attention
 * probabilities (B,H,L,L), 2+4l dropout1 (B,L,D), 3+4l FFN dropout (B,L,ffn), 4+4l dropout2. */
int tdm_dropout_keep_u8(float p_drop, uint64_t seed, int site, int64_t idx0, int64_t n, uint8_t* keep_host);
/* the same with the site key salted, key' = hash32(key ^ salt * 0x9E3779B9) (what tdm_tt_loss_grad_philox_f32's kernels
 * apply: salt = low word of the advanced Philox offset; any salt value, 0 included, gives a salted key)             */
int tdm_dropout_keep_salted_u8(float p_drop, uint64_t seed, uint32_t salt, int site, int64_t idx0, int64_t n,
                               uint8_t* keep_host);
/* Fused feed-forward chain of one encoder layer (src/shakespeare.py:108-111: linear1 -> ReLU -> dropout -> linear2) and of
 * its data gradient, hidden tile in registers (csrc/ffn_chain.hip).  All matrix operands are S16 (tdm_split_s16_f32):
 * x16 (M,256), wa16 (F,256), wb16 (256,F); y (M,256) fp32.  D must be 256, F a multiple of 32 (<= 2048).
 *   mode 0: y = relu(x wa^T + bias_a) wb^T + bias_b                              (inference: no dropout)
 *   mode 1: y = drop_out(drop_mid(relu(x wa^T + bias_a)) wb^T + bias_b)         (dropout sites site_mid / site_out of `seed`),
 *           and mid16 (M,F) S16 = the hidden activation, mask = its sign bits (tdm_ffn_chain_mask_count 32-bit words)
 *   mode 2: mid16 = (x wa^T) where the mask bit is set, times gate_scale, else 0;  y = mid wb^T   (biases / dropout unused)
 * nprod 3 = bf16x3 split operands (parity arithmetic), 1 = plain bf16.                                              */
int tdm_ffn_chain_f32(int mode, int nprod, const float* x16, const float* wa16, const float* bias_a, const float* wb16,
                      const float* bias_b, float* y, float* mid16, uint32_t* mask, float gate_scale, float p_drop,
                      uint64_t seed, int site_mid, int site_out, int64_t M, int D, int F, void* stream);
int64_t tdm_ffn_chain_mask_count(int64_t M, int F);
int tdm_attn_set_ablate(int bits);        /* same, attention kernels (tools/time_attn.py --ablate) */
int tdm_ffn_chain_set_ablate(int bits);   /* timing diagnostics of tools/time_ffn.py (results are wrong when nonzero) */
/* ---- the FULL text train step as one replayable launch sequence (src/shakespeare.py:221-250) ----
 * tdm_tt_loss_grad_philox_f32 with dx_noisy (B,L,D; nullable) = d loss / d x_noisy: what learned embeddings receive
 * through q_sample (:225-233).                                                                                      */
int tdm_tt_loss_grad_philox_dx_f32(const float* params, const float* x0, const float* sqrt_acp, const float* sqrt_1m_acp,
                                   uint64_t seed, int64_t* rng_state, int64_t* t_buf, float* noise, float* x_noisy, float* pred,
                                   float* dpred, float* loss_out, float* grads, float* dx_noisy, float* ws, float* slabs,
                                   int64_t B, int L, int D, int H, int depth, int ffn, float p_drop, uint64_t drop_seed,
                                   void* stream);
/* AdamW with the learning rate and (optionally) a gradient factor in DEVICE memory, so that a captured step survives the
 * reference's per-step LambdaLR (:200-202, :250) and per-epoch rounding weight (:216, :243):
 * lr = lr_tab[min(steps taken, lr_n - 1)], gradient = g * grad_scale * (grad_scale_dev ? *grad_scale_dev : 1).  step_state as
 * tdm_adamw_flat_devstep_f32; several tensors of one optimizer step share it: pass bump = 1 for the last one only.   */
int tdm_adamw_flat_devsched_f32(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_tab, int lr_n,
                                float beta1, float beta2, float eps, float weight_decay, int64_t* step_state, float grad_scale,
                                const float* grad_scale_dev, int bump, void* stream);
/* out = sqrt_acp[t[b]] * dx_noisy + rw * dx_round over (B, inner): d total / d x0 of a learned embedding (:225-243)  */
int tdm_text_combine_dx0_f32(const float* dx_noisy, const int64_t* t, const float* sqrt_acp, const float* dx_round,
                             const float* rw_dev, float* out, int64_t B, int64_t inner, void* stream);
/* losses3 = {diff, rnd, diff + rw * rnd};  acc4 += {diff, rnd, total, 1}  (the epoch's running sums, :252-255)       */
int tdm_text_loss_f32(const float* diff, const float* rnd, const float* rw_dev, float* losses3, float* acc4, void* stream);
/* ---- N1: learned embedding table and rounding head of the text train step
 *      (src/shakespeare.py:46-102 modules, :225-243 train step, :387-390 decode) ----
 * table (V,D) fp32, ids (M,) int64 token ids in [0,V), W (V,D) / b (V,) = LearnedRounding.decoder.  */
/* out[m] = table[ids[m]]   (LearnedEmbedding.forward, src/shakespeare.py:67) */
int tdm_embed_gather_f32(const float* table, const int64_t* ids, float* out, int64_t M, int V, int D,
                         void* stream);
/* dtable[ids[m]] += scale * g[m]   (its gradient; dtable must be zeroed / hold the running sum;
 * float atomics: summation order over repeated ids is not fixed, like torch's GPU embedding backward) */
int tdm_embed_scatter_add_f32(const float* g, const int64_t* ids, float* dtable, int64_t M, int V, int D,
                              float scale, void* stream);
int64_t tdm_round_workspace_floats(int64_t M, int V, int D);
/* (The (M, V) logits live once in ws — written by the logits GEMM, whose epilogue also emits the log-sum-exp
 *  partials, and read by the two gradient GEMMs, which regenerate softmax - onehot in their loaders; the
 *  gradient tensor itself is never stored.)
 * rounding loss of src/shakespeare.py:239-240 and its gradients in one call:
 *   loss_out[0] = cross_entropy(x W^T + b, ids)  (mean over the M tokens, unweighted)
 *   dx (M,D; may be NULL), dW (V,D), db (V) = gradients of  grad_scale * loss  (grad_scale = the
 *   rounding weight of :243); all three are overwritten.  ws: tdm_round_workspace_floats(M,V,D).  */
int tdm_round_ce_loss_grad_f32(const float* x, const float* W, const float* b, const int64_t* ids,
                               float grad_scale, float* loss_out, float* dx, float* dW, float* db,
                               float* ws, int64_t M, int V, int D, void* stream);
/* The same loss and gradients WITHOUT ever holding the (M, V) logits (src/shakespeare.py:239-240 at vocabulary
 * sizes where M x V floats do not fit): one statistics pass (the logits GEMM with its log-sum-exp epilogue and no
 * store), then per chunk of Vc vocabulary entries (a multiple of 128) the chunk's logits are recomputed into a
 * (M, Vc) scratch and consumed by the two gradient GEMMs.  One GEMM pass more, V / Vc times less workspace:
 * ws = tdm_round_workspace_chunked_floats(M, V, D, Vc).                                                      */
int64_t tdm_round_workspace_chunked_floats(int64_t M, int V, int D, int Vc);
int tdm_round_ce_loss_grad_chunked_f32(const float* x, const float* W, const float* b, const int64_t* ids,
                                       float grad_scale, float* loss_out, float* dx, float* dW, float* db,
                                       float* ws, int64_t M, int V, int D, int Vc, void* stream);
/* The same contract with the logits in REGISTERS only (csrc/ce_chain.hip: two chained-MFMA passes — token-stationary online
 * softmax + dX, vocabulary-stationary dW / db; no logits tensor, no per-chunk scratch, no split-K slabs).  D must be 256
 * (tdm_round_fused_ok).  nseg (1..8): token segments of the weight-gradient pass.  ws: tdm_round_workspace_fused_floats.  */
int tdm_round_fused_ok(int64_t M, int V, int D);
int64_t tdm_round_workspace_fused_floats(int64_t M, int V, int D, int nseg);
int tdm_round_ce_loss_grad_fused_f32(const float* x, const float* W, const float* b, const int64_t* ids, float grad_scale,
                                     float* loss_out, float* dx, float* dW, float* db, float* ws, int64_t M, int V, int D,
                                     int nseg, void* stream);
/* logits (M, ld >= V, ld % 4 == 0) = x W^T + b   (LearnedRounding.forward, src/shakespeare.py:101) */
int tdm_round_logits_f32(const float* x, const float* W, const float* b, float* logits, int64_t ld,
                         int64_t M, int V, int D, void* stream);
/* out_ids[m] = argmax_v (x W^T + b)[m][v]   (decode, src/shakespeare.py:389-390) */
int tdm_round_argmax_f32(const float* x, const float* W, const float* b, int64_t* out_ids, float* ws,
                         int64_t M, int V, int D, void* stream);
/* out_ids[m] = argmax_v cos(x[m], E[v])  — the cosine-similarity fallback decode (src/shakespeare.py:393-401:
 * F.normalize both sides, matmul, argmax).  E (V,D): embedding matrix.  ws: tdm_round_workspace_floats(M,V,D). */
int tdm_cosine_argmax_f32(const float* x, const float* E, int64_t* out_ids, float* ws, int64_t M, int V, int D,
                          void* stream);
/* "S16" tensors (the pre-split operand form of the bf16x3 kernels): same shape and byte size as the fp32
 * tensor; every 64-byte group of 16 consecutive elements holds hi[16] | lo[16] as bf16 with hi = bf16(x),
 * lo = bf16(x - hi).  out = S16 of in, n % 16 == 0 elements, 64-byte aligned.                           */
int tdm_split_s16_f32(const float* in, float* out, int64_t n, void* stream);
/* general strided GEMM of the transformer's linear layers in the selected gemm mode (tests / profiling):
 * C[i][j] = sum_k A[i*a_rs + k*a_cs] * B[k*b_rs + j*b_cs] (+bias[j]) (+res[i][j]) (relu)
 * `relu` is a flag word: bit 0 ReLU; bit 1 (bf16 modes): A and B are S16 tensors (K-contiguous form: K % 16
 * == 0; token-major form: M, N % 16 == 0); bit 2 (bf16 modes, K-contiguous form): C is written as an S16
 * tensor (N % 16 == 0); bit 4 (bf16 modes, token-major form with S16 operands, M and N >= 256): the
 * 256 x 256-tile LDS-DMA kernel the denoiser's weight gradients run on (gemm_tn_ring.hip).             */
int tdm_gemm_f32(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs,
                 float* C, int64_t c_rs, const float* bias, const float* res, int M, int N, int K,
                 int relu, int splitk, int64_t c_split_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TDM_HIP_H */
