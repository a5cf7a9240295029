// Device-side randomness and device-side step counting for the DDPM loops, so that a whole
// train step / reverse step is ONE hipGraph-replayable launch sequence with no host-written
// scalars: the Philox offset, the reverse-step index and AdamW's step count all live in
// device memory and are advanced by the kernels that consume them (last-arriving workgroup).
//   * draw_q_sample_kernel: t ~ U{0..999}, noise ~ N(0,1), x_noisy = q_sample(x0, t, noise)
//     (src/mnist.py:154-156) — replaces randint + randn + q_sample (three launches, one
//     write + one read of the noise) by one pass;
//   * p_update_philox_kernel: x_{t-1} = p_sample update with z drawn in registers
//     (src/mnist.py:173-180), then t -= 1 — replaces randn + update + the t decrement;
//   * adamw_devstep_kernel: torch.optim.AdamW with the bias corrections derived on the
//     device from a device-resident step count (src/mnist.py:148,159).
// The arithmetic of q_sample / the update / AdamW is the same as in elementwise.hip
// (separately rounded mul / add, no FMA contraction).
#include "tdm_common.h"
#include "tdm_philox.h"
#include <math.h>

namespace {

constexpr int RB = 256;
inline int rng_grid(int64_t nwork) {
    int64_t g = (nwork + RB - 1) / RB;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

// State advances (Philox offset, reverse-step index, AdamW step count) are made by a one-workgroup kernel launched
// right AFTER the kernel that consumed the old values (stream order = every workgroup of the consumer has finished).
// The first version let the consumer's last-arriving workgroup do it (one same-address atomic per workgroup): 709 / 2048
// serialised atomics cost 17 us in AdamW and more in the B = 4096 update — three times the kernels' own time.
__global__ __launch_bounds__(64) void bump_offset_kernel(int64_t* __restrict__ state) {
    if (threadIdx.x == 0) state[0] += 1;
}
__global__ __launch_bounds__(RB) void bump_t_kernel(int64_t* __restrict__ state, int64_t* __restrict__ t, int64_t B) {
    for (int64_t b = threadIdx.x; b < B; b += RB) {
        const int64_t v = t[b] - 1;
        t[b] = v < 0 ? 0 : v;
    }
    if (threadIdx.x == 0) state[0] += 1;
}
__global__ __launch_bounds__(64) void bump_adam_kernel(int64_t* __restrict__ state, float beta1, float beta2) {
    if (threadIdx.x == 0) {
        const double b1p = (state[0] == 0 ? 1.0 : __longlong_as_double(state[2])) * (double)beta1;
        const double b2p = (state[0] == 0 ? 1.0 : __longlong_as_double(state[3])) * (double)beta2;
        state[0] += 1;
        state[2] = __double_as_longlong(b1p);
        state[3] = __double_as_longlong(b2p);
    }
}

__global__ __launch_bounds__(RB) void philox_normal_kernel(uint64_t seed, uint64_t offset, float* __restrict__ out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * RB)
        reinterpret_cast<float4*>(out)[i] = tdm_philox_normal4(seed, offset, (uint64_t)i);
}

__global__ __launch_bounds__(RB) void philox_u32_kernel(uint64_t seed, uint64_t offset, int kind, uint32_t* __restrict__ out,
                                                       int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * RB) {
        const tdm_u32x4 r = tdm_philox_at(seed, offset, (uint64_t)i, kind);
        reinterpret_cast<uint4*>(out)[i] = make_uint4(r.x, r.y, r.z, r.w);
    }
}

// state[0] = Philox offset of this call (advanced afterwards by bump_offset_kernel / the next kernel of the step)
// Batch source: x0 (B, inner) directly, or — EpochSrc::perm != nullptr — gathered on the fly from a device-resident dataset:
// image b of the step is row perm[(steps - base) * stride + offset + b] of `x0` (= the dataset), where steps = AdamW's
// device-side step count and base = its value when the epoch began: the train loop's per-step gather needs no launch and no
// host-written index (src/mnist.py:150-152's DataLoader batch; the positions are dp.shard_batch_indices').
struct EpochSrc { const int64_t* perm; const int64_t* steps; const int64_t* base; int64_t n, stride, offset; };
__global__ __launch_bounds__(RB) void draw_q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ ta,
                                                           const float* __restrict__ ts, uint64_t seed,
                                                           int64_t* __restrict__ state, int64_t* __restrict__ t_out,
                                                           float* __restrict__ noise_out, float* __restrict__ xn_out,
                                                           int64_t B, int64_t inner4, EpochSrc es) {
    const uint64_t offset = (uint64_t)state[0];
    const int64_t n4 = B * inner4;
    int64_t pos0 = 0;
    if (es.perm != nullptr) pos0 = (es.steps[0] - es.base[0]) * es.stride + es.offset;
    for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * RB) {
        const int64_t b = i / inner4;
        const int64_t tt = tdm_philox_step(seed, offset, (uint64_t)b, TDM_TIMESTEPS);
        if (i - b * inner4 == 0) t_out[b] = tt;
        const float a = ta[tt], s = ts[tt];
        int64_t src = i;
        if (es.perm != nullptr) {   // (clamped: a position or row outside the dataset is the host's bug, not a fault)
            const int64_t pos = min(max(pos0 + b, (int64_t)0), es.n - 1);
            const int64_t row = min(max(es.perm[pos], (int64_t)0), es.n - 1);
            src = row * inner4 + (i - b * inner4);
        }
        const float4 x = reinterpret_cast<const float4*>(x0)[src];
        const float4 n = tdm_philox_normal4(seed, offset, (uint64_t)i);
        float4 o;
        o.x = __fadd_rn(__fmul_rn(a, x.x), __fmul_rn(s, n.x));
        o.y = __fadd_rn(__fmul_rn(a, x.y), __fmul_rn(s, n.y));
        o.z = __fadd_rn(__fmul_rn(a, x.z), __fmul_rn(s, n.z));
        o.w = __fadd_rn(__fmul_rn(a, x.w), __fmul_rn(s, n.w));
        reinterpret_cast<float4*>(noise_out)[i] = n;
        reinterpret_cast<float4*>(xn_out)[i] = o;
    }
}

__device__ __forceinline__ float p_upd(float x, float e, float z, float cr, float ce, float cs) {
    return __fadd_rn(__fmul_rn(cr, __fsub_rn(x, __fmul_rn(ce, e))), __fmul_rn(cs, z));
}

// t: per-sample step index in DEVICE memory (decremented afterwards by bump_t_kernel); tsg[0] must be 0 so that the
// t == 0 step returns the mean (src/mnist.py:176-177) without a host-side branch
__global__ __launch_bounds__(RB) void p_update_philox_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                             const float* __restrict__ tr, const float* __restrict__ te,
                                                             const float* __restrict__ tsg, int64_t* __restrict__ t,
                                                             uint64_t seed, int64_t* __restrict__ state,
                                                             float* __restrict__ out, int64_t B, int64_t inner4) {
    const uint64_t offset = (uint64_t)state[0];
    const int64_t n4 = B * inner4;
    for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * RB) {
        const int64_t tt = t[i / inner4];
        const float cr = tr[tt], ce = te[tt], cs = tsg[tt];
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        const float4 ev = reinterpret_cast<const float4*>(eps)[i];
        const float4 zv = tdm_philox_normal4(seed, offset, (uint64_t)i);
        float4 o;
        o.x = p_upd(xv.x, ev.x, zv.x, cr, ce, cs);
        o.y = p_upd(xv.y, ev.y, zv.y, cr, ce, cs);
        o.z = p_upd(xv.z, ev.z, zv.z, cr, ce, cs);
        o.w = p_upd(xv.w, ev.w, zv.w, cr, ce, cs);
        reinterpret_cast<float4*>(out)[i] = o;
    }
}

// AdamW with the step count in device memory: state[0] = steps taken so far (this call performs step state[0] + 1
// and stores it), state[1] = arrival counter, state[2..3] = beta1^steps, beta2^steps (doubles).  Same update as adamw_kernel (elementwise.hip); the scalar prologue
// runs in double like torch's Python floats.
__global__ __launch_bounds__(RB) void adamw_devstep_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                           float lr, float beta1, float beta2, float eps, float wd,
                                                           int64_t* __restrict__ state, float gscale) {
    // beta^step as running products kept in the state (doubles, bit patterns in state[2..3]; 0 = "not started": 1.0):
    // a double pow() per workgroup was most of this kernel's time (22 us for 726 KB of parameters)
    const double b1p = (state[0] == 0 ? 1.0 : __longlong_as_double(state[2])) * (double)beta1;
    const double b2p = (state[0] == 0 ? 1.0 : __longlong_as_double(state[3])) * (double)beta2;
    const float step_size = (float)((double)lr / (1.0 - b1p)), bc2_sqrt = (float)sqrt(1.0 - b2p);
    const float decay = (float)(1.0 - (double)lr * (double)wd);
    const float one_m_b1 = (float)(1.0 - (double)beta1), one_m_b2 = (float)(1.0 - (double)beta2);
    for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += (int64_t)gridDim.x * RB) {
        const float gi = g[i] * gscale;
        float pi = p[i] * decay;
        float mi = m[i];
        mi = mi + one_m_b1 * (gi - mi);
        const float vi = v[i] * beta2 + one_m_b2 * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - step_size * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

// AdamW whose learning rate and gradient scale come from DEVICE memory, so that a hipGraph-captured step survives a per-step LR
// schedule (src/shakespeare.py:200-202, :250: LambdaLR stepped after every optimizer step) and a per-epoch loss weight
// (:216, :243): lr = lr_tab[min(steps taken, lr_n - 1)] — the host computes the whole schedule once with the reference's own
// Python arithmetic (lr_tab[i] = float(base_lr * lr_lambda(i)), what param_groups[0]['lr'] holds during step i + 1) — and the
// gradient is read as g * gscale * (gscale_dev ? *gscale_dev : 1).  Same update arithmetic as adamw_devstep_kernel.
__global__ __launch_bounds__(RB) void adamw_devsched_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                            const float* __restrict__ lr_tab, int lr_n, float beta1, float beta2,
                                                            float eps, float wd, const int64_t* __restrict__ state, float gscale,
                                                            const float* __restrict__ gscale_dev) {
    const int64_t taken = state[0];
    const float lr = lr_tab[taken < (int64_t)lr_n ? (int)taken : lr_n - 1];
    const float gs = gscale_dev != nullptr ? gscale * gscale_dev[0] : gscale;
    const double b1p = (taken == 0 ? 1.0 : __longlong_as_double(state[2])) * (double)beta1;
    const double b2p = (taken == 0 ? 1.0 : __longlong_as_double(state[3])) * (double)beta2;
    const float step_size = (float)((double)lr / (1.0 - b1p)), bc2_sqrt = (float)sqrt(1.0 - b2p);
    const float decay = (float)(1.0 - (double)lr * (double)wd);
    const float one_m_b1 = (float)(1.0 - (double)beta1), one_m_b2 = (float)(1.0 - (double)beta2);
    for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += (int64_t)gridDim.x * RB) {
        const float gi = g[i] * gs;
        float pi = p[i] * decay;
        float mi = m[i];
        mi = mi + one_m_b1 * (gi - mi);
        const float vi = v[i] * beta2 + one_m_b2 * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - step_size * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

// text train step glue (src/shakespeare.py:225-243 with learned embeddings):
//   dx0[b][i] = sqrt_acp[t[b]] * dxn[b][i] + rw * dxr[b][i]      (d total / d x0: the q_sample path + the rounding path)
__global__ __launch_bounds__(RB) void combine_dx0_kernel(const float* __restrict__ dxn, const int64_t* __restrict__ t,
                                                         const float* __restrict__ ta, const float* __restrict__ dxr,
                                                         const float* __restrict__ rw, float* __restrict__ out, int64_t B,
                                                         int64_t inner4) {
    const float w = rw[0];
    const int64_t n4 = B * inner4;
    for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * RB) {
        const float a = ta[t[i / inner4]];
        const float4 x = reinterpret_cast<const float4*>(dxn)[i], r = reinterpret_cast<const float4*>(dxr)[i];
        float4 o;
        o.x = __fadd_rn(__fmul_rn(a, x.x), __fmul_rn(w, r.x)); o.y = __fadd_rn(__fmul_rn(a, x.y), __fmul_rn(w, r.y));
        o.z = __fadd_rn(__fmul_rn(a, x.z), __fmul_rn(w, r.z)); o.w = __fadd_rn(__fmul_rn(a, x.w), __fmul_rn(w, r.w));
        reinterpret_cast<float4*>(out)[i] = o;
    }
}
//   losses[0..2] = {diff, rnd, diff + rw * rnd};  acc[0..3] += {diff, rnd, total, 1}   (the epoch's running sums, :252-255)
__global__ __launch_bounds__(64) void text_loss_kernel(const float* __restrict__ diff, const float* __restrict__ rnd,
                                                       const float* __restrict__ rw, float* __restrict__ losses,
                                                       float* __restrict__ acc) {
    if (threadIdx.x == 0) {
        const float d = diff[0], r = rnd[0], tot = __fadd_rn(d, __fmul_rn(rw[0], r));
        losses[0] = d; losses[1] = r; losses[2] = tot;
        acc[0] += d; acc[1] += r; acc[2] += tot; acc[3] += 1.f;
    }
}

}  // namespace

// internal: the fused train step advances the offset in its next kernel (timebias) instead of a bump launch
// (perm != nullptr: x0 is the whole dataset and the batch is gathered on the fly, see EpochSrc)
int tdm_launch_draw_q_sample(const float* x0, const float* sqrt_acp, const float* sqrt_1m_acp, uint64_t seed, int64_t* rng_state,
                             int64_t* t_out, float* noise_out, float* x_noisy_out, int64_t B, int64_t inner, bool bump,
                             hipStream_t st, const int64_t* perm, const int64_t* steps, const int64_t* base, int64_t n_rows,
                             int64_t stride, int64_t offset) {
    const EpochSrc es{perm, steps, base, n_rows, stride, offset};
    hipLaunchKernelGGL(draw_q_sample_kernel, dim3(rng_grid(B * inner / 4)), dim3(RB), 0, st, x0, sqrt_acp, sqrt_1m_acp, seed,
                       rng_state, t_out, noise_out, x_noisy_out, B, inner / 4, es);
    TDM_CHECK_LAUNCH("draw_q_sample");
    if (bump) {
        hipLaunchKernelGGL(bump_offset_kernel, dim3(1), dim3(64), 0, st, rng_state);
        TDM_CHECK_LAUNCH("bump_offset");
    }
    return 0;
}

extern "C" {

int tdm_philox_normal_f32(uint64_t seed, uint64_t offset, float* out, int64_t n, void* stream) {
    TDM_REQUIRE(out != nullptr && n > 0 && (n & 3) == 0, "philox_normal: n=%lld must be a positive multiple of 4", (long long)n);
    hipLaunchKernelGGL(philox_normal_kernel, dim3(rng_grid(n / 4)), dim3(RB), 0, (hipStream_t)stream, seed, offset, out, n / 4);
    TDM_CHECK_LAUNCH("philox_normal");
    return 0;
}

int tdm_philox_u32(uint64_t seed, uint64_t offset, int kind, uint32_t* out, int64_t n, void* stream) {
    TDM_REQUIRE(out != nullptr && n > 0 && (n & 3) == 0, "philox_u32: n=%lld must be a positive multiple of 4", (long long)n);
    TDM_REQUIRE(kind == 0 || kind == 1, "philox_u32: kind %d", kind);
    hipLaunchKernelGGL(philox_u32_kernel, dim3(rng_grid(n / 4)), dim3(RB), 0, (hipStream_t)stream, seed, offset, kind, out, n / 4);
    TDM_CHECK_LAUNCH("philox_u32");
    return 0;
}

int tdm_philox_u32_host(uint64_t seed, uint64_t offset, int kind, uint64_t idx, uint32_t* out4) {
    TDM_REQUIRE(out4 != nullptr && (kind == 0 || kind == 1), "philox_u32_host: bad arguments");
    const tdm_u32x4 r = tdm_philox_at(seed, offset, idx, kind);
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
    return 0;
}

int tdm_ddpm_draw_q_sample_f32(const float* x0, const float* sqrt_acp, const float* sqrt_1m_acp, uint64_t seed,
                               int64_t* rng_state, int64_t* t_out, float* noise_out, float* x_noisy_out, int64_t B,
                               int64_t inner, void* stream) {
    TDM_REQUIRE(x0 && sqrt_acp && sqrt_1m_acp && rng_state && t_out && noise_out && x_noisy_out, "draw_q_sample: NULL pointer");
    TDM_REQUIRE(B > 0 && inner > 0 && (inner & 3) == 0, "draw_q_sample: B=%lld inner=%lld (inner must be a multiple of 4)",
                (long long)B, (long long)inner);
    return tdm_launch_draw_q_sample(x0, sqrt_acp, sqrt_1m_acp, seed, rng_state, t_out, noise_out, x_noisy_out, B, inner, true,
                                    (hipStream_t)stream);
}

int tdm_p_sample_update_philox_f32(const float* x, const float* eps, const float* tab_recip, const float* tab_eps,
                                   const float* tab_sigma0, int64_t* t_dev, uint64_t seed, int64_t* rng_state, float* out,
                                   int64_t B, int64_t inner, void* stream) {
    TDM_REQUIRE(x && eps && tab_recip && tab_eps && tab_sigma0 && t_dev && rng_state && out, "p_sample_update_philox: NULL pointer");
    TDM_REQUIRE(B > 0 && inner > 0 && (inner & 3) == 0, "p_sample_update_philox: B=%lld inner=%lld (inner must be a multiple of 4)",
                (long long)B, (long long)inner);
    hipLaunchKernelGGL(p_update_philox_kernel, dim3(rng_grid(B * inner / 4)), dim3(RB), 0, (hipStream_t)stream, x, eps,
                       tab_recip, tab_eps, tab_sigma0, t_dev, seed, rng_state, out, B, inner / 4);
    TDM_CHECK_LAUNCH("p_sample_update_philox");
    hipLaunchKernelGGL(bump_t_kernel, dim3(1), dim3(RB), 0, (hipStream_t)stream, rng_state, t_dev, B);
    TDM_CHECK_LAUNCH("bump_t");
    return 0;
}

int tdm_adamw_flat_devstep_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int64_t* step_state, float grad_scale, void* stream) {
    TDM_REQUIRE(p && g && m && v && step_state && n > 0, "adamw_devstep: bad arguments (n=%lld)", (long long)n);
    hipLaunchKernelGGL(adamw_devstep_kernel, dim3(rng_grid(n)), dim3(RB), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                       beta2, eps, weight_decay, step_state, grad_scale);
    TDM_CHECK_LAUNCH("adamw_devstep");
    // (round 5 measured the step count advanced by the last-arriving workgroup of the kernel above instead of this launch: 709
    //  arrivals on one counter made the 5 us kernel a 22 us one - profiles/r05: reverted)
    hipLaunchKernelGGL(bump_adam_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_state, beta1, beta2);
    TDM_CHECK_LAUNCH("bump_adam");
    return 0;
}

int tdm_adamw_flat_devsched_f32(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_tab, int lr_n,
                                float beta1, float beta2, float eps, float weight_decay, int64_t* step_state, float grad_scale,
                                const float* grad_scale_dev, int bump, void* stream) {
    TDM_REQUIRE(p && g && m && v && step_state && lr_tab && n > 0 && lr_n >= 1, "adamw_devsched: bad arguments (n=%lld, lr_n=%d)",
                (long long)n, lr_n);
    hipLaunchKernelGGL(adamw_devsched_kernel, dim3(rng_grid(n)), dim3(RB), 0, (hipStream_t)stream, p, g, m, v, n, lr_tab, lr_n,
                       beta1, beta2, eps, weight_decay, step_state, grad_scale, grad_scale_dev);
    TDM_CHECK_LAUNCH("adamw_devsched");
    if (bump) {   // the LAST tensor of a step advances the shared step count (several tensors, one optimizer step)
        hipLaunchKernelGGL(bump_adam_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_state, beta1, beta2);
        TDM_CHECK_LAUNCH("bump_adam");
    }
    return 0;
}

int tdm_text_combine_dx0_f32(const float* dx_noisy, const int64_t* t, const float* sqrt_acp, const float* dx_round,
                             const float* rw_dev, float* out, int64_t B, int64_t inner, void* stream) {
    TDM_REQUIRE(dx_noisy && t && sqrt_acp && dx_round && rw_dev && out, "text_combine_dx0: NULL pointer");
    TDM_REQUIRE(B > 0 && inner > 0 && (inner & 3) == 0, "text_combine_dx0: B=%lld inner=%lld (inner must be a multiple of 4)",
                (long long)B, (long long)inner);
    hipLaunchKernelGGL(combine_dx0_kernel, dim3(rng_grid(B * inner / 4)), dim3(RB), 0, (hipStream_t)stream, dx_noisy, t, sqrt_acp,
                       dx_round, rw_dev, out, B, inner / 4);
    TDM_CHECK_LAUNCH("text_combine_dx0");
    return 0;
}

int tdm_text_loss_f32(const float* diff, const float* rnd, const float* rw_dev, float* losses3, float* acc4, void* stream) {
    TDM_REQUIRE(diff && rnd && rw_dev && losses3 && acc4, "text_loss: NULL pointer");
    hipLaunchKernelGGL(text_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, diff, rnd, rw_dev, losses3, acc4);
    TDM_CHECK_LAUNCH("text_loss");
    return 0;
}

}  // extern "C"
