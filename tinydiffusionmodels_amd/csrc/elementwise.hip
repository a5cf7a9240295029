// HBM-bound elementwise / reduction kernels of the DDPM path: q_sample,
// p_sample update, timestep bias, the Cin=1 first convolution, average pool,
// the 32->1 output convolution, fused MSE forward+backward, ReLU backward
// with the timestep-bias gradient, AdamW.  All float4-coalesced, 64-lane
// wavefront reductions, deterministic (no float atomics).
#include <cstdlib>
#include "tdm_common.h"
#include "tdm_timebias.h"
#include "tdm_s16.h"

namespace {

constexpr int EW_BLOCK = 256;
inline int ew_grid(int64_t nwork) {
    int64_t g = (nwork + EW_BLOCK - 1) / EW_BLOCK;
    if (g > 2048) g = 2048;  // grid-stride the rest
    if (g < 1) g = 1;
    return (int)g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// block-wide sum of one float per thread (256 threads); result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* sh /* >= 4 floats */) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ---- q_sample -------------------------------------------------------------
// src/mnist.py:36-42: out = A[t[b]]*x0 + S[t[b]]*noise, as mul, mul, add
// (__fmul_rn/__fadd_rn are never contracted into an FMA).
__global__ __launch_bounds__(EW_BLOCK) void q_sample_kernel(const float* __restrict__ x0,
                                                            const float* __restrict__ noise,
                                                            const int64_t* __restrict__ t,
                                                            const float* __restrict__ ta,
                                                            const float* __restrict__ ts, float* __restrict__ out,
                                                            int64_t B, int64_t inner) {
    const int64_t total = B * inner;
    if ((inner & 3) == 0) {
        const int64_t n4 = total >> 2, inner4 = inner >> 2;
        for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_BLOCK) {
            const int64_t b = i / inner4;
            const int64_t tt = t[b];
            const float a = ta[tt], s = ts[tt];
            const float4 x = reinterpret_cast<const float4*>(x0)[i];
            const float4 n = reinterpret_cast<const float4*>(noise)[i];
            float4 o;
            o.x = __fadd_rn(__fmul_rn(a, x.x), __fmul_rn(s, n.x));
            o.y = __fadd_rn(__fmul_rn(a, x.y), __fmul_rn(s, n.y));
            o.z = __fadd_rn(__fmul_rn(a, x.z), __fmul_rn(s, n.z));
            o.w = __fadd_rn(__fmul_rn(a, x.w), __fmul_rn(s, n.w));
            reinterpret_cast<float4*>(out)[i] = o;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
            const int64_t tt = t[i / inner];
            out[i] = __fadd_rn(__fmul_rn(ta[tt], x0[i]), __fmul_rn(ts[tt], noise[i]));
        }
    }
}

// out[b,i] = tab[t[b]] * x[b,i]: the gradient of q_sample w.r.t. x0 (d x_noisy / d x0 = sqrt_acp[t], src/shakespeare.py:41-44
// differentiated) — learned embeddings receive their diffusion-loss gradient through it
__global__ __launch_bounds__(EW_BLOCK) void scale_by_table_kernel(const float* __restrict__ x, const int64_t* __restrict__ t,
                                                                  const float* __restrict__ tab, float* __restrict__ out,
                                                                  int64_t B, int64_t inner) {
    const int64_t total = B * inner;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK)
        out[i] = __fmul_rn(tab[t[i / inner]], x[i]);
}

// ---- p_sample update --------------------------------------------------------
// src/mnist.py:173-180: mean = c_recip * (x - c_eps * eps); out = mean + c_sigma * z
__device__ __forceinline__ float p_update(float x, float e, float z, float cr, float ce, float cs, bool add) {
    const float mean = __fmul_rn(cr, __fsub_rn(x, __fmul_rn(ce, e)));
    return add ? __fadd_rn(mean, __fmul_rn(cs, z)) : mean;
}

__global__ __launch_bounds__(EW_BLOCK) void p_update_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                            const float* __restrict__ noise,
                                                            const float* __restrict__ tr, const float* __restrict__ te,
                                                            const float* __restrict__ tsg, const int64_t* __restrict__ t,
                                                            int t_index, int add_noise, float* __restrict__ out,
                                                            int64_t B, int64_t inner) {
    // t == nullptr: uniform step t_index for the whole batch
    const int64_t total = B * inner;
    const bool add = add_noise != 0 && noise != nullptr;
    if ((inner & 3) == 0) {
        const int64_t n4 = total >> 2, inner4 = inner >> 2;
        for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_BLOCK) {
            const int64_t tt = (t != nullptr) ? t[i / inner4] : (int64_t)t_index;
            const float cr = tr[tt], ce = te[tt], cs = tsg[tt];
            const float4 xv = reinterpret_cast<const float4*>(x)[i];
            const float4 ev = reinterpret_cast<const float4*>(eps)[i];
            float4 zv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (add) zv = reinterpret_cast<const float4*>(noise)[i];
            float4 o;
            o.x = p_update(xv.x, ev.x, zv.x, cr, ce, cs, add);
            o.y = p_update(xv.y, ev.y, zv.y, cr, ce, cs, add);
            o.z = p_update(xv.z, ev.z, zv.z, cr, ce, cs, add);
            o.w = p_update(xv.w, ev.w, zv.w, cr, ce, cs, add);
            reinterpret_cast<float4*>(out)[i] = o;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
            const int64_t tt = (t != nullptr) ? t[i / inner] : (int64_t)t_index;
            out[i] = p_update(x[i], eps[i], add ? noise[i] : 0.f, tr[tt], te[tt], tsg[tt], add);
        }
    }
}

// (clamp(x,-1,1)+1)/2 and uint8 = trunc(clamp(x01*255 + 0.5, 0, 255))
__global__ __launch_bounds__(EW_BLOCK) void to_unit_u8_kernel(const float* __restrict__ x, float* __restrict__ x01,
                                                              uint8_t* __restrict__ u8, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_BLOCK) {
        float v = x[i];
        v = fminf(fmaxf(v, -1.f), 1.f);
        v = __fdiv_rn(__fadd_rn(v, 1.f), 2.f);
        if (x01 != nullptr) x01[i] = v;
        if (u8 != nullptr) {
            float q = __fadd_rn(__fmul_rn(v, 255.f), 0.5f);
            q = fminf(fmaxf(q, 0.f), 255.f);
            u8[i] = (uint8_t)q;
        }
    }
}

// ---- timestep bias ----------------------------------------------------------
// src/mnist.py:77 and :58: that = t.float()/1000; tb[b][c] = w[c]*that + bias[c]
// for the four blocks (32+64+64+32 = 192 channels per sample).
__global__ __launch_bounds__(EW_BLOCK) void timebias_kernel(TimebiasArgs a) {
    static_assert(EW_BLOCK == 256, "tdm_timebias_body assumes 256 threads");
    tdm_timebias_body(a, blockIdx.x, gridDim.x);
}

// tb[b][c] = fma(w[c], that[b], bias[c]) for ONE residual block whose caller already holds t-hat as floats
// (ResidualBlock.forward(x, t), src/mnist.py:56-59: `self.time_emb(t)` on a (B,1,1,1) float tensor)
__global__ __launch_bounds__(EW_BLOCK) void timebias_float_kernel(const float* __restrict__ that, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ tb,
                                                                  int B, int C) {
    const int total = B * C;
    for (int i = blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += gridDim.x * EW_BLOCK) {
        const int b = i / C, c = i - b * C;
        tb[i] = fmaf(w[c], that[b], bias[c]);
    }
}

// ---- rb1.conv1 (Cin = 1) + rb1.skip ----------------------------------------
// src/mnist.py:57 with in_ch=1 and :52: a1 = relu(conv3x3(x)+b1), s = x*ws+bs.
// HBM-bound (4 B in, 256 B out per pixel): 8 lanes per pixel, float4 stores.
__global__ __launch_bounds__(EW_BLOCK) void conv_first_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ w1,
                                                              const float* __restrict__ b1,
                                                              const float* __restrict__ ws,
                                                              const float* __restrict__ bs, float* __restrict__ a1,
                                                              float* __restrict__ s, int B) {
    const int64_t total = (int64_t)B * 784 * 8;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c4 = (int)(i & 7);
        const int64_t m = i >> 3;
        const int b = (int)(m / 784);
        const int rem = (int)(m - (int64_t)b * 784);
        const int y = rem / 28, xx = rem - y * 28;
        float4 acc = *reinterpret_cast<const float4*>(b1 + c4 * 4);
        float xc = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xq = xx + tap % 3 - 1;
            float xv = 0.f;
            if (yy >= 0 && yy < 28 && xq >= 0 && xq < 28) xv = x[(int64_t)b * 784 + yy * 28 + xq];
            if (tap == 4) xc = xv;
            const float4 wv = *reinterpret_cast<const float4*>(w1 + tap * 32 + c4 * 4);
            acc.x = fmaf(xv, wv.x, acc.x); acc.y = fmaf(xv, wv.y, acc.y);
            acc.z = fmaf(xv, wv.z, acc.z); acc.w = fmaf(xv, wv.w, acc.w);
        }
        acc.x = acc.x < 0.f ? 0.f : acc.x; acc.y = acc.y < 0.f ? 0.f : acc.y;
        acc.z = acc.z < 0.f ? 0.f : acc.z; acc.w = acc.w < 0.f ? 0.f : acc.w;
        *reinterpret_cast<float4*>(a1 + m * 32 + c4 * 4) = acc;
        const float4 wsv = *reinterpret_cast<const float4*>(ws + c4 * 4);
        const float4 bsv = *reinterpret_cast<const float4*>(bs + c4 * 4);
        float4 sv;
        sv.x = fmaf(xc, wsv.x, bsv.x); sv.y = fmaf(xc, wsv.y, bsv.y);
        sv.z = fmaf(xc, wsv.z, bsv.z); sv.w = fmaf(xc, wsv.w, bsv.w);
        *reinterpret_cast<float4*>(s + m * 32 + c4 * 4) = sv;
    }
}

// ---- F.avg_pool2d(h, 2), NHWC  (src/mnist.py:80) ----------------------------
__global__ __launch_bounds__(EW_BLOCK) void avgpool_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int B, int Ho, int C) {
    const int C4 = C >> 2, Wo = Ho, Wi = 2 * Ho;
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c4 = (int)(i % C4);
        const int64_t p = i / C4;
        const int xo = (int)(p % Wo);
        const int64_t q = p / Wo;
        const int yo = (int)(q % Ho);
        const int64_t b = q / Ho;
        const float4* src = reinterpret_cast<const float4*>(in) + ((b * Wi + 2 * yo) * Wi + 2 * xo) * C4 + c4;
        const float4 v00 = src[0], v01 = src[C4], v10 = src[(int64_t)Wi * C4], v11 = src[(int64_t)Wi * C4 + C4];
        float4 o;
        o.x = (((v00.x + v01.x) + v10.x) + v11.x) * 0.25f;
        o.y = (((v00.y + v01.y) + v10.y) + v11.y) * 0.25f;
        o.z = (((v00.z + v01.z) + v10.z) + v11.z) * 0.25f;
        o.w = (((v00.w + v01.w) + v10.w) + v11.w) * 0.25f;
        reinterpret_cast<float4*>(out)[i] = o;
    }
}

// ---- out = Conv2d(32, 1, 1)  (src/mnist.py:74,87) ----------------------------
// 8 lanes per pixel (one float4 each), reduced with 3 xor-shuffles.
__global__ __launch_bounds__(EW_BLOCK) void conv_out_kernel(const float* __restrict__ h, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ eps,
                                                            int64_t M) {
    const int64_t total = M * 8;
    const int c4 = threadIdx.x & 7;
    const float4 wv = *reinterpret_cast<const float4*>(w + c4 * 4);
    const float bias = b[0];
    // all 8 lanes of a pixel run the same iterations (EW_BLOCK and the stride are multiples of 8)
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < ((total + 7) & ~(int64_t)7);
         i += (int64_t)gridDim.x * EW_BLOCK) {
        float v = 0.f;
        if (i < total) {
            const float4 hv = reinterpret_cast<const float4*>(h)[i];
            v = ((hv.x * wv.x + hv.y * wv.y) + hv.z * wv.z) + hv.w * wv.w;
        }
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        if (c4 == 0 && i < total) eps[i >> 3] = v + bias;
    }
}

// ---- F.mse_loss forward + backward  (src/mnist.py:158) ------------------------
__global__ __launch_bounds__(EW_BLOCK) void mse_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                       float* __restrict__ dpred, float* __restrict__ partial,
                                                       int64_t n, float scale) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_BLOCK) {
        const float d = pred[i] - tgt[i];
        acc += d * d;
        dpred[i] = d * scale;
    }
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(EW_BLOCK) void mse_final_kernel(const float* __restrict__ partial, int nparts,
                                                             float* __restrict__ loss, float inv_n) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nparts; i += EW_BLOCK) acc += partial[i];
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) loss[0] = s * inv_n;
}

// ---- backward of the output conv + ReLU mask of rb4.conv2 ---------------------
// dout[p][c] = deps[p]*w[c]; dc2 = dout * (a2 > 0);
// slab partials: dw[c] = sum_p deps[p]*h4[p][c], db = sum_p deps[p]
__global__ __launch_bounds__(EW_BLOCK) void out_bwd_kernel(const float* __restrict__ deps, const float* __restrict__ h4,
                                                           const float* __restrict__ w, const float* __restrict__ a2,
                                                           float* __restrict__ dout, float* __restrict__ dc2,
                                                           float* __restrict__ slab, long slab_stride, int w_off,
                                                           int b_off, int64_t M) {
    __shared__ float4 shw[EW_BLOCK];
    __shared__ float shb[4];
    const int c4 = threadIdx.x & 7;
    const float4 wv = *reinterpret_cast<const float4*>(w + c4 * 4);
    float4 gw = make_float4(0.f, 0.f, 0.f, 0.f);
    float gb = 0.f;
    const int64_t total = M * 8;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const float d = deps[i >> 3];
        const float4 hv = reinterpret_cast<const float4*>(h4)[i];
        const float4 av = reinterpret_cast<const float4*>(a2)[i];
        float4 o;
        o.x = d * wv.x; o.y = d * wv.y; o.z = d * wv.z; o.w = d * wv.w;
        if (dout != nullptr) reinterpret_cast<float4*>(dout)[i] = o;   // (nullptr: nobody reads the rank-one gradient itself)
        float4 mk;
        mk.x = av.x > 0.f ? o.x : 0.f; mk.y = av.y > 0.f ? o.y : 0.f;
        mk.z = av.z > 0.f ? o.z : 0.f; mk.w = av.w > 0.f ? o.w : 0.f;
        reinterpret_cast<float4*>(dc2)[i] = mk;
        gw.x += d * hv.x; gw.y += d * hv.y; gw.z += d * hv.z; gw.w += d * hv.w;
        if (c4 == 0) gb += d;
    }
    shw[threadIdx.x] = gw;
    const float sb = block_sum(gb, shb);  // contains __syncthreads
    if (threadIdx.x < 8) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = threadIdx.x; k < EW_BLOCK; k += 8) {
            const float4 v = shw[k];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        float* dst = slab + (long)blockIdx.x * slab_stride;
        *reinterpret_cast<float4*>(dst + w_off + threadIdx.x * 4) = s;
        if (threadIdx.x == 0) dst[b_off] = sb;
    }
}

// rb4.skip's weight / bias gradient of the exact-fp32 pipeline, factored (the S16 pipeline's out_bwd_s16_kernel does the same from
// the S16 twins).  The gradient of rb4's output is rank one over the channels, dout4[p][co] = deps[p] * w_out[co]
// (src/mnist.py:87: the output conv is 32 -> 1), so the 1 x 1 skip conv's gradient (src/mnist.py:52,61 backward) is
//   dW[ci][co] = sum_p cat[p][ci] dout4[p][co] = v[ci] w_out[co],  v[ci] = sum_p cat[p][ci] deps[p],   db[co] = (sum_p deps[p]) w_out[co]
// with cat = [up2(h3) (64 channels), h1 (32)]: a 96-vector reduction over the pixels instead of two token-major MFMA launches
// (101 + 57 us at B = 512 on the fp32 matrix cores).  For the up-sampled channels the sum runs over the 14 x 14 SOURCE pixels with
// deps summed over each pixel's four outputs.  Per-workgroup partials -> slab[block][skw_off + ci * 32 + co], [skb_off + co].
__global__ __launch_bounds__(EW_BLOCK) void skip4_factored_kernel(const float* __restrict__ deps, const float* __restrict__ h1,
                                                                  const float* __restrict__ h3, const float* __restrict__ w_out,
                                                                  float* __restrict__ slab, long slab_stride, int skw_off,
                                                                  int skb_off, int B) {
    __shared__ float4 sh[EW_BLOCK];
    __shared__ float shb[4];
    __shared__ float vs[96];
    __shared__ float sd_s;
    const int64_t S = (int64_t)gridDim.x * EW_BLOCK;
    // h1: 8 channel quads per 28 x 28 pixel
    float4 g1 = make_float4(0.f, 0.f, 0.f, 0.f);
    float gd = 0.f;
    const int64_t n1 = (int64_t)B * 784 * 8;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n1; i += S) {
        const float d = deps[i >> 3];
        const float4 hv = reinterpret_cast<const float4*>(h1)[i];
        g1.x += d * hv.x; g1.y += d * hv.y; g1.z += d * hv.z; g1.w += d * hv.w;
        if ((threadIdx.x & 7) == 0) gd += d;
    }
    // h3: 16 channel quads per 14 x 14 source pixel, deps summed over its 2 x 2 outputs
    float4 g3 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t n3 = (int64_t)B * 196 * 16;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n3; i += S) {
        const int s = (int)(i >> 4);
        const int b = s / 196, r = s - b * 196, y = r / 14, x = r - y * 14;
        const float* dp = deps + (long)b * 784 + (2 * y) * 28 + 2 * x;
        const float d = (dp[0] + dp[1]) + (dp[28] + dp[29]);
        const float4 hv = reinterpret_cast<const float4*>(h3)[i];
        g3.x += d * hv.x; g3.y += d * hv.y; g3.z += d * hv.z; g3.w += d * hv.w;
    }
    // block sums per channel quad, fixed order: thread & 7 (h1) / thread & 15 (h3) is the quad (the grid stride is a multiple of 16)
    sh[threadIdx.x] = g1;
    const float sb = block_sum(gd, shb);   // (contains __syncthreads)
    if (threadIdx.x < 8) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = threadIdx.x; k < EW_BLOCK; k += 8) { const float4 v = sh[k]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        vs[64 + threadIdx.x * 4 + 0] = a.x; vs[64 + threadIdx.x * 4 + 1] = a.y; vs[64 + threadIdx.x * 4 + 2] = a.z; vs[64 + threadIdx.x * 4 + 3] = a.w;
    }
    if (threadIdx.x == 0) sd_s = sb;
    __syncthreads();
    sh[threadIdx.x] = g3;
    __syncthreads();
    if (threadIdx.x < 16) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = threadIdx.x; k < EW_BLOCK; k += 16) { const float4 v = sh[k]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        vs[threadIdx.x * 4 + 0] = a.x; vs[threadIdx.x * 4 + 1] = a.y; vs[threadIdx.x * 4 + 2] = a.z; vs[threadIdx.x * 4 + 3] = a.w;
    }
    __syncthreads();
    float* dst = slab + (long)blockIdx.x * slab_stride;
    for (int e = threadIdx.x; e < 96 * 32; e += EW_BLOCK) dst[skw_off + e] = vs[e >> 5] * w_out[e & 31];
    if (threadIdx.x < 32) dst[skb_off + threadIdx.x] = sd_s * w_out[threadIdx.x];
}

// dc = dout * (a > 0)
__global__ __launch_bounds__(EW_BLOCK) void relu_mask_kernel(const float* __restrict__ dout, const float* __restrict__ a,
                                                             float* __restrict__ dc, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_BLOCK) {
        const float4 d = reinterpret_cast<const float4*>(dout)[i];
        const float4 av = reinterpret_cast<const float4*>(a)[i];
        float4 o;
        o.x = av.x > 0.f ? d.x : 0.f; o.y = av.y > 0.f ? d.y : 0.f;
        o.z = av.z > 0.f ? d.z : 0.f; o.w = av.w > 0.f ? d.w : 0.f;
        reinterpret_cast<float4*>(dc)[i] = o;
    }
}

// One block per image: S[b][c] = sum_pix dh[b][pix][c] (gradient of the
// timestep bias, src/mnist.py:58-59), then dh <- dh * (a1 > 0) in place
// (gradient through the first ReLU, :57).
__global__ __launch_bounds__(EW_BLOCK) void relu_bwd_tb_kernel(float* __restrict__ dh, const float* __restrict__ a1,
                                                               float* __restrict__ S, int HWpix, int C) {
    __shared__ float4 sh[EW_BLOCK];
    const int C4 = C >> 2;
    const int c4 = threadIdx.x % C4, pg = threadIdx.x / C4, npg = EW_BLOCK / C4;
    const int64_t base = (int64_t)blockIdx.x * HWpix * C4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = pg; p < HWpix; p += npg) {
        const int64_t i = base + (int64_t)p * C4 + c4;
        const float4 d = reinterpret_cast<const float4*>(dh)[i];
        const float4 av = reinterpret_cast<const float4*>(a1)[i];
        acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
        float4 o;
        o.x = av.x > 0.f ? d.x : 0.f; o.y = av.y > 0.f ? d.y : 0.f;
        o.z = av.z > 0.f ? d.z : 0.f; o.w = av.w > 0.f ? d.w : 0.f;
        reinterpret_cast<float4*>(dh)[i] = o;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < C4) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int g = 0; g < npg; ++g) {
            const float4 v = sh[g * C4 + threadIdx.x];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4*>(S)[(int64_t)blockIdx.x * C4 + threadIdx.x] = s;
    }
}

// d_tw[c] = sum_b that[b]*S[b][c]; d_tb[c] = sum_b S[b][c].  One block per job (layer):
// 256 threads = (256/C) batch slices x C channels, 8 independent loads in flight per thread.
struct TimeGradJobs { const float* S[4]; const float* S2[4]; float* d_tw[4]; float* d_tb[4]; float* d_b[4]; int C[4]; int n; };
constexpr int TG_BLOCK = 1024;   // one workgroup per job: the loop over samples is latency-bound, use all 16 waves
__global__ __launch_bounds__(TG_BLOCK) void time_grad_kernel(TimeGradJobs jb, const float* __restrict__ that, int B) {
    __shared__ float shw[TG_BLOCK], shb[TG_BLOCK];
    const int job = blockIdx.x;
    const float* __restrict__ S = jb.S[job];
    const int C = jb.C[job];
    // (C need not divide the block: the threads past the last whole batch slice start beyond the batch and idle)
    const int c = threadIdx.x % C, ng = TG_BLOCK / C;
    const int g = (int)threadIdx.x / C < ng ? (int)threadIdx.x / C : B;
    float aw = 0.f, ab = 0.f;
    int b = g;
    for (; b + 7 * ng < B; b += 8 * ng) {
        float sv[8], tv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { sv[u] = S[(int64_t)(b + u * ng) * C + c]; tv[u] = that[b + u * ng]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { aw += tv[u] * sv[u]; ab += sv[u]; }
    }
    for (; b < B; b += ng) {
        const float sv = S[(int64_t)b * C + c];
        aw += that[b] * sv;
        ab += sv;
    }
    shw[threadIdx.x] = aw;
    shb[threadIdx.x] = ab;
    __syncthreads();
    if (threadIdx.x < C) {
        float sw = 0.f, sb = 0.f;
        for (int k = 0; k < ng; ++k) { sw += shw[k * C + threadIdx.x]; sb += shb[k * C + threadIdx.x]; }
        jb.d_tw[job][threadIdx.x] = sw;
        jb.d_tb[job][threadIdx.x] = sb;
    }
    // optional: conv1 bias gradient = sum over samples of the per-sample masked sums S2
    if (jb.S2[job] != nullptr) {
        const float* __restrict__ S2 = jb.S2[job];
        float a2 = 0.f;
        int bb = g;
        for (; bb + 7 * ng < B; bb += 8 * ng) {
            float sv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) sv[u] = S2[(int64_t)(bb + u * ng) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) a2 += sv[u];
        }
        for (; bb < B; bb += ng) a2 += S2[(int64_t)bb * C + c];
        __syncthreads();
        shw[threadIdx.x] = a2;
        __syncthreads();
        if (threadIdx.x < C) {
            float sacc = 0.f;
            for (int k = 0; k < ng; ++k) sacc += shw[k * C + threadIdx.x];
            jb.d_b[job][threadIdx.x] = sacc;
        }
    }
}

// dout3[b][y][x][c] = sum_{2x2} dcat[b][2y+dy][2x+dx][c], c < 64
// (backward of F.interpolate(scale 2, nearest) on the first 64 channels of the concat)
__global__ __launch_bounds__(EW_BLOCK) void split_dcat_kernel(const float* __restrict__ dcat, float* __restrict__ dout3,
                                                              int B) {
    const int64_t total = (int64_t)B * 196 * 16;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c4 = (int)(i & 15);
        const int64_t p = i >> 4;
        const int xo = (int)(p % 14);
        const int64_t q = p / 14;
        const int yo = (int)(q % 14);
        const int64_t b = q / 14;
        const float4* src = reinterpret_cast<const float4*>(dcat) + ((b * 28 + 2 * yo) * 28 + 2 * xo) * 24 + c4;
        const float4 v00 = src[0], v01 = src[24], v10 = src[28 * 24], v11 = src[28 * 24 + 24];
        float4 o;
        o.x = ((v00.x + v01.x) + v10.x) + v11.x; o.y = ((v00.y + v01.y) + v10.y) + v11.y;
        o.z = ((v00.z + v01.z) + v10.z) + v11.z; o.w = ((v00.w + v01.w) + v10.w) + v11.w;
        reinterpret_cast<float4*>(dout3)[i] = o;
    }
}

// dout1[p][c] = dcat[p][64+c] + 0.25*dp1[p/2][c]  (concat skip + avg-pool backward)
__global__ __launch_bounds__(EW_BLOCK) void combine_dh1_kernel(const float* __restrict__ dcat,
                                                               const float* __restrict__ dp1, float* __restrict__ dout1,
                                                               int B) {
    const int64_t total = (int64_t)B * 784 * 8;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c4 = (int)(i & 7);
        const int64_t p = i >> 3;
        const int x = (int)(p % 28);
        const int64_t q = p / 28;
        const int y = (int)(q % 28);
        const int64_t b = q / 28;
        const float4 dc = reinterpret_cast<const float4*>(dcat)[p * 24 + 16 + c4];
        const float4 dp = reinterpret_cast<const float4*>(dp1)[((b * 14 + (y >> 1)) * 14 + (x >> 1)) * 8 + c4];
        float4 o;
        o.x = dc.x + 0.25f * dp.x; o.y = dc.y + 0.25f * dp.y; o.z = dc.z + 0.25f * dp.z; o.w = dc.w + 0.25f * dp.w;
        reinterpret_cast<float4*>(dout1)[i] = o;
    }
}

// weight gradients of rb1.conv1 (Cin = 1, 3x3) and rb1.skip (1x1), per-block slabs.
// 1024 threads (16 waves per CU: one workgroup per slab); thread (c4 = tid & 7, g = tid >> 3) owns channels 4*c4 .. +3 of
// pixels g, g + 128, ... of the block's range.  The input pixels of the range (+ one image row and one pixel either side)
// are staged in LDS once per 4096-pixel piece: a tap is then an LDS broadcast read at a constant offset, masked by 3 row +
// 3 column validity bits, instead of a bounds-checked global load per tap and thread (the kernel issued 9 of them per 16-byte
// gradient load and ran at 2.3 TB/s); the pixel's (row, column) advances incrementally (no division in the loop), and
// four pixels' gradient loads are in flight per thread.  Partial sums: lanes of a wave by shuffles, the 16 waves through
// LDS, fixed order.
constexpr int FW_BLOCK = 1024;
constexpr int FW_PIECE = 4096;                 // pixels per staged piece
constexpr int FW_HALO = 32;                    // staged floats before the piece's first pixel (>= 29)
template <bool SKIP>   // SKIP: rb1.skip's weight / bias gradients are taken too (from dout1); the S16 pipeline takes them elsewhere
__global__ __launch_bounds__(FW_BLOCK) void first_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dc1,
                                                               const float* __restrict__ dout1, float* __restrict__ slab,
                                                               long slab_stride, int w1_off, int b1_off, int ws_off,
                                                               int bs_off, int B) {
    __shared__ float sh[16][12][32];
    __shared__ float xs[FW_PIECE + 2 * FW_HALO];
    const int c4 = threadIdx.x & 7, g = threadIdx.x >> 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t M = (int64_t)B * 784;
    const int64_t per = (M + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = (int64_t)blockIdx.x * per, p1 = min(p0 + per, M);
    float4 acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fma4 = [](float a, const float4& d, float4& r) {
        r.x = fmaf(a, d.x, r.x); r.y = fmaf(a, d.y, r.y); r.z = fmaf(a, d.z, r.z); r.w = fmaf(a, d.w, r.w);
    };
    constexpr bool skip = SKIP;
    constexpr int NF = SKIP ? 2 : 3;                             // pixels in flight per thread (register budget: 128 at this block size)
    for (int64_t q0 = p0; q0 < p1; q0 += FW_PIECE) {           // (one piece at B <= 1024 with 256 slabs)
        const int np = (int)min((int64_t)FW_PIECE, p1 - q0);
        __syncthreads();                                        // the previous piece's readers are done
        for (int e = threadIdx.x; e < np + 2 * FW_HALO; e += FW_BLOCK) {
            const int64_t m = q0 - FW_HALO + e;
            xs[e] = (m >= 0 && m < M) ? x[m] : 0.f;
        }
        __syncthreads();
        // (row, column) of this thread's first pixel of the piece; then + 128 pixels = + 4 rows + 16 columns per step
        int q = g;
        const int rem = (int)((q0 + g) % 784);
        int y = rem / 28, xx = rem - y * 28;
        auto one = [&](int ql, int yy, int xc_, const float4& d1, const float4& d2) {
            const float* c = xs + FW_HALO + ql;
            const bool r0 = yy >= 1, r2 = yy <= 26, c0 = xc_ >= 1, c2 = xc_ <= 26;
            const float v0 = (r0 && c0) ? c[-29] : 0.f, v1 = r0 ? c[-28] : 0.f, v2 = (r0 && c2) ? c[-27] : 0.f;
            const float v3 = c0 ? c[-1] : 0.f, v4 = c[0], v5 = c2 ? c[1] : 0.f;
            const float v6 = (r2 && c0) ? c[27] : 0.f, v7 = r2 ? c[28] : 0.f, v8 = (r2 && c2) ? c[29] : 0.f;
            fma4(v0, d1, acc[0]); fma4(v1, d1, acc[1]); fma4(v2, d1, acc[2]);
            fma4(v3, d1, acc[3]); fma4(v4, d1, acc[4]); fma4(v5, d1, acc[5]);
            fma4(v6, d1, acc[6]); fma4(v7, d1, acc[7]); fma4(v8, d1, acc[8]);
            acc[9].x += d1.x; acc[9].y += d1.y; acc[9].z += d1.z; acc[9].w += d1.w;
            if (skip) {
                fma4(v4, d2, acc[10]);
                acc[11].x += d2.x; acc[11].y += d2.y; acc[11].z += d2.z; acc[11].w += d2.w;
            }
        };
        auto advance = [&]() {   // the pixel 128 further on
            q += 128; xx += 16; y += 4;
            if (xx >= 28) { xx -= 28; ++y; }
            if (y >= 28) y -= 28;
        };
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* gd = dc1 + q0 * 32 + c4 * 4;
        const float* gs = skip ? dout1 + q0 * 32 + c4 * 4 : nullptr;
        for (; q + 128 * (NF - 1) < np; ) {   // NF pixels per trip: their gradient loads are independent
            float4 d[NF], e2[NF]; int qs[NF], ys[NF], xq[NF];
#pragma unroll
            for (int u = 0; u < NF; ++u) {
                qs[u] = q; ys[u] = y; xq[u] = xx;
                d[u] = *reinterpret_cast<const float4*>(gd + (long)q * 32);
                e2[u] = skip ? *reinterpret_cast<const float4*>(gs + (long)q * 32) : z4;
                advance();
            }
#pragma unroll
            for (int u = 0; u < NF; ++u) one(qs[u], ys[u], xq[u], d[u], e2[u]);
        }
        for (; q < np; advance())
            one(q, y, xx, *reinterpret_cast<const float4*>(gd + (long)q * 32), skip ? *reinterpret_cast<const float4*>(gs + (long)q * 32) : z4);
    }
    // lanes with the same c4 (lane & 7) hold different pixels: sum over lane bits 3..5
#pragma unroll
    for (int k = 0; k < (SKIP ? 12 : 10); ++k) {
#pragma unroll
        for (int o = 8; o <= 32; o <<= 1) {
            acc[k].x += __shfl_xor(acc[k].x, o); acc[k].y += __shfl_xor(acc[k].y, o);
            acc[k].z += __shfl_xor(acc[k].z, o); acc[k].w += __shfl_xor(acc[k].w, o);
        }
        if (lane < 8) *reinterpret_cast<float4*>(&sh[wave][k][lane * 4]) = acc[k];
    }
    __syncthreads();
    float* dst = slab + (long)blockIdx.x * slab_stride;
    for (int e = threadIdx.x; e < (SKIP ? 12 : 10) * 32; e += FW_BLOCK) {
        const int k = e >> 5, cc = e & 31;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += sh[w][k][cc];
        if (k < 9) dst[w1_off + k * 32 + cc] = s;
        else if (k == 9) dst[b1_off + cc] = s;
        else if (k == 10) dst[ws_off + cc] = s;
        else dst[bs_off + cc] = s;
    }
}

__global__ __launch_bounds__(EW_BLOCK) void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                int B, int HWpix, int C) {
    const int64_t total = (int64_t)B * HWpix * C;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int p = (int)(i % HWpix);
        const int64_t q = i / HWpix;
        const int c = (int)(q % C);
        const int64_t b = q / C;
        out[i] = in[(b * HWpix + p) * C + c];
    }
}

// ---- torch.optim.AdamW, single-tensor form  (src/mnist.py:148) ------------------
__global__ __launch_bounds__(EW_BLOCK) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                         float decay, float one_m_b1, float b2, float one_m_b2,
                                                         float step_size, float bc2_sqrt, float eps, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_BLOCK) {
        const float gi = g[i] * gscale;
        float pi = p[i] * decay;                        // param.mul_(1 - lr*wd)
        float mi = m[i];
        mi = mi + one_m_b1 * (gi - mi);                 // exp_avg.lerp_(grad, 1-beta1)
        const float vi = v[i] * b2 + one_m_b2 * gi * gi; // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - step_size * (mi / denom);             // param.addcdiv_(exp_avg, denom, -step_size)
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

}  // namespace

// ======================= S16-pipeline producers (tdm_s16.h) =======================
// rb1.conv1 + skip as conv_first_kernel, plus the pre-split copy split(a1 + tb) that rb1.conv2 reads.
// A workgroup owns a CONTIGUOUS pixel range; thread (c8 = tid & 3, g = tid >> 2) computes channels 8 c8 .. + 7 of pixels
// g, g + 64, ...: its 72 tap weights live in registers, the input pixels of the range (+ one image row and one pixel either
// side) are staged in LDS once, so a tap is an LDS broadcast read at a constant offset masked by 3 row + 3 column validity
// bits, and the pixel's (image, row, column) advances incrementally.  Stores are 16-byte pieces (hi / lo halves of a 16-channel
// S16 group), the ReLU mask 2 bytes.  (The previous form — 4 channels per thread, grid-stride — issued 9 tap loads and an
// index division chain per 16 bytes of output and ran at 2.8 TB/s.)
constexpr int CF_PIECE = 784, CF_HALO = 32;   // a piece spans at most two images: both time-bias rows are loaded before the loop
__global__ __launch_bounds__(EW_BLOCK) void conv_first_s16_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ w1,
                                                                  const float* __restrict__ b1,
                                                                  const float* __restrict__ ws,
                                                                  const float* __restrict__ bs,
                                                                  const float* __restrict__ tb, int tb_stride,
                                                                  float* __restrict__ a1, unsigned char* __restrict__ a1m,
                                                                  float* __restrict__ a1_s16, float* __restrict__ s, int B) {
    __shared__ float xs[CF_PIECE + 2 * CF_HALO];
    const int c8 = threadIdx.x & 3, g = threadIdx.x >> 2;
    float wv[9][8];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const float4 lo = *reinterpret_cast<const float4*>(w1 + tap * 32 + c8 * 8), hi = *reinterpret_cast<const float4*>(w1 + tap * 32 + c8 * 8 + 4);
        wv[tap][0] = lo.x; wv[tap][1] = lo.y; wv[tap][2] = lo.z; wv[tap][3] = lo.w;
        wv[tap][4] = hi.x; wv[tap][5] = hi.y; wv[tap][6] = hi.z; wv[tap][7] = hi.w;
    }
    float bv[8];
    {
        const float4 lo = *reinterpret_cast<const float4*>(b1 + c8 * 8), hi = *reinterpret_cast<const float4*>(b1 + c8 * 8 + 4);
        bv[0] = lo.x; bv[1] = lo.y; bv[2] = lo.z; bv[3] = lo.w; bv[4] = hi.x; bv[5] = hi.y; bv[6] = hi.z; bv[7] = hi.w;
    }
    const int M = B * 784;
    const int per = (M + gridDim.x - 1) / gridDim.x;
    const int p0 = min((int)blockIdx.x * per, M), p1 = min(p0 + per, M);
    for (int q0 = p0; q0 < p1; q0 += CF_PIECE) {
        const int np = min(CF_PIECE, p1 - q0);
        __syncthreads();
        for (int e = threadIdx.x; e < np + 2 * CF_HALO; e += EW_BLOCK) {
            const int m = q0 - CF_HALO + e;
            xs[e] = (m >= 0 && m < M) ? x[m] : 0.f;
        }
        __syncthreads();
        int b = (q0 + g) / 784;
        const int rem = (q0 + g) - b * 784;
        int y = rem / 28, xx = rem - y * 28;
        // time biases of the (at most two: a piece is no longer than an image) images this thread meets
        const int bA = b, bB = min(b + 1, B - 1);
        const float4 tA0 = *reinterpret_cast<const float4*>(tb + bA * tb_stride + c8 * 8), tA1 = *reinterpret_cast<const float4*>(tb + bA * tb_stride + c8 * 8 + 4);
        const float4 tB0 = *reinterpret_cast<const float4*>(tb + bB * tb_stride + c8 * 8), tB1 = *reinterpret_cast<const float4*>(tb + bB * tb_stride + c8 * 8 + 4);
        for (int q = g; q < np; q += 64) {
            const float* c = xs + CF_HALO + q;
            const bool r0 = y >= 1, r2 = y <= 26, c0 = xx >= 1, c2 = xx <= 26;
            float xv[9];
            xv[0] = (r0 && c0) ? c[-29] : 0.f; xv[1] = r0 ? c[-28] : 0.f; xv[2] = (r0 && c2) ? c[-27] : 0.f;
            xv[3] = c0 ? c[-1] : 0.f;          xv[4] = c[0];              xv[5] = c2 ? c[1] : 0.f;
            xv[6] = (r2 && c0) ? c[27] : 0.f;  xv[7] = r2 ? c[28] : 0.f;  xv[8] = (r2 && c2) ? c[29] : 0.f;
            // NO global load inside this loop: vmcnt counts loads and stores in order, so waiting for a load issued after the
            // previous pixel's stores means waiting for those stores to be acknowledged (that wait, not bytes or FMAs, was
            // the previous forms' 20 us)
            const float4 t0 = b == bA ? tA0 : tB0, t1 = b == bA ? tA1 : tB1;
            float acc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = bv[k];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = fmaf(xv[tap], wv[tap][k], acc[k]);
            unsigned mk = 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) { acc[k] = acc[k] < 0.f ? 0.f : acc[k]; mk |= (acc[k] > 0.f ? 1u : 0u) << ((k & 3) + 8 * (k >> 2)); }
            const int m = q0 + q;
            if (a1 != nullptr) {
                *reinterpret_cast<float4*>(a1 + (long)m * 32 + c8 * 8) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                *reinterpret_cast<float4*>(a1 + (long)m * 32 + c8 * 8 + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
            }
            if (a1m != nullptr)   // one byte per channel quad (bits 0..3), as the quad-per-thread consumers read it
                *reinterpret_cast<unsigned short*>(a1m + (long)m * 8 + c8 * 2) = (unsigned short)mk;
            {
                const float tv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                tdm_bf16x4 h0, l0, h1, l1;
                tdm_split4(make_float4(acc[0] + tv[0], acc[1] + tv[1], acc[2] + tv[2], acc[3] + tv[3]), h0, l0);
                tdm_split4(make_float4(acc[4] + tv[4], acc[5] + tv[5], acc[6] + tv[6], acc[7] + tv[7]), h1, l1);
                typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
                bf16x8_ vh, vl;
                vh[0] = h0[0]; vh[1] = h0[1]; vh[2] = h0[2]; vh[3] = h0[3]; vh[4] = h1[0]; vh[5] = h1[1]; vh[6] = h1[2]; vh[7] = h1[3];
                vl[0] = l0[0]; vl[1] = l0[1]; vl[2] = l0[2]; vl[3] = l0[3]; vl[4] = l1[0]; vl[5] = l1[1]; vl[6] = l1[2]; vl[7] = l1[3];
                // The four threads of a pixel exchange pieces inside their lane quad so that ONE store instruction writes a
                // whole 64-byte S16 group per pixel ([hi0 hi1 lo0 lo1] = lanes r = 0..3) instead of two instructions writing
                // its hi and its lo half: partial 64-byte blocks cost the L2 write path a request each (tdm_s16.h).
                const uint4 uh = __builtin_bit_cast(uint4, vh), ul = __builtin_bit_cast(uint4, vl);
                const bool hi_lane = c8 < 2;
                uint4 o0, o1;
#define TDM_QP(v, ctrl) (unsigned)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xf, 0xf, true)
                // (every lane executes BOTH exchanges, the select comes afterwards: a lane that skipped one would be read as 0)
                const uint4 hA = make_uint4(TDM_QP(uh.x, 0x44), TDM_QP(uh.y, 0x44), TDM_QP(uh.z, 0x44), TDM_QP(uh.w, 0x44));
                const uint4 lA = make_uint4(TDM_QP(ul.x, 0x44), TDM_QP(ul.y, 0x44), TDM_QP(ul.z, 0x44), TDM_QP(ul.w, 0x44));
                const uint4 hB = make_uint4(TDM_QP(uh.x, 0xEE), TDM_QP(uh.y, 0xEE), TDM_QP(uh.z, 0xEE), TDM_QP(uh.w, 0xEE));
                const uint4 lB = make_uint4(TDM_QP(ul.x, 0xEE), TDM_QP(ul.y, 0xEE), TDM_QP(ul.z, 0xEE), TDM_QP(ul.w, 0xEE));
                o0 = hi_lane ? hA : lA;
                o1 = hi_lane ? hB : lB;
#undef TDM_QP
                char* gp = reinterpret_cast<char*>(a1_s16 + (long)m * 32) + c8 * 16;   // group 0: bytes 0..63 of the pixel, group 1: 64..127
                *reinterpret_cast<uint4*>(gp) = o0;
                *reinterpret_cast<uint4*>(gp + 64) = o1;
            }
            if (s != nullptr) {
                const float4 w0 = *reinterpret_cast<const float4*>(ws + c8 * 8), w1_ = *reinterpret_cast<const float4*>(ws + c8 * 8 + 4);
                const float4 s0 = *reinterpret_cast<const float4*>(bs + c8 * 8), s1 = *reinterpret_cast<const float4*>(bs + c8 * 8 + 4);
                const float xc = xv[4];
                *reinterpret_cast<float4*>(s + (long)m * 32 + c8 * 8) =
                    make_float4(fmaf(xc, w0.x, s0.x), fmaf(xc, w0.y, s0.y), fmaf(xc, w0.z, s0.z), fmaf(xc, w0.w, s0.w));
                *reinterpret_cast<float4*>(s + (long)m * 32 + c8 * 8 + 4) =
                    make_float4(fmaf(xc, w1_.x, s1.x), fmaf(xc, w1_.y, s1.y), fmaf(xc, w1_.z, s1.z), fmaf(xc, w1_.w, s1.w));
            }
            // the pixel 64 further on: + 2 rows + 8 columns
            xx += 8; y += 2;
            if (xx >= 28) { xx -= 28; ++y; }
            if (y >= 28) { y -= 28; ++b; }
        }
    }
}

__global__ __launch_bounds__(EW_BLOCK) void avgpool_s16_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                               float* __restrict__ out_s16, int B, int Ho, int C) {
    const int C4 = C >> 2, Wo = Ho, Wi = 2 * Ho;
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c4 = (int)(i % C4);
        const int64_t p = i / C4;
        const int xo = (int)(p % Wo);
        const int64_t q = p / Wo;
        const int yo = (int)(q % Ho);
        const int64_t b = q / Ho;
        const float4* src = reinterpret_cast<const float4*>(in) + ((b * Wi + 2 * yo) * Wi + 2 * xo) * C4 + c4;
        const float4 v00 = src[0], v01 = src[C4], v10 = src[(int64_t)Wi * C4], v11 = src[(int64_t)Wi * C4 + C4];
        float4 o;
        o.x = (((v00.x + v01.x) + v10.x) + v11.x) * 0.25f;
        o.y = (((v00.y + v01.y) + v10.y) + v11.y) * 0.25f;
        o.z = (((v00.z + v01.z) + v10.z) + v11.z) * 0.25f;
        o.w = (((v00.w + v01.w) + v10.w) + v11.w) * 0.25f;
        if (out != nullptr) reinterpret_cast<float4*>(out)[i] = o;
        tdm_store_s16_4(out_s16, p, C, c4 * 4, o);
    }
}

// rb2's entry in one pass (src/mnist.py:80 + the block's 1x1 skip, :52,61): p1 = avg_pool2d(h1, 2) -> S16 twin (the
// only form rb2.conv1 reads) and s2 = skip(p1) + bias in exact fp32 on the vector units (32 x 64 MACs per pixel: 0.4 GFLOP
// at B = 512 — a matrix-core launch for it cost 15 us of a train step and 4 % of a B = 4096 reverse step).
// 256 threads = 16 output pixels per trip: threads 0..127 pool one channel quad each into LDS, then thread
// (pixel, co quad) runs the 32-term dot products against the LDS-resident weights.
__global__ __launch_bounds__(EW_BLOCK) void pool_skip_s16_kernel(const float* __restrict__ h1, const float* __restrict__ wsk,
                                                                 const float* __restrict__ bsk, float* __restrict__ p1_s16,
                                                                 float* __restrict__ s2, int B) {
    __shared__ float4 Wl[32 * 16];     // [ci][co quad]
    __shared__ float pooled[16][32];
    for (int i = threadIdx.x; i < 32 * 16; i += EW_BLOCK) Wl[i] = reinterpret_cast<const float4*>(wsk)[i];
    const int npix = B * 196;
    const int px = threadIdx.x >> 4, cq = threadIdx.x & 15;
    const float4 bias = reinterpret_cast<const float4*>(bsk)[cq];
    // software pipeline: the four S16 pixels of the NEXT trip are requested before the 32 x 64 products of the current one
    const int lp = threadIdx.x >> 3, c4 = threadIdx.x & 7;   // pooling role (threads 0..127): pixel lp of the trip, channel quad c4
    float4 v00, v01, v10, v11;
    auto fetch = [&](int p0) {
        const int p = p0 + lp;
        v00 = v01 = v10 = v11 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (threadIdx.x < 128 && p < npix) {
            const int q = p / 14, xo = p - q * 14;
            const int b = q / 14, yo = q - b * 14;
            const long m00 = (long)(b * 28 + 2 * yo) * 28 + 2 * xo;   // h1 is read from its S16 twin (hi + lo): no fp32 copy exists
            v00 = tdm_load_s16_4(h1, m00, 32, c4 * 4); v01 = tdm_load_s16_4(h1, m00 + 1, 32, c4 * 4);
            v10 = tdm_load_s16_4(h1, m00 + 28, 32, c4 * 4); v11 = tdm_load_s16_4(h1, m00 + 29, 32, c4 * 4);
        }
    };
    const int step = gridDim.x * 16;
    int p0 = blockIdx.x * 16;
    if (p0 < npix) fetch(p0);
    for (; p0 < npix; p0 += step) {
        __syncthreads();   // previous trip's pooled values consumed (and, first trip, Wl complete)
        if (threadIdx.x < 128) {
            float4 o;
            o.x = (((v00.x + v01.x) + v10.x) + v11.x) * 0.25f;
            o.y = (((v00.y + v01.y) + v10.y) + v11.y) * 0.25f;
            o.z = (((v00.z + v01.z) + v10.z) + v11.z) * 0.25f;
            o.w = (((v00.w + v01.w) + v10.w) + v11.w) * 0.25f;
            if (p0 + lp < npix) tdm_store_s16_4(p1_s16, p0 + lp, 32, c4 * 4, o);
            *reinterpret_cast<float4*>(&pooled[lp][c4 * 4]) = o;
        }
        __syncthreads();
        if (p0 + step < npix) fetch(p0 + step);
        const int p = p0 + px;
        if (p < npix) {
            float4 acc = bias;
#pragma unroll
            for (int ci = 0; ci < 32; ++ci) {
                const float a = pooled[px][ci];
                const float4 w4 = Wl[ci * 16 + cq];
                acc.x = fmaf(a, w4.x, acc.x); acc.y = fmaf(a, w4.y, acc.y);
                acc.z = fmaf(a, w4.z, acc.z); acc.w = fmaf(a, w4.w, acc.w);
            }
            reinterpret_cast<float4*>(s2)[(long)p * 16 + cq] = acc;
        }
    }
}

// block-level per-channel-quad reduction: threads with equal (tid % C4) hold partial sums of the same 4 channels
__device__ __forceinline__ void quad_reduce_store(float4 v, float4* sh, int C4, float* dst /* slab + off, or nullptr */) {
    __syncthreads();
    sh[threadIdx.x] = v;
    __syncthreads();
    if (dst != nullptr && threadIdx.x < C4) {
        float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = threadIdx.x; k < EW_BLOCK; k += C4) {
            const float4 t = sh[k];
            sacc.x += t.x; sacc.y += t.y; sacc.z += t.z; sacc.w += t.w;
        }
        *reinterpret_cast<float4*>(dst + threadIdx.x * 4) = sacc;
    }
}

// ReLU backward with a byte mask (ConvArgs::mask_out): bit e of mk = (post-ReLU channel e of the quad > 0)
__device__ __forceinline__ float4 mask4(const float4 d, unsigned mk) {
    return make_float4((mk & 1u) ? d.x : 0.f, (mk & 2u) ? d.y : 0.f, (mk & 4u) ? d.z : 0.f, (mk & 8u) ? d.w : 0.f);
}

// Backward of the model's 1x1 output conv (src/mnist.py:87) and everything that follows from dout4 = d(loss)/d(h4) being
// RANK ONE: dout4[m][c] = d[m] * w_out[c] with d = d(loss)/d(eps).  rb4's skip conv (a 1x1 conv over the 96-channel concat,
// src/mnist.py:52,61) receives exactly dout4 as its output gradient, so
//   dW_skip[ci][co] = (sum_m cat[m][ci] d[m]) * w_out[co]     -> this kernel emits the 96-vector v = sum_m cat[m] d[m]
//   db_skip[co]     = (sum_m d[m]) * w_out[co]
//   d cat[m][ci]   += d[m] * u[ci],  u = W_skip w_out          -> added in the epilogue of rb4.conv1's data gradient
// and dout4 itself is never written (it was a 51 MB S16 tensor read by three launches, and a second K source with its own
// staged images in the data gradient).  What stays a tensor: dc2 = dout4 * (a2 > 0) (S16), the input of rb4.conv2's
// gradients.  Also: dW_out = sum d h4, db_out = sum d, db(rb4.conv2) = sum dc2, and — fused — F.mse_loss forward/backward.
// cat = [up2(h3) | h1] exists only as the S16 tensors h3s (B,14,14,64) and h1s (B,28,28,32).
__global__ __launch_bounds__(EW_BLOCK) void out_bwd_s16_kernel(const float* __restrict__ deps,
                                                               const float* __restrict__ h4, const float* __restrict__ w,
                                                               const unsigned char* __restrict__ a2m,
                                                               const float* __restrict__ h1s, const float* __restrict__ h3s,
                                                               float* __restrict__ dc2_s16,
                                                               float* __restrict__ slab, long slab_stride, int w_off,
                                                               int b_off, int c2b_off, int skb_off, int vsk_off, int64_t M,
                                                               const float* __restrict__ eps, const float* __restrict__ noise,
                                                               float* __restrict__ deps_out, float dscale, int loss_off,
                                                               const float* __restrict__ o1_sums, int o1_rows) {
    __shared__ float4 shw[EW_BLOCK];
    __shared__ float shb[4];
    const int c4 = threadIdx.x & 7;
    const float4 wv = *reinterpret_cast<const float4*>(w + c4 * 4);
    float4 gw = make_float4(0.f, 0.f, 0.f, 0.f), g_c2 = gw, g_v1 = gw, g_v3 = gw;
    float gb = 0.f, gl = 0.f;
    const int64_t total = M * 8;
    const int64_t S = (int64_t)gridDim.x * EW_BLOCK;
    const bool fused = deps == nullptr;      // MSE backward fused: d = (eps - noise) * 2/M  (mse_kernel's arithmetic)
    auto dval = [&](int64_t m) {
        if (!fused) return deps[m];
        const float df = eps[m] - noise[m];
        if (c4 == 0) {
            gl += df * df;
            if (deps_out != nullptr) deps_out[m] = df * dscale;
        }
        return df * dscale;
    };
    auto dpure = [&](int64_t m) { return fused ? (eps[m] - noise[m]) * dscale : deps[m]; };   // (no side effects: second pass)
    // o1_sums != nullptr: rb4.conv2's forward epilogue has already taken the MSE backward (deps holds d) and the output
    // conv's gradients as per-32-pixel-group rows (ConvArgs::o1_sums) — h4 was never written; this kernel only condenses
    // the rows of its slice into its partial row.
    const bool has_h4 = h4 != nullptr;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    auto finish = [&](int64_t i, float d, const float4& hv, unsigned am, const float4& h1v) {
        const int64_t m = i >> 3;
        float4 o;
        o.x = d * wv.x; o.y = d * wv.y; o.z = d * wv.z; o.w = d * wv.w;
        const float4 mk = mask4(o, am);
        tdm_store_s16_4(dc2_s16, m, 32, c4 * 4, mk);
        gw.x += d * hv.x; gw.y += d * hv.y; gw.z += d * hv.z; gw.w += d * hv.w;
        g_c2.x += mk.x; g_c2.y += mk.y; g_c2.z += mk.z; g_c2.w += mk.w;
        g_v1.x += d * h1v.x; g_v1.y += d * h1v.y; g_v1.z += d * h1v.z; g_v1.w += d * h1v.w;
        if (c4 == 0) gb += d;
    };
    int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x;
    // four independent items in flight per thread (one workgroup per slab = one wave per SIMD: latency-bound otherwise)
    for (; i + 3 * S < total; i += 4 * S) {
        float d[4]; float4 hv[4], h1v[4]; unsigned am[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t k = i + u * S;
            d[u] = dval(k >> 3); hv[u] = has_h4 ? reinterpret_cast<const float4*>(h4)[k] : zero; am[u] = a2m[k];
            h1v[u] = tdm_load_s16_4(h1s, k >> 3, 32, c4 * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) finish(i + u * S, d[u], hv[u], am[u], h1v[u]);
    }
    for (; i < total; i += S)
        finish(i, dval(i >> 3), has_h4 ? reinterpret_cast<const float4*>(h4)[i] : zero, a2m[i], tdm_load_s16_4(h1s, i >> 3, 32, c4 * 4));
    // v over the up-sampled channels: sum over the half-resolution pixels of h3 * (sum of d over its 2x2 fine pixels)
    {
        const int q4 = threadIdx.x & 15;                 // channel quad of the 64 (the grid stride is a multiple of 16)
        const int64_t total3 = (M >> 2) * 16;
        for (int64_t k = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; k < total3; k += S) {
            const int p = (int)(k >> 4);
            const int q = p / 14, xo = p - q * 14;
            const int b = q / 14, yo = q - b * 14;
            const int64_t m00 = (int64_t)(b * 28 + 2 * yo) * 28 + 2 * xo;
            const float ds = (dpure(m00) + dpure(m00 + 1)) + (dpure(m00 + 28) + dpure(m00 + 29));
            const float4 hv = tdm_load_s16_4(h3s, p, 64, q4 * 4);
            g_v3.x += ds * hv.x; g_v3.y += ds * hv.y; g_v3.z += ds * hv.z; g_v3.w += ds * hv.w;
        }
    }
    float* dst = slab + (long)blockIdx.x * slab_stride;
    if (fused && loss_off >= 0) {
        const float sl = block_sum(gl, shb);
        if (threadIdx.x == 0) dst[loss_off] = sl;
    }
    const float sb = block_sum(gb, shb);
    if (o1_sums == nullptr) quad_reduce_store(gw, shw, 8, dst + w_off);
    quad_reduce_store(g_c2, shw, 8, dst + c2b_off);
    quad_reduce_store(g_v3, shw, 16, dst + vsk_off);
    quad_reduce_store(g_v1, shw, 8, dst + vsk_off + 64);
    if (threadIdx.x == 0 && o1_sums == nullptr) dst[b_off] = sb;
    if (o1_sums != nullptr && threadIdx.x < 34) {   // [0..31] d W_out, [32] d b_out, [33] sum of squared errors: fixed row order
        float acc = 0.f;
        for (int r = blockIdx.x; r < o1_rows; r += gridDim.x) acc += o1_sums[(long)r * 40 + threadIdx.x];
        if (threadIdx.x < 32) dst[w_off + threadIdx.x] = acc;
        else if (threadIdx.x == 32) dst[b_off] = acc;
        else if (loss_off >= 0) dst[loss_off] = acc;
    }
    __syncthreads();
    if (threadIdx.x < 32) dst[skb_off + threadIdx.x] = w[threadIdx.x] * ((shb[0] + shb[1]) + (shb[2] + shb[3]));   // db_skip partial = w_out * sum d
}

// dc_s16 = split(dout * (a > 0)); per-channel slab partials of the masked (and unmasked) gradient
__global__ __launch_bounds__(EW_BLOCK) void relu_mask_s16_kernel(const float* __restrict__ dout,
                                                                 const unsigned char* __restrict__ am, float* __restrict__ dc_s16,
                                                                 float* __restrict__ slab, long slab_stride,
                                                                 int b_masked_off, int b_unmasked_off, int64_t M, int C) {
    __shared__ float4 sh[EW_BLOCK];
    const int C4 = C >> 2;
    const int c4 = threadIdx.x % C4;             // fixed per thread: the grid stride is a multiple of C4
    float4 gm = make_float4(0.f, 0.f, 0.f, 0.f), gu = gm;
    const int64_t total = M * C4;
    const int64_t S = (int64_t)gridDim.x * EW_BLOCK;
    int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x;
    // four independent items in flight per thread (one workgroup per slab = one wave per SIMD: the loop is latency-bound)
    for (; i + 3 * S < total; i += 4 * S) {
        float4 d[4]; unsigned mk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { d[u] = reinterpret_cast<const float4*>(dout)[i + u * S]; mk[u] = am[i + u * S]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 o = mask4(d[u], mk[u]);
            tdm_store_s16_4(dc_s16, (i + u * S) / C4, C, c4 * 4, o);
            gm.x += o.x; gm.y += o.y; gm.z += o.z; gm.w += o.w;
            gu.x += d[u].x; gu.y += d[u].y; gu.z += d[u].z; gu.w += d[u].w;
        }
    }
    for (; i < total; i += S) {
        const float4 d = reinterpret_cast<const float4*>(dout)[i];
        const float4 o = mask4(d, am[i]);
        tdm_store_s16_4(dc_s16, i / C4, C, c4 * 4, o);
        gm.x += o.x; gm.y += o.y; gm.z += o.z; gm.w += o.w;
        gu.x += d.x; gu.y += d.y; gu.z += d.z; gu.w += d.w;
    }
    float* dst = slab + (long)blockIdx.x * slab_stride;
    quad_reduce_store(gm, sh, C4, dst + b_masked_off);
    quad_reduce_store(gu, sh, C4, b_unmasked_off >= 0 ? dst + b_unmasked_off : nullptr);
}

// Upsample backward + ReLU mask of rb3.conv2's output in one pass (rb3): dout3 = 2x2 sum of d cat[.., 0:64].  The data
// gradient of rb4.conv1 has already added horizontally adjacent pixels in its epilogue (the pair sits in one wave's transpose
// block): `dch` is (B, 28, 14, 64), and this kernel adds the two rows — 2 loads per output instead of 4, half the bytes.
// dc2_s16 = split(dout3 * (a2 > 0)), slab partials of the masked gradient (rb3.conv2 bias gradient).
// One workgroup per slab; c4 = tid & 15 is fixed per thread (the grid stride is a multiple of 16).
__global__ __launch_bounds__(EW_BLOCK) void split_dcat_mask_s16_kernel(const float* __restrict__ dch,
                                                                       const unsigned char* __restrict__ a2m,
                                                                       float* __restrict__ dout3, float* __restrict__ dc_s16,
                                                                       float* __restrict__ slab, long slab_stride,
                                                                       int b_masked_off, int B) {
    __shared__ float4 sh[EW_BLOCK];
    float4 gm = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t total = (int64_t)B * 196 * 16;
    const int64_t S = (int64_t)gridDim.x * EW_BLOCK;
    const int c4 = threadIdx.x & 15;   // (the grid stride is a multiple of 16)
    struct Item { float4 v0, v1; unsigned am; };
    auto fetch = [&](int64_t i, Item& it) {   // (32-bit index math)
        const int p = (int)(i >> 4);
        const int q = p / 14, xo = p - q * 14;      // q = b * 14 + yo: rows 2q and 2q + 1 of the (B * 28)-row half-width image
        const float4* src = reinterpret_cast<const float4*>(dch) + (unsigned)((2 * q * 14 + xo) * 16 + c4);
        it.v0 = src[0]; it.v1 = src[14 * 16];
        it.am = a2m[i];
    };
    auto finish = [&](int64_t i, const Item& it) {
        float4 d;
        d.x = it.v0.x + it.v1.x; d.y = it.v0.y + it.v1.y; d.z = it.v0.z + it.v1.z; d.w = it.v0.w + it.v1.w;
        reinterpret_cast<float4*>(dout3)[i] = d;
        const float4 o = mask4(d, it.am);
        tdm_store_s16_4(dc_s16, i >> 4, 64, c4 * 4, o);
        gm.x += o.x; gm.y += o.y; gm.z += o.z; gm.w += o.w;
    };
    int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x;
    for (; i + 3 * S < total; i += 4 * S) {   // four items (eight 16-byte loads) in flight per thread
        Item a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) fetch(i + u * S, a[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) finish(i + u * S, a[u]);
    }
    for (; i < total; i += S) { Item a; fetch(i, a); finish(i, a); }
    quad_reduce_store(gm, sh, 16, slab + (long)blockIdx.x * slab_stride + b_masked_off);
}

// combine_dh1_kernel + relu_mask_s16_kernel in one pass (rb1): dout1 = dcat_h1 (= d cat[.., 64:96], (M, 32)) + 0.25 * dp1 (concat skip +
// avg-pool backward), dc2_s16 = split(dout1 * (a2 > 0)), slab partials of the masked gradient (rb1.conv2 bias).
__global__ __launch_bounds__(EW_BLOCK) void combine_dh1_mask_s16_kernel(const float* __restrict__ dcat,
                                                                        const float* __restrict__ dp1,
                                                                        const unsigned char* __restrict__ a2m,
                                                                        float* __restrict__ dout1, float* __restrict__ dc_s16,
                                                                        float* __restrict__ slab, long slab_stride,
                                                                        int b_masked_off, int B, const float* __restrict__ xin,
                                                                        int ws_off, int bs_off) {
    __shared__ float4 sh[EW_BLOCK];
    float4 gm = make_float4(0.f, 0.f, 0.f, 0.f), gsw = gm, gsb = gm;
    const int64_t total = (int64_t)B * 784 * 8;
    const int64_t S = (int64_t)gridDim.x * EW_BLOCK;
    const int c4 = threadIdx.x & 7;   // (the grid stride is a multiple of 8)
    auto fetch = [&](int64_t i, float4& dc, float4& dp, unsigned& am, float& xv) {   // (32-bit index math: B * 784 * 24 < 2^31 / 4)
        const int p = (int)(i >> 3);
        const int q = p / 28, x = p - q * 28;
        const int b = q / 28, y = q - b * 28;
        dc = reinterpret_cast<const float4*>(dcat)[(unsigned)(p * 8 + c4)];           // d cat[.., 64:96] as its own (M, 32) tensor
        dp = reinterpret_cast<const float4*>(dp1)[(unsigned)(((b * 14 + (y >> 1)) * 14 + (x >> 1)) * 8 + c4)];
        am = a2m[(unsigned)i];
        xv = xin != nullptr ? xin[p] : 0.f;
    };
    auto finish = [&](int64_t i, const float4& dc, const float4& dp, unsigned am, float xv) {
        float4 d;
        d.x = dc.x + 0.25f * dp.x; d.y = dc.y + 0.25f * dp.y; d.z = dc.z + 0.25f * dp.z; d.w = dc.w + 0.25f * dp.w;
        if (dout1 != nullptr) reinterpret_cast<float4*>(dout1)[i] = d;
        const float4 o = mask4(d, am);
        tdm_store_s16_4(dc_s16, i >> 3, 32, c4 * 4, o);
        gm.x += o.x; gm.y += o.y; gm.z += o.z; gm.w += o.w;
        // rb1.skip (one input channel): dW[c] += x[m] * dout1[m][c], db[c] += dout1[m][c]  (first_wgrad_kernel's fma)
        gsw.x = fmaf(xv, d.x, gsw.x); gsw.y = fmaf(xv, d.y, gsw.y); gsw.z = fmaf(xv, d.z, gsw.z); gsw.w = fmaf(xv, d.w, gsw.w);
        gsb.x += d.x; gsb.y += d.y; gsb.z += d.z; gsb.w += d.w;
    };
    int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x;
    // four independent items in flight per thread (one workgroup per slab = one wave per SIMD: latency-bound otherwise)
    for (; i + 3 * S < total; i += 4 * S) {
        float4 dc[4], dp[4]; unsigned am[4]; float xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) fetch(i + u * S, dc[u], dp[u], am[u], xv[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) finish(i + u * S, dc[u], dp[u], am[u], xv[u]);
    }
    for (; i < total; i += S) {
        float4 dc, dp; unsigned am; float xv;
        fetch(i, dc, dp, am, xv);
        finish(i, dc, dp, am, xv);
    }
    float* dst = slab + (long)blockIdx.x * slab_stride;
    quad_reduce_store(gm, sh, 8, dst + b_masked_off);
    quad_reduce_store(gsw, sh, 8, (xin != nullptr && ws_off >= 0) ? dst + ws_off : nullptr);
    quad_reduce_store(gsb, sh, 8, (xin != nullptr && bs_off >= 0) ? dst + bs_off : nullptr);
}

// time_emb / conv1-bias gradient partials of all four blocks from the per-group sums (GroupSumJobs, tdm_common.h):
// d_tw[c] = sum_b that[b] * S[b][c], d_tb[c] = sum_b S[b][c] (S = unmasked sums), d_b(conv1)[c] = sum of the masked sums
// (src/mnist.py:58-59 backward).  One workgroup per slab; a group's slot 0 / 1 belong to image img0 / img0 + 1.
__global__ __launch_bounds__(EW_BLOCK) void group_sums_kernel(GroupSumJobs jb, const float* __restrict__ that, int B,
                                                              float* __restrict__ slab, long slab_stride) {
    __shared__ float4 sh[EW_BLOCK];
    float* dst = slab + (long)blockIdx.x * slab_stride;
    {
        const int job = jb.job0 + blockIdx.y;   // the blocks' sums side by side: grid = (partial rows, jobs)
        const int C = jb.C[job], C4 = C >> 2, HW = jb.HWpix[job];
        const long M = (long)B * HW;
        const int G = (int)((M + 31) >> 5);
        const int per = (G + gridDim.x - 1) / gridDim.x;
        const int g0 = blockIdx.x * per, g1 = min(g0 + per, G);
        const int c4 = threadIdx.x % C4, gi = threadIdx.x / C4, ng = EW_BLOCK / C4;
        float4 aw = make_float4(0.f, 0.f, 0.f, 0.f), ab = aw, a2 = aw;
        for (int g = g0 + gi; g < g1; g += ng) {
            const int img0 = (int)(((long)g << 5) / HW);
            const float t0 = that[img0], t1 = that[min(img0 + 1, B - 1)];
            const float4* p = reinterpret_cast<const float4*>(jb.gs[job] + (long)g * 4 * C) + c4;   // [slot][kind][C]
            const float4 s00 = p[0], s01 = p[C4], s10 = p[2 * C4], s11 = p[3 * C4];
            aw.x += t0 * s00.x + t1 * s10.x; aw.y += t0 * s00.y + t1 * s10.y;
            aw.z += t0 * s00.z + t1 * s10.z; aw.w += t0 * s00.w + t1 * s10.w;
            ab.x += s00.x + s10.x; ab.y += s00.y + s10.y; ab.z += s00.z + s10.z; ab.w += s00.w + s10.w;
            a2.x += s01.x + s11.x; a2.y += s01.y + s11.y; a2.z += s01.z + s11.z; a2.w += s01.w + s11.w;
        }
        quad_reduce_store(aw, sh, C4, jb.tew[job] >= 0 ? dst + jb.tew[job] : nullptr);
        quad_reduce_store(ab, sh, C4, jb.tew[job] >= 0 ? dst + jb.tew[job] + C : (jb.ub[job] > 0 ? dst + jb.ub[job] : nullptr));
        quad_reduce_store(a2, sh, C4, jb.c1b[job] >= 0 ? dst + jb.c1b[job] : nullptr);
    }
}

// ------------------------------- launchers -----------------------------------
int tdm_launch_timebias(const int64_t* t, const float* params, const int* te_w_off, const int* te_b_off, float* that,
                        float* tb, int B, hipStream_t st, int64_t* bump, float* u96, int skw4_off, int outw_off) {
    TeOffs o;
    for (int i = 0; i < 4; ++i) { o.w[i] = te_w_off[i]; o.b[i] = te_b_off[i]; }
    o.skw4 = skw4_off; o.outw = outw_off;
    hipLaunchKernelGGL(timebias_kernel, dim3(ew_grid((int64_t)B * 192)), dim3(EW_BLOCK), 0, st, TimebiasArgs{t, params, o, that, tb, B, bump, u96});
    TDM_CHECK_LAUNCH("timebias");
    return 0;
}
int tdm_launch_timebias_float(const float* that, const float* w, const float* bias, float* tb, int B, int C, hipStream_t st) {
    hipLaunchKernelGGL(timebias_float_kernel, dim3(ew_grid((int64_t)B * C)), dim3(EW_BLOCK), 0, st, that, w, bias, tb, B, C);
    TDM_CHECK_LAUNCH("timebias_float");
    return 0;
}

int tdm_launch_conv_first(const float* x, const float* w1, const float* b1, const float* ws, const float* bs, float* a1,
                          float* s, int B, hipStream_t st) {
    hipLaunchKernelGGL(conv_first_kernel, dim3(ew_grid((int64_t)B * 784 * 8)), dim3(EW_BLOCK), 0, st, x, w1, b1, ws, bs,
                       a1, s, B);
    TDM_CHECK_LAUNCH("conv_first");
    return 0;
}
int tdm_launch_avgpool(const float* in, float* out, int B, int Hout, int C, hipStream_t st) {
    hipLaunchKernelGGL(avgpool_kernel, dim3(ew_grid((int64_t)B * Hout * Hout * (C / 4))), dim3(EW_BLOCK), 0, st, in, out,
                       B, Hout, C);
    TDM_CHECK_LAUNCH("avgpool");
    return 0;
}
int tdm_launch_conv_out(const float* h, const float* w, const float* b, float* eps, int64_t M, hipStream_t st) {
    hipLaunchKernelGGL(conv_out_kernel, dim3(ew_grid(M * 8)), dim3(EW_BLOCK), 0, st, h, w, b, eps, M);
    TDM_CHECK_LAUNCH("conv_out");
    return 0;
}
int tdm_launch_out_bwd(const float* deps, const float* h4, const float* w, const float* a2, float* dout, float* dc2,
                       float* slab, long slab_stride, int w_off, int b_off, int64_t M, int nslab, hipStream_t st) {
    hipLaunchKernelGGL(out_bwd_kernel, dim3(nslab), dim3(EW_BLOCK), 0, st, deps, h4, w, a2, dout, dc2, slab, slab_stride,
                       w_off, b_off, M);
    TDM_CHECK_LAUNCH("out_bwd");
    return 0;
}
int tdm_launch_skip4_factored(const float* deps, const float* h1, const float* h3, const float* w_out, float* slab, long slab_stride,
                              int skw_off, int skb_off, int B, int nslab, hipStream_t st) {
    hipLaunchKernelGGL(skip4_factored_kernel, dim3(nslab), dim3(EW_BLOCK), 0, st, deps, h1, h3, w_out, slab, slab_stride, skw_off,
                       skb_off, B);
    TDM_CHECK_LAUNCH("skip4_factored");
    return 0;
}
int tdm_launch_relu_mask(const float* dout, const float* a, float* dc, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(relu_mask_kernel, dim3(ew_grid(n / 4)), dim3(EW_BLOCK), 0, st, dout, a, dc, n / 4);
    TDM_CHECK_LAUNCH("relu_mask");
    return 0;
}
// S[b][c] (kind 0) and S2[b][c] (kind 1) from the per-32-pixel-group partials a data-gradient conv launch wrote
// (ConvArgs::sums): one workgroup per image, thread (kind, c); fixed summation order.
__global__ __launch_bounds__(128) void image_sums_kernel(const float* __restrict__ sums, float* __restrict__ S,
                                                         float* __restrict__ S2, int HWpix, int C) {
    const int b = blockIdx.x;
    const int kd = threadIdx.x / C, c = threadIdx.x - kd * C;
    if (kd >= 2) return;
    const long p0 = (long)b * HWpix, p1 = p0 + HWpix;      // pixel range of the image
    const long g0 = p0 >> 5, g1 = (p1 - 1) >> 5;
    // only the first group can start in the previous image (it then reaches this image through slot 1)
    const int sl0 = ((g0 << 5) < p0) ? 1 : 0;
    const long gstride = 4L * C;                            // floats per group: [slot][kind][C]
    const float* base = sums + kd * C + c;
    float a0 = base[g0 * gstride + sl0 * 2 * C], a1 = 0.f, a2 = 0.f, a3 = 0.f;
    long g = g0 + 1;
    for (; g + 3 <= g1; g += 4) {                           // 4 independent loads in flight
        a0 += base[g * gstride]; a1 += base[(g + 1) * gstride];
        a2 += base[(g + 2) * gstride]; a3 += base[(g + 3) * gstride];
    }
    for (; g <= g1; ++g) a0 += base[g * gstride];
    (kd == 0 ? S : S2)[(long)b * C + c] = (a0 + a1) + (a2 + a3);
}
int tdm_launch_image_sums(const float* sums, float* S, float* S2, int B, int HWpix, int C, hipStream_t st) {
    TDM_REQUIRE(2 * C <= 128, "image_sums: C=%d", C);
    hipLaunchKernelGGL(image_sums_kernel, dim3(B), dim3(128), 0, st, sums, S, S2, HWpix, C);
    TDM_CHECK_LAUNCH("image_sums");
    return 0;
}
int tdm_launch_relu_bwd_tb(float* dh, const float* a1, float* S, int B, int HWpix, int C, hipStream_t st) {
    hipLaunchKernelGGL(relu_bwd_tb_kernel, dim3(B), dim3(EW_BLOCK), 0, st, dh, a1, S, HWpix, C);
    TDM_CHECK_LAUNCH("relu_bwd_tb");
    return 0;
}
int tdm_launch_time_grad(const float* S, const float* that, float* d_tw, float* d_tb, int B, int C, hipStream_t st) {
    const float* Sv[1] = {S};
    float* tw[1] = {d_tw};
    float* tbv[1] = {d_tb};
    const int Cv[1] = {C};
    return tdm_launch_time_grad_multi(Sv, tw, tbv, Cv, 1, that, B, st);
}
int tdm_launch_time_grad_multi(const float* const* S, float* const* d_tw, float* const* d_tb, const int* C, int n,
                               const float* that, int B, hipStream_t st) {
    return tdm_launch_time_grad_multi2(S, nullptr, d_tw, d_tb, nullptr, C, n, that, B, st);
}
int tdm_launch_time_grad_multi2(const float* const* S, const float* const* S2, float* const* d_tw, float* const* d_tb,
                                float* const* d_b, const int* C, int n, const float* that, int B, hipStream_t st) {
    TimeGradJobs jb{};
    TDM_REQUIRE(n >= 1 && n <= 4, "time_grad: %d jobs", n);
    for (int i = 0; i < n; ++i) {
        TDM_REQUIRE(C[i] > 0 && C[i] <= TG_BLOCK, "time_grad: C=%d must be in 1..%d", C[i], TG_BLOCK);
        jb.S[i] = S[i]; jb.d_tw[i] = d_tw[i]; jb.d_tb[i] = d_tb[i]; jb.C[i] = C[i];
        jb.S2[i] = (S2 != nullptr) ? S2[i] : nullptr;
        jb.d_b[i] = (d_b != nullptr) ? d_b[i] : nullptr;
    }
    jb.n = n;
    hipLaunchKernelGGL(time_grad_kernel, dim3(n), dim3(TG_BLOCK), 0, st, jb, that, B);
    TDM_CHECK_LAUNCH("time_grad");
    return 0;
}
int tdm_launch_split_dcat(const float* dcat, float* dout3, int B, hipStream_t st) {
    hipLaunchKernelGGL(split_dcat_kernel, dim3(ew_grid((int64_t)B * 196 * 16)), dim3(EW_BLOCK), 0, st, dcat, dout3, B);
    TDM_CHECK_LAUNCH("split_dcat");
    return 0;
}
int tdm_launch_combine_dh1(const float* dcat, const float* dp1, float* dout1, int B, hipStream_t st) {
    hipLaunchKernelGGL(combine_dh1_kernel, dim3(ew_grid((int64_t)B * 784 * 8)), dim3(EW_BLOCK), 0, st, dcat, dp1, dout1, B);
    TDM_CHECK_LAUNCH("combine_dh1");
    return 0;
}
int tdm_launch_first_wgrad(const float* x, const float* dc1, const float* dout1, float* slab, long slab_stride,
                           int w1_off, int b1_off, int ws_off, int bs_off, int B, int nslab, hipStream_t st) {
    if (dout1 != nullptr)
        hipLaunchKernelGGL(first_wgrad_kernel<true>, dim3(nslab), dim3(FW_BLOCK), 0, st, x, dc1, dout1, slab, slab_stride, w1_off,
                           b1_off, ws_off, bs_off, B);
    else
        hipLaunchKernelGGL(first_wgrad_kernel<false>, dim3(nslab), dim3(FW_BLOCK), 0, st, x, dc1, dout1, slab, slab_stride, w1_off,
                           b1_off, ws_off, bs_off, B);
    TDM_CHECK_LAUNCH("first_wgrad");
    return 0;
}
int tdm_launch_nhwc_to_nchw(const float* in, float* out, int B, int HWpix, int C, hipStream_t st) {
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_grid((int64_t)B * HWpix * C)), dim3(EW_BLOCK), 0, st, in, out, B,
                       HWpix, C);
    TDM_CHECK_LAUNCH("nhwc_to_nchw");
    return 0;
}

namespace {
__global__ __launch_bounds__(EW_BLOCK) void s16_to_nchw_kernel(const float* __restrict__ in_s16, float* __restrict__ out, int B,
                                                               int HWpix, int C) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)B * HWpix * C4;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c4 = (int)(i % C4);
        const int64_t m = i / C4;
        const int64_t b = m / HWpix, p = m - b * HWpix;
        const float4 v = tdm_load_s16_4(in_s16, m, C, c4 * 4);
        float* o = out + (b * C + c4 * 4) * HWpix + p;
        o[0] = v.x; o[HWpix] = v.y; o[2 * (int64_t)HWpix] = v.z; o[3 * (int64_t)HWpix] = v.w;
    }
}
}  // namespace
int tdm_launch_s16_to_nchw(const float* in_s16, float* out, int B, int HWpix, int C, hipStream_t st) {
    hipLaunchKernelGGL(s16_to_nchw_kernel, dim3(ew_grid((int64_t)B * HWpix * (C / 4))), dim3(EW_BLOCK), 0, st, in_s16, out, B,
                       HWpix, C);
    TDM_CHECK_LAUNCH("s16_to_nchw");
    return 0;
}

int tdm_launch_conv_first_s16(const float* x, const float* w1, const float* b1, const float* ws, const float* bs,
                              const float* tb, int tb_stride, float* a1, unsigned char* a1m, float* a1_s16, float* s, int B,
                              hipStream_t st) {
    // contiguous pixel ranges of ~196 pixels (a quarter image) per workgroup, at most 2048 workgroups.  (Swept 512 .. 2048
    // workgroups at B = 512: 20.4 - 23.3 us, flat; with stores, FMAs and staging all switched off the launch still took
    // 18.5 us — per-workgroup prologue latency (72 weights, the input range, two barriers) at 3 waves per SIMD, not bytes:
    // tools/micro/store_patterns.hip writes the same 51 MB in 9 us.)
    const int64_t want = ((int64_t)B * 784 + 195) / 196;
    hipLaunchKernelGGL(conv_first_s16_kernel, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(EW_BLOCK), 0, st, x, w1, b1, ws, bs,
                       tb, tb_stride, a1, a1m, a1_s16, s, B);
    TDM_CHECK_LAUNCH("conv_first_s16");
    return 0;
}
int tdm_launch_pool_skip_s16(const float* h1, const float* wsk, const float* bsk, float* p1_s16, float* s2, int B,
                             hipStream_t st) {
    const int64_t groups = ((int64_t)B * 196 + 15) / 16;
    hipLaunchKernelGGL(pool_skip_s16_kernel, dim3((unsigned)(groups < 4096 ? groups : 4096)), dim3(EW_BLOCK), 0, st, h1, wsk, bsk,
                       p1_s16, s2, B);
    TDM_CHECK_LAUNCH("pool_skip_s16");
    return 0;
}
int tdm_launch_avgpool_s16(const float* in, float* out, float* out_s16, int B, int Hout, int C, hipStream_t st) {
    hipLaunchKernelGGL(avgpool_s16_kernel, dim3(ew_grid((int64_t)B * Hout * Hout * (C / 4))), dim3(EW_BLOCK), 0, st, in,
                       out, out_s16, B, Hout, C);
    TDM_CHECK_LAUNCH("avgpool_s16");
    return 0;
}
int tdm_launch_out_bwd_s16(const float* deps, const float* h4, const float* w, const unsigned char* a2, const float* h1s,
                           const float* h3s, float* dc2_s16, float* slab, long slab_stride, int w_off, int b_off,
                           int c2b_off, int skb_off, int vsk_off, int64_t M, int nslab, hipStream_t st, const float* eps,
                           const float* noise, float* deps_out, int loss_off, const float* o1_sums) {
    TDM_REQUIRE(deps != nullptr || (eps != nullptr && noise != nullptr), "out_bwd_s16: needs deps, or eps and noise");
    TDM_REQUIRE((h4 != nullptr) != (o1_sums != nullptr) && (o1_sums == nullptr || deps != nullptr),
                "out_bwd_s16: either h4 (output conv gradients taken here) or the forward epilogue's partial rows + deps");
    TDM_REQUIRE(h1s != nullptr && h3s != nullptr && (M % 784) == 0, "out_bwd_s16: needs the S16 concat sources of whole 28x28 images");
    hipLaunchKernelGGL(out_bwd_s16_kernel, dim3(nslab), dim3(EW_BLOCK), 0, st, deps, h4, w, a2, h1s, h3s, dc2_s16,
                       slab, slab_stride, w_off, b_off, c2b_off, skb_off, vsk_off, M, eps, noise, deps_out, 2.0f / (float)M, loss_off,
                       o1_sums, (int)((M + 31) / 32));
    TDM_CHECK_LAUNCH("out_bwd_s16");
    return 0;
}
int tdm_launch_group_sums(const GroupSumJobs& jb, const float* that, int B, float* slab, long slab_stride, int nslab,
                          hipStream_t st) {
    const int njobs = jb.njobs > 0 ? jb.njobs : 4 - jb.job0;
    TDM_REQUIRE(jb.job0 >= 0 && njobs >= 1 && jb.job0 + njobs <= TDM_GS_JOBS, "group_sums: jobs %d .. +%d", jb.job0, njobs);
    for (int i = jb.job0; i < jb.job0 + njobs; ++i)
        TDM_REQUIRE(jb.gs[i] != nullptr && jb.C[i] % 16 == 0 && EW_BLOCK % (jb.C[i] / 4) == 0, "group_sums: job %d", i);
    hipLaunchKernelGGL(group_sums_kernel, dim3(nslab, njobs), dim3(EW_BLOCK), 0, st, jb, that, B, slab, slab_stride);
    TDM_CHECK_LAUNCH("group_sums");
    return 0;
}
int tdm_launch_relu_mask_s16(const float* dout, const unsigned char* a, float* dc_s16, float* slab, long slab_stride,
                             int b_masked_off, int b_unmasked_off, int64_t M, int C, int nslab, hipStream_t st) {
    TDM_REQUIRE(C % 16 == 0 && EW_BLOCK % (C / 4) == 0, "relu_mask_s16: C=%d", C);
    hipLaunchKernelGGL(relu_mask_s16_kernel, dim3(nslab), dim3(EW_BLOCK), 0, st, dout, a, dc_s16, slab, slab_stride,
                       b_masked_off, b_unmasked_off, M, C);
    TDM_CHECK_LAUNCH("relu_mask_s16");
    return 0;
}
int tdm_launch_split_dcat_mask_s16(const float* dcat, const unsigned char* a2, float* dout3, float* dc_s16, float* slab,
                                   long slab_stride, int b_masked_off, int B, int nslab, hipStream_t st) {
    hipLaunchKernelGGL(split_dcat_mask_s16_kernel, dim3(nslab), dim3(EW_BLOCK), 0, st, dcat, a2, dout3, dc_s16, slab,
                       slab_stride, b_masked_off, B);
    TDM_CHECK_LAUNCH("split_dcat_mask_s16");
    return 0;
}
int tdm_launch_combine_dh1_mask_s16(const float* dcat, const float* dp1, const unsigned char* a2, float* dout1, float* dc_s16,
                                    float* slab, long slab_stride, int b_masked_off, int B, int nslab, hipStream_t st,
                                    const float* x, int ws_off, int bs_off) {
    TDM_REQUIRE(dout1 != nullptr || x != nullptr, "combine_dh1_mask_s16: dout1 may only be dropped with the skip gradients fused");
    hipLaunchKernelGGL(combine_dh1_mask_s16_kernel, dim3(nslab), dim3(EW_BLOCK), 0, st, dcat, dp1, a2, dout1, dc_s16, slab,
                       slab_stride, b_masked_off, B, x, ws_off, bs_off);
    TDM_CHECK_LAUNCH("combine_dh1_mask_s16");
    return 0;
}

// ReLU byte masks <-> one 0/1 byte per element in NCHW (test accessor: teacher-forced masks)
namespace {
__global__ __launch_bounds__(EW_BLOCK) void mask_io_kernel(unsigned char* __restrict__ packed, unsigned char* __restrict__ nchw,
                                                           int B, int HWpix, int C, int write) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)B * HWpix * C4;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c4 = (int)(i % C4);
        const int64_t m = i / C4;
        const int64_t b = m / HWpix, p = m - b * HWpix;
        unsigned char* e = nchw + (b * C + c4 * 4) * HWpix + p;
        if (write) {
            packed[i] = (unsigned char)((e[0] ? 1 : 0) | (e[HWpix] ? 2 : 0) | (e[2 * (int64_t)HWpix] ? 4 : 0) | (e[3 * (int64_t)HWpix] ? 8 : 0));
        } else {
            const unsigned mk = packed[i];
            e[0] = mk & 1u; e[HWpix] = (mk >> 1) & 1u; e[2 * (int64_t)HWpix] = (mk >> 2) & 1u; e[3 * (int64_t)HWpix] = (mk >> 3) & 1u;
        }
    }
}
}  // namespace
int tdm_launch_mask_io(unsigned char* packed, unsigned char* nchw, int B, int HWpix, int C, int write, hipStream_t st) {
    hipLaunchKernelGGL(mask_io_kernel, dim3(ew_grid((int64_t)B * HWpix * (C / 4))), dim3(EW_BLOCK), 0, st, packed, nchw, B,
                       HWpix, C, write);
    TDM_CHECK_LAUNCH("mask_io");
    return 0;
}

// ------------------------------- C ABI ---------------------------------------
extern "C" {

int tdm_q_sample_f32(const float* x0, const float* noise, const int64_t* t, const float* sqrt_acp,
                     const float* sqrt_1m_acp, float* out, int64_t B, int64_t inner, void* stream) {
    TDM_REQUIRE(B > 0 && inner > 0, "q_sample: empty input (B=%lld inner=%lld)", (long long)B, (long long)inner);
    const int64_t work = (inner & 3) == 0 ? B * inner / 4 : B * inner;
    hipLaunchKernelGGL(q_sample_kernel, dim3(ew_grid(work)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x0, noise, t,
                       sqrt_acp, sqrt_1m_acp, out, B, inner);
    TDM_CHECK_LAUNCH("q_sample");
    return 0;
}

int tdm_scale_by_table_f32(const float* x, const int64_t* t, const float* tab, float* out, int64_t B, int64_t inner,
                           void* stream) {
    TDM_REQUIRE(x && t && tab && out && B > 0 && inner > 0, "scale_by_table: bad arguments");
    hipLaunchKernelGGL(scale_by_table_kernel, dim3(ew_grid(B * inner)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x, t, tab, out, B,
                       inner);
    TDM_CHECK_LAUNCH("scale_by_table");
    return 0;
}

int tdm_p_sample_update_f32(const float* x, const float* eps, const float* noise, const float* tab_recip,
                            const float* tab_eps, const float* tab_sigma, int t_index, float* out, int64_t n,
                            void* stream) {
    TDM_REQUIRE(n > 0, "p_sample_update: empty input");
    TDM_REQUIRE(t_index >= 0 && t_index < TDM_TIMESTEPS, "p_sample_update: t_index %d out of range", t_index);
    const int64_t work = (n & 3) == 0 ? n / 4 : n;
    hipLaunchKernelGGL(p_update_kernel, dim3(ew_grid(work)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x, eps, noise,
                       tab_recip, tab_eps, tab_sigma, (const int64_t*)nullptr, t_index, noise != nullptr ? 1 : 0, out,
                       (int64_t)1, n);
    TDM_CHECK_LAUNCH("p_sample_update");
    return 0;
}

int tdm_p_sample_update_pert_f32(const float* x, const float* eps, const float* noise, const float* tab_recip,
                                 const float* tab_eps, const float* tab_sigma, const int64_t* t, int add_noise,
                                 float* out, int64_t B, int64_t inner, void* stream) {
    TDM_REQUIRE(B > 0 && inner > 0, "p_sample_update_pert: empty input");
    TDM_REQUIRE(t != nullptr, "p_sample_update_pert: t is NULL");
    const int64_t work = (inner & 3) == 0 ? B * inner / 4 : B * inner;
    hipLaunchKernelGGL(p_update_kernel, dim3(ew_grid(work)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x, eps, noise,
                       tab_recip, tab_eps, tab_sigma, t, 0, add_noise, out, B, inner);
    TDM_CHECK_LAUNCH("p_sample_update_pert");
    return 0;
}

int tdm_to_unit_u8_f32(const float* x, float* x01, uint8_t* u8, int64_t n, void* stream) {
    TDM_REQUIRE(n > 0, "to_unit_u8: empty input");
    hipLaunchKernelGGL(to_unit_u8_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x, x01, u8, n);
    TDM_CHECK_LAUNCH("to_unit_u8");
    return 0;
}

int tdm_mse_fwd_bwd_f32(const float* pred, const float* target, float* loss_out, float* dpred, float* scratch, int64_t n,
                        void* stream) {
    TDM_REQUIRE(n > 0, "mse: empty input");
    int grid = ew_grid(n);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(mse_kernel, dim3(grid), dim3(EW_BLOCK), 0, (hipStream_t)stream, pred, target, dpred, scratch, n,
                       2.0f / (float)n);
    TDM_CHECK_LAUNCH("mse");
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(EW_BLOCK), 0, (hipStream_t)stream, scratch, grid, loss_out,
                       1.0f / (float)n);
    TDM_CHECK_LAUNCH("mse_final");
    return 0;
}

int tdm_adamw_flat_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                       float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
    TDM_REQUIRE(n > 0 && step >= 1, "adamw: n=%lld step=%lld", (long long)n, (long long)step);
    // scalar prologue in double, as Python floats are in torch's _single_tensor_adamw
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const double step_size = (double)lr / bc1;
    const double bc2_sqrt = sqrt(bc2);
    hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, p, g, m, v, n,
                       (float)(1.0 - (double)lr * (double)weight_decay), (float)(1.0 - (double)beta1), beta2,
                       (float)(1.0 - (double)beta2), (float)step_size, (float)bc2_sqrt, eps, grad_scale);
    TDM_CHECK_LAUNCH("adamw");
    return 0;
}

}  // extern "C"
