// Timestep bias of the four residual blocks (src/mnist.py:77 and :58): that = t.float()/1000; tb[b][c] = w[c]*that + bias[c]
// (32+64+64+32 = 192 channels per sample) — the body of timebias_kernel (elementwise.hip) and of the S16 pipeline's
// pack_timebias_kernel (conv_pack.hip: one launch with the weight pre-pack).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

struct TeOffs { int w[4]; int b[4]; int skw4, outw; };
struct TimebiasArgs {
    const int64_t* t; const float* params; TeOffs o; float* that; float* tb; int B;
    int64_t* bump;   // != nullptr: the Philox offset of the fused train step, advanced here (rng.hip)
    float* u96;      // != nullptr: u = W_skip(rb4) w_out (out_bwd_s16_kernel's header)
};

// bx / nbx: this workgroup's index and the number of workgroups that share the work; 256 threads
__device__ __forceinline__ void tdm_timebias_body(const TimebiasArgs& a, int bx, int nbx) {
    const float* __restrict__ params = a.params;
    if (a.bump != nullptr && bx == 0 && threadIdx.x == 0) a.bump[0] += 1;
    // u[ci] = sum_co W_skip(rb4)[ci][co] * w_out[co]: the vector that turns d(loss)/d(eps) into the skip path's share of
    // d(loss)/d(cat) (out_bwd_s16_kernel's header), 96 x 32 products by the first workgroup
    if (a.u96 != nullptr && bx == 0 && threadIdx.x < 96) {
        float acc = 0.f;
        for (int co = 0; co < 32; ++co) acc = fmaf(params[a.o.skw4 + threadIdx.x * 32 + co], params[a.o.outw + co], acc);
        a.u96[threadIdx.x] = acc;
        if (threadIdx.x < 32) a.u96[96 + threadIdx.x] = 0.f;   // 32 zeros behind it: the bias of the rank-1 epilogue that adds d[m] * u[64 + c]
    }
    const int total = a.B * 192;
    for (int i = bx * 256 + threadIdx.x; i < total; i += nbx * 256) {
        const int b = i / 192, c = i - b * 192;
        const float th = __fdiv_rn((float)a.t[b], 1000.f);
        int blk, cc;
        if (c < 32) { blk = 0; cc = c; }
        else if (c < 96) { blk = 1; cc = c - 32; }
        else if (c < 160) { blk = 2; cc = c - 96; }
        else { blk = 3; cc = c - 160; }
        a.tb[i] = fmaf(params[a.o.w[blk] + cc], th, params[a.o.b[blk] + cc]);
        if (c == 0) a.that[b] = th;
    }
}
