// "S16" tensors: fp32 activations / gradients stored pre-split for the bf16x3
// MFMA kernels.  An S16 tensor has the same shape and byte size as its fp32
// NHWC counterpart [M pixels][C channels] (C a multiple of 16); every 64-byte
// group of 16 channels holds  hi[16] as bf16 (32 B)  then  lo[16] as bf16 (32 B)
// with hi = bf16(x), lo = bf16(x - hi)  (x ~ hi + lo to 16 mantissa bits).
// Producers write it next to (or instead of) the fp32 tensor from values they
// already hold in registers; consumers (conv / wgrad loaders) then stage K
// chunks with plain 16-byte copies — no conversion, no address arithmetic beyond
// one add — which is what the loader-bound bf16x3 kernels need.
#pragma once
#include <hip/hip_runtime.h>

typedef __bf16 tdm_bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void tdm_split4(const float4 v, tdm_bf16x4& hi, tdm_bf16x4& lo) {
    hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
    lo[0] = (__bf16)(v.x - (float)hi[0]);
    lo[1] = (__bf16)(v.y - (float)hi[1]);
    lo[2] = (__bf16)(v.z - (float)hi[2]);
    lo[3] = (__bf16)(v.w - (float)hi[3]);
}

// store channels c..c+3 (c % 4 == 0) of pixel m of an S16 tensor with C channels
__device__ __forceinline__ void tdm_store_s16_4(float* s16, long m, int C, int c, const float4 v) {
    tdm_bf16x4 hi, lo;
    tdm_split4(v, hi, lo);
    char* base = reinterpret_cast<char*>(s16 + m * C + (c & ~15)) + (c & 15) * 2;
    *reinterpret_cast<tdm_bf16x4*>(base) = hi;
    *reinterpret_cast<tdm_bf16x4*>(base + 32) = lo;
}

// channels c..c+3 (c % 4 == 0) of pixel m of an S16 tensor, reassembled: hi + lo (>= 16 significant bits of the original)
__device__ __forceinline__ float4 tdm_load_s16_4(const float* s16, long m, int C, int c) {
    const char* base = reinterpret_cast<const char*>(s16 + m * C + (c & ~15)) + (c & 15) * 2;
    const uint2 h = *reinterpret_cast<const uint2*>(base);
    const uint2 l = *reinterpret_cast<const uint2*>(base + 32);
    float4 v;
    v.x = __uint_as_float(h.x << 16) + __uint_as_float(l.x << 16);
    v.y = __uint_as_float(h.x & 0xffff0000u) + __uint_as_float(l.x & 0xffff0000u);
    v.z = __uint_as_float(h.y << 16) + __uint_as_float(l.y << 16);
    v.w = __uint_as_float(h.y & 0xffff0000u) + __uint_as_float(l.y & 0xffff0000u);
    return v;
}
