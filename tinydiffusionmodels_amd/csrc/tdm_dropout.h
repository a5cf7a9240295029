// Counter-based dropout masks for the transformer denoiser's train mode
// (src/shakespeare.py:106-119 builds nn.TransformerEncoderLayer(dropout=p) and an
// nn.Dropout(p) on the input; torch draws the masks from its Philox stream, which no
// other implementation can replay).  Here a mask is a pure function of
//   (seed, site, flat element index)
// so forward and backward regenerate it in registers (nothing is stored) and the
// CPU oracle recomputes the identical mask with integer arithmetic:
//   key  = hash32(seed_lo ^ hash32(seed_hi + 0x9E3779B9 * (site + 1)))
//   key' = hash32(key ^ (salt * 0x9E3779B9))   when the step is salted (graph-replayed train step), key' = key otherwise
//   u    = hash32((idx_lo ^ key) + 0x9E3779B9 * idx_hi)      (one hash per element; idx_hi = 0 below 2^32 elements)
//   keep = u >= thr,  thr = round(p * 2^32)  =>  P(keep) = 1 - p
//   y    = keep ? x * (1 / (1 - p)) : 0                (torch: x * (mask / (1 - p)))
// hash32 is the "lowbias32" integer finaliser.  Sites are numbered in the order the
// reference's forward reaches them: 0 = input dropout; for layer l: 1+4l attention
// probabilities (B,H,L,L), 2+4l dropout1 (B,L,D), 3+4l FFN dropout (B,L,F), 4+4l dropout2.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

struct DropArgs {
    uint32_t thr;   // 0 = dropout off
    uint32_t key;
    float scale;    // 1 / (1 - p)
    // Optional per-step salt in DEVICE memory, mixed into the site key by the kernels: a hipGraph-captured train step bakes
    // `key` into its kernel arguments, so what changes from replay to replay has to be read from memory (the low word of the
    // step's Philox offset, tdm_tt_loss_grad_philox_f32).  nullptr (and always on the host) = unsalted.
    // The salt goes THROUGH the hash (tdm_salted_key): the offset advances by 1 per step, and a salt merely XORed into the
    // key made step s+1's mask an XOR-translate of step s's by the constant s ^ (s+1) — the same co-drop pattern, permuted,
    // for the whole run.  Hashed, consecutive steps' keys are unrelated 32-bit values.
    const uint32_t* salt;
};

__host__ __device__ __forceinline__ uint32_t tdm_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

__host__ __device__ __forceinline__ uint32_t tdm_salted_key(uint32_t key, uint32_t salt) {
    return tdm_hash32(key ^ (salt * 0x9E3779B9U));
}

__host__ __device__ __forceinline__ bool tdm_keep(const DropArgs& d, unsigned long long idx) {
    uint32_t key = d.key;
#ifdef __HIP_DEVICE_COMPILE__
    if (d.salt != nullptr) key = tdm_salted_key(key, *d.salt);   // (uniform: one scalar load + scalar hash, hoisted out of the element loops)
#endif
    const uint32_t u = tdm_hash32(((uint32_t)idx ^ key) + 0x9E3779B9U * (uint32_t)(idx >> 32));
    return u >= d.thr;
}

inline DropArgs tdm_drop_site(float p, uint64_t seed, int site) {
    DropArgs d{};
    if (!(p > 0.f)) return d;
    double t = (double)p * 4294967296.0 + 0.5;
    if (t > 4294967295.0) t = 4294967295.0;
    d.thr = (uint32_t)t;
    if (d.thr == 0) d.thr = 1;
    d.key = tdm_hash32((uint32_t)seed ^ tdm_hash32((uint32_t)(seed >> 32) + 0x9E3779B9U * (uint32_t)(site + 1)));
    d.scale = 1.0f / (1.0f - p);
    d.salt = nullptr;
    return d;
}
