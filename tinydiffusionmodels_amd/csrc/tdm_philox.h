// Philox4x32-10 (Salmon et al., SC'11) counter-based generator and the two draws the
// DDPM loops need: standard normals (noise of q_sample / p_sample, src/mnist.py:154-155,178)
// and uniform step indices in [0, 1000) (src/mnist.py:154).  Pure functions of
// (seed, offset, element index): nothing is stored, a hipGraph replay that advances the
// device-resident offset draws fresh numbers, and the host can evaluate the same stream
// (tests; oracle/ddpm_oracle.py restates the integer part in numpy).
//
// Stream layout: key = (seed_lo, seed_hi); counter = (idx_lo, idx_hi | kind << 28, offset_lo, offset_hi)
// where idx is the float4 index inside the tensor (normals: 4 per counter) or the sample
// index (step indices: 1 per counter) and kind separates the two uses.
#pragma once
#include <stdint.h>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define TDM_HD __host__ __device__ __forceinline__
#else
#define TDM_HD inline
#endif

struct tdm_u32x4 { uint32_t x, y, z, w; };

TDM_HD uint32_t tdm_mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }

TDM_HD tdm_u32x4 tdm_philox4x32_10(tdm_u32x4 c, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r > 0) { k0 += W0; k1 += W1; }
        const uint32_t hi0 = tdm_mulhi32(M0, c.x), lo0 = M0 * c.x;
        const uint32_t hi1 = tdm_mulhi32(M1, c.z), lo1 = M1 * c.z;
        tdm_u32x4 n;
        n.x = hi1 ^ c.y ^ k0; n.y = lo1; n.z = hi0 ^ c.w ^ k1; n.w = lo0;
        c = n;
    }
    return c;
}

enum { TDM_PHILOX_KIND_NORMAL = 0, TDM_PHILOX_KIND_STEP = 1 };

TDM_HD tdm_u32x4 tdm_philox_at(uint64_t seed, uint64_t offset, uint64_t idx, int kind) {
    tdm_u32x4 c;
    c.x = (uint32_t)idx;
    c.y = ((uint32_t)(idx >> 32) & 0x0FFFFFFFu) | ((uint32_t)kind << 28);
    c.z = (uint32_t)offset;
    c.w = (uint32_t)(offset >> 32);
    return tdm_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// step index in [0, nsteps): multiply-shift of one 32-bit draw (bias < nsteps / 2^32)
TDM_HD int64_t tdm_philox_step(uint64_t seed, uint64_t offset, uint64_t sample, uint32_t nsteps) {
    const tdm_u32x4 r = tdm_philox_at(seed, offset, sample, TDM_PHILOX_KIND_STEP);
    return (int64_t)tdm_mulhi32(r.x, nsteps);
}

#ifdef __HIPCC__
// four N(0,1) draws of float4 index idx: two Box-Muller pairs over uniforms in (0, 1]
__device__ __forceinline__ float4 tdm_philox_normal4(uint64_t seed, uint64_t offset, uint64_t idx) {
    const tdm_u32x4 r = tdm_philox_at(seed, offset, idx, TDM_PHILOX_KIND_NORMAL);
    const float S = 2.3283064365386963e-10f, H = 1.1641532182693481e-10f;   // 2^-32, 2^-33
    const float u0 = fmaf((float)r.x, S, H), u1 = fmaf((float)r.y, S, H);
    const float u2 = fmaf((float)r.z, S, H), u3 = fmaf((float)r.w, S, H);
    const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    float s0, c0, s1, c1;
    sincospif(2.0f * u1, &s0, &c0);
    sincospif(2.0f * u3, &s1, &c1);
    return make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
}
#endif
