// Weight pre-pack of the bf16x3 convolution kernels (conv_s16.hip): fp32 HWIO -> bf16 hi / lo planes in MFMA fragment
// order, for the forward and for the transposed convolution, once per call, so that staging a K chunk of weights is a linear
// 16-byte copy and the conv kernel itself is direction-agnostic.  (The kernels that split their ACTIVATIONS while staging —
// "conv mode 1", round 1 — were superseded by the pre-split S16 pipeline and are no longer built: DESIGN.md section 4.)
#include "tdm_common.h"
#include "tdm_timebias.h"

namespace {

// ---------------------------------------------------------------------------
// weight pre-pack: fp32 HWIO -> bf16 hi/lo in MFMA B-fragment order
//   forward : B[k = ci][n = co] = W[tap][ci][co]            chunks over ci
//   dgrad   : B[k = co][n = ci] = W[8-tap][ci][co] (3x3)    chunks over co
// element (chunk, tap, nt, part, lane, j):  n = nt*32 + (lane&31),  k = chunk*16 + 8*(lane>>5) + j
// ---------------------------------------------------------------------------
__device__ __forceinline__ void pack_body(const float* __restrict__ P, const PackDesc& d, unsigned short* __restrict__ out, int bx,
                                          int nbx) {
    if (d.phase == 1) {   // phase form of an up-sampled source's forward weights (PackDesc's comment; conv_s16.hip "PH")
        const int NTp = d.cout / 32;
        const int total = (d.kuse / 16) * 16 * NTp * 512;
        for (int e = bx * 256 + threadIdx.x; e < total; e += nbx * 256) {
            const int jj = e & 7;
            const int lane = (e >> 3) & 63;
            int r = e >> 9;
            const int nt = r % NTp; r /= NTp;
            const int tap16 = r & 15;
            const int chunk = r >> 4;
            const int py = tap16 >> 3, px = (tap16 >> 2) & 1, ap = (tap16 >> 1) & 1, bp = tap16 & 1;
            const int ky0 = ap ? (py ? 2 : 1) : 0, ky1 = ap ? 2 : (py ? 1 : 0);
            const int kx0 = bp ? (px ? 2 : 1) : 0, kx1 = bp ? 2 : (px ? 1 : 0);
            const int n = nt * 32 + (lane & 31);
            const int k = chunk * 16 + 8 * (lane >> 5) + jj;
            float x = 0.f;
            for (int ky = ky0; ky <= ky1; ++ky)
                for (int kx = kx0; kx <= kx1; ++kx) x += P[d.src_off + (long)((ky * 3 + kx) * d.cin + k) * d.cout + n];
            const __bf16 hi = (__bf16)x;
            const __bf16 lo = (__bf16)(x - (float)hi);
            const long base = d.dst_off + ((long)((chunk * 16 + tap16) * NTp + nt) * 2) * 512 + lane * 8 + jj;
            out[base] = __builtin_bit_cast(unsigned short, hi);
            out[base + 512] = __builtin_bit_cast(unsigned short, lo);
        }
        return;
    }
    if (d.phase == 2) {   // data gradient of an up-sampled source at SOURCE resolution (PackDesc's comment; conv_s16.hip "S2D")
        const int NTp = d.kuse / 32, KC = d.cout / 16;
        const int total = 4 * KC * 4 * NTp * 512;
        for (int e = bx * 256 + threadIdx.x; e < total; e += nbx * 256) {
            const int jj = e & 7;
            const int lane = (e >> 3) & 63;
            int r = e >> 9;
            const int nt = r % NTp; r /= NTp;
            const int tap4 = r & 3; r >>= 2;
            const int chunk = r;                       // (2 p + q) * KC + kc
            const int kc = chunk % KC, pq = chunk / KC, p = pq >> 1, q = pq & 1;
            const int u = 2 * ((tap4 >> 1) - p) + p + 1, v = 2 * ((tap4 & 1) - q) + q + 1;   // dy = a - p, dx = b - q
            const int ky0 = u == 0 ? 2 : (u == 1 ? 1 : 0), ky1 = u == 0 ? 2 : (u == 1 ? 2 : (u == 2 ? 1 : 0));
            const int kx0 = v == 0 ? 2 : (v == 1 ? 1 : 0), kx1 = v == 0 ? 2 : (v == 1 ? 2 : (v == 2 ? 1 : 0));
            const int n = nt * 32 + (lane & 31);
            const int k = kc * 16 + 8 * (lane >> 5) + jj;
            float x = 0.f;
            for (int ky = ky0; ky <= ky1; ++ky)
                for (int kx = kx0; kx <= kx1; ++kx) x += P[d.src_off + (long)((ky * 3 + kx) * d.cin + n) * d.cout + k];
            const __bf16 hi = (__bf16)x;
            const __bf16 lo = (__bf16)(x - (float)hi);
            const long base = d.dst_off + ((long)((chunk * 4 + tap4) * NTp + nt) * 2) * 512 + lane * 8 + jj;
            out[base] = __builtin_bit_cast(unsigned short, hi);
            out[base + 512] = __builtin_bit_cast(unsigned short, lo);
        }
        return;
    }
    const int K = d.dgrad ? d.cout : d.cin;       // contraction length
    const int Nn = d.dgrad ? (d.nuse > 0 ? d.nuse : d.cin) : d.cout;      // output channels of this direction
    const int NT = Nn / 32;
    const int total = (K / 16) * d.taps * NT * 512;   // (hi, lo) pairs
    for (int e = bx * 256 + threadIdx.x; e < total; e += nbx * 256) {
        const int jj = e & 7;
        const int lane = (e >> 3) & 63;
        int r = e >> 9;
        const int nt = r % NT; r /= NT;
        const int tap = r % d.taps; r /= d.taps;
        const int chunk = r;
        const int n = nt * 32 + (lane & 31);
        const int k = chunk * 16 + 8 * (lane >> 5) + jj;
        float x;
        if (!d.dgrad) x = P[d.src_off + (long)(tap * d.cin + k) * d.cout + n];
        else x = P[d.src_off + (long)((d.taps == 9 ? 8 - tap : 0) * d.cin + d.n0 + n) * d.cout + k];
        const __bf16 hi = (__bf16)x;
        const __bf16 lo = (__bf16)(x - (float)hi);
        const long base = d.dst_off + ((long)((chunk * d.taps + tap) * NT + nt) * 2) * 512 + lane * 8 + jj;
        out[base] = __builtin_bit_cast(unsigned short, hi);
        out[base + 512] = __builtin_bit_cast(unsigned short, lo);
    }
}

constexpr int PACK_WGS = 36;   // workgroups per descriptor
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ P, PackArgs pa,
                                                           unsigned short* __restrict__ out) {
    pack_body(P, pa.d[blockIdx.y], out, blockIdx.x, gridDim.x);
}

// The S16 forward's two prologue launches as one (neither reads what the other writes): workgroups [0, 36 n) pack the
// weights, the rest compute the timestep biases.
__global__ __launch_bounds__(256) void pack_timebias_kernel(const float* __restrict__ P, PackArgs pa,
                                                            unsigned short* __restrict__ out, TimebiasArgs ta) {
    const int npack = PACK_WGS * pa.n;
    if ((int)blockIdx.x < npack) pack_body(P, pa.d[blockIdx.x / PACK_WGS], out, blockIdx.x % PACK_WGS, PACK_WGS);
    else tdm_timebias_body(ta, blockIdx.x - npack, gridDim.x - npack);
}

}  // namespace

int tdm_launch_pack(const float* params, const PackArgs& pa, unsigned short* out, hipStream_t st) {
    TDM_REQUIRE(pa.n >= 1 && pa.n <= TDM_MAX_PACK, "pack: %d descriptors", pa.n);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(PACK_WGS, pa.n), dim3(256), 0, st, params, pa, out);
    TDM_CHECK_LAUNCH("pack_weights");
    return 0;
}

int tdm_launch_pack_timebias(const float* params, const PackArgs& pa, unsigned short* out, const int64_t* t, const int* te_w_off,
                             const int* te_b_off, float* that, float* tb, int B, int64_t* bump, float* u96, int skw4_off,
                             int outw_off, hipStream_t st) {
    TDM_REQUIRE(pa.n >= 1 && pa.n <= TDM_MAX_PACK, "pack: %d descriptors", pa.n);
    TimebiasArgs ta{};
    for (int i = 0; i < 4; ++i) { ta.o.w[i] = te_w_off[i]; ta.o.b[i] = te_b_off[i]; }
    ta.o.skw4 = skw4_off; ta.o.outw = outw_off;
    ta.t = t; ta.params = params; ta.that = that; ta.tb = tb; ta.B = B; ta.bump = bump; ta.u96 = u96;
    const long nb = ((long)B * 192 + 255) / 256;
    const int ntb = (int)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
    hipLaunchKernelGGL(pack_timebias_kernel, dim3(PACK_WGS * pa.n + ntb), dim3(256), 0, st, params, pa, out, ta);
    TDM_CHECK_LAUNCH("pack_timebias");
    return 0;
}
