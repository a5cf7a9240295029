// Implicit-GEMM 3x3 / 1x1 convolution, transposed convolution and weight
// gradient for the SimpleUNet residual blocks (src/mnist.py:45-61), written
// for gfx950: fp32 MFMA (v_mfma_f32_32x32x2_f32 — exact fp32, bitwise an
// fmaf chain), 64-lane wavefronts, LDS-staged haloed input tiles.
//
// Data layout.  Activations are NHWC fp32.  A workgroup owns TILE_PX = 256
// consecutive pixels of the flattened (b, y, x) index — the GEMM M dimension —
// whatever image rows or image boundaries that range crosses.  Its input is
// staged in "padded tall" coordinates: image b occupies padded rows
// b*(H+2) .. b*(H+2)+H+1 (first and last are zero rows), columns 0 and W+1 are
// zero, so every tap of every pixel is an unconditional LDS read at a
// compile-time offset from the pixel's own position.
//
// GEMM mapping (forward): M = pixels, N = Cout, K = taps x Cin, walked in
// chunks of CK = 16 input channels.  Per MFMA (32x32x2): lane l supplies
// A[pixel l&31][k = l>>5] and B[k = l>>5][cout l&31]; one ds_read_b128 per
// lane fetches the 4 channels {kg*8 + 4*(l>>5) + s, s=0..3} that feed 4
// consecutive MFMAs.  The LDS pixel stride is 20 floats = 5 x 16 B (an odd
// number of 16-B slots), which makes those reads bank-conflict free.
#include "tdm_common.h"
#include <cstdlib>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int CK = 16;        // input channels per K chunk
constexpr int CPAD = 20;      // LDS floats per staged pixel (16 + 4 pad)
constexpr int TILE_PX = 256;  // pixels per workgroup: 4 waves x 2 M-tiles x 32

template <int HW> struct Geo;
template <> struct Geo<28> { static constexpr int H = 28, W = 28, HP = 30, WP = 30, NR = 15; };
template <> struct Geo<14> { static constexpr int H = 14, W = 14, HP = 16, WP = 16, NR = 26; };

// Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a
// contiguous range of tiles so that neighbouring tiles' halo rows hit its L2.
// Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

template <int HW>
__device__ __forceinline__ int padded_row(int m) {  // padded-tall row of flat pixel m
    using G = Geo<HW>;
    const int b = m / (G::H * G::W);
    const int y = (m - b * (G::H * G::W)) / G::W;
    return b * G::HP + y + 1;
}

// Stage rows [PR0, PR0+nrows) x all padded columns x NC4 float4 channel groups
// of one source into LDS at `tile` (pixel stride PSTRIDE floats).
template <int HW, int NC4, int PSTRIDE>
__device__ __forceinline__ void stage_input(float* tile, const ConvSrc& s, int chan0, int PR0, int nrows, int B,
                                            int tid) {
    using G = Geo<HW>;
    const int nelem = nrows * G::WP * NC4;
    const int up = s.up;
    const int Hs = G::H >> up, Ws = G::W >> up;
    for (int e = tid; e < nelem; e += 256) {
        const int c4 = e % NC4;
        const int pos = e / NC4;
        const int lr = pos / G::WP;
        const int pc = pos - lr * G::WP;
        const int PR = PR0 + lr;
        const int b = PR / G::HP;
        const int py = PR - b * G::HP;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (py >= 1 && py <= G::H && pc >= 1 && pc <= G::W && b < B) {
            const int y = (py - 1) >> up, x = (pc - 1) >> up;
            const long idx = ((long)(b * Hs + y) * Ws + x) * s.C + s.c0 + chan0 + c4 * 4;
            v = *reinterpret_cast<const float4*>(s.ptr + idx);
            if (s.tb != nullptr) {
                const float4 t4 = *reinterpret_cast<const float4*>(s.tb + (long)b * s.tb_stride + chan0 + c4 * 4);
                v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w;
            }
        }
        *reinterpret_cast<float4*>(tile + pos * PSTRIDE + c4 * 4) = v;
    }
}

// ---------------------------------------------------------------------------
// forward convolution / transposed convolution
// ---------------------------------------------------------------------------
template <int HW, int NT, bool DGRAD>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
    using G = Geo<HW>;
    constexpr int N = NT * 32;
    constexpr int TILE_F = G::NR * G::WP * CPAD;
    extern __shared__ float4 smem4[];
    float* tile = reinterpret_cast<float*>(smem4);
    float* wl = tile + TILE_F;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int Mtot = a.B * G::H * G::W;
    const int m0 = t * TILE_PX;
    const int mlast = min(m0 + TILE_PX - 1, Mtot - 1);
    const int PR0 = padded_row<HW>(m0) - 1;
    const int nrows = padded_row<HW>(mlast) - PR0 + 2;

    int aoff[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = min(m0 + (wave * 2 + mt) * 32 + j, Mtot - 1);
        const int b = m / (G::H * G::W);
        const int rem = m - b * (G::H * G::W);
        const int y = rem / G::W, x = rem - y * G::W;
        const int lr = b * G::HP + y + 1 - PR0;
        aoff[mt] = (lr * G::WP + x + 1) * CPAD + h * 4;
    }

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    for (int si = 0; si < a.nsrc; ++si) {
        const ConvSrc s = a.src[si];
        const int taps = s.taps;
        for (int kc = 0; kc < s.nch; kc += CK) {
            __syncthreads();  // everyone is done reading the previous chunk
            stage_input<HW, CK / 4, CPAD>(tile, s, kc, PR0, nrows, a.B, tid);
            if (!DGRAD) {
                // wl[(tap*16 + r)*N + n] = w[(tap*w_rows + w_r0 + kc + r)*N + n]
                const int nelem = taps * CK * (N / 4);
                for (int e = tid; e < nelem; e += 256) {
                    const int n4 = e % (N / 4);
                    const int row = e / (N / 4);
                    const int tap = row >> 4, r = row & 15;
                    const float4 v = *reinterpret_cast<const float4*>(
                        s.w + (long)(tap * s.w_rows + s.w_r0 + kc + r) * N + n4 * 4);
                    *reinterpret_cast<float4*>(wl + row * N + n4 * 4) = v;
                }
            } else {
                // wl[(tap*N + n)*CPAD + r] = w[((8-tap)*w_rows + n)*w_cols + kc + r]
                const int nelem = taps * N * 4;
                for (int e = tid; e < nelem; e += 256) {
                    const int r4 = e & 3;
                    const int rn = e >> 2;
                    const int tap = rn / N;
                    const int n = rn - tap * N;
                    const int ftap = (taps == 9) ? 8 - tap : 0;
                    const float4 v = *reinterpret_cast<const float4*>(
                        s.w + (long)(ftap * s.w_rows + n) * s.w_cols + kc + r4 * 4);
                    *reinterpret_cast<float4*>(wl + rn * CPAD + r4 * 4) = v;
                }
            }
            __syncthreads();

            auto do_tap = [&](int tap, int toff) {
#pragma unroll
                for (int kg = 0; kg < 2; ++kg) {
                    float4 av[2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        av[mt] = *reinterpret_cast<const float4*>(tile + aoff[mt] + toff + kg * 8);
                    float bv[NT][4];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if (!DGRAD) {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                bv[nt][q] = wl[(tap * CK + kg * 8 + h * 4 + q) * N + nt * 32 + j];
                        } else {
                            const float4 b4 =
                                *reinterpret_cast<const float4*>(wl + (tap * N + nt * 32 + j) * CPAD + kg * 8 + h * 4);
                            bv[nt][0] = b4.x; bv[nt][1] = b4.y; bv[nt][2] = b4.z; bv[nt][3] = b4.w;
                        }
                    }
                    const float a0[4] = {av[0].x, av[0].y, av[0].z, av[0].w};
                    const float a1[4] = {av[1].x, av[1].y, av[1].z, av[1].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], bv[nt][q], acc[0][nt], 0, 0, 0);
                            acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], bv[nt][q], acc[1][nt], 0, 0, 0);
                        }
                }
            };
            if (taps == 9) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) do_tap(tap, ((tap / 3 - 1) * G::WP + (tap % 3 - 1)) * CPAD);
            } else {
                do_tap(0, 0);
            }
        }
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = nt * 32 + j;
            const float bz = (a.bias != nullptr) ? a.bias[co] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wave * 2 + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < Mtot) {
                    float v = acc[mt][nt][r] + bz;
                    if (a.r1_x != nullptr) v = fmaf(a.r1_x[m], a.r1_w[co], v);   // a rank-one term x[m] * w[co] (rb4.skip's share of d cat)
                    if (a.relu) v = (v < 0.f) ? 0.f : v;
                    const long o = (long)m * N + co;
                    if (a.aux != nullptr) a.aux[o] = v;
                    if (a.res != nullptr) v += a.res[o];
                    a.out[o] = v;
                }
            }
        }
}

template <int HW, int NT, bool DGRAD>
int launch_conv_t(const ConvArgs& a, hipStream_t st) {
    using G = Geo<HW>;
    constexpr int N = NT * 32;
    constexpr size_t lds = (size_t)(G::NR * G::WP * CPAD + (DGRAD ? 9 * N * CPAD : 9 * CK * N)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<HW, NT, DGRAD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            tdm_set_error("conv_mfma: hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
            return 100 + (int)e;
        }
        attr_set = true;
    }
    const long Mtot = (long)a.B * G::H * G::W;
    const int ntiles = (int)((Mtot + TILE_PX - 1) / TILE_PX);
    hipLaunchKernelGGL((conv_mfma_kernel<HW, NT, DGRAD>), dim3(ntiles), dim3(256), lds, st, a);
    TDM_CHECK_LAUNCH("conv_mfma");
    return 0;
}

// ---------------------------------------------------------------------------
// weight gradient: dW[tap][ci][co] = sum_p A[p + tap][ci] * G[p][co]
// GEMM: M = ci (32 per workgroup), N = co (32 per workgroup), K = pixels.
// Workgroup (bx, by): by selects the (ci tile, co tile); bx walks pixel tiles
// bx, bx+gridDim.x, ... and keeps its 9 tap accumulators in registers; its 4
// waves split each tile's pixels, are summed through LDS at the end and the
// result is stored to slab bx (summed over slabs by reduce_slabs_kernel —
// deterministic, no atomics).
// ---------------------------------------------------------------------------
template <int HW>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradArgs a) {
    using G = Geo<HW>;
    constexpr int TILE_F = G::NR * G::WP * 32;
    extern __shared__ float4 smem4[];
    float* tile = reinterpret_cast<float*>(smem4);
    int* pixoff = reinterpret_cast<int*>(tile + TILE_F);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int ci_tile = blockIdx.y % a.nci, co_tile = blockIdx.y / a.nci;
    const int ci0 = ci_tile * 32, co0 = co_tile * 32;
    const int Mtot = a.B * G::H * G::W;
    const int taps = a.a.taps;

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float bsum = 0.f;

    for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        const int m0 = t * TILE_PX;
        const int mlast = min(m0 + TILE_PX - 1, Mtot - 1);
        const int PR0 = padded_row<HW>(m0) - 1;
        const int nrows = padded_row<HW>(mlast) - PR0 + 2;
        __syncthreads();  // previous tile fully consumed
        {
            const int m = min(m0 + tid, Mtot - 1);
            const int b = m / (G::H * G::W);
            const int rem = m - b * (G::H * G::W);
            const int y = rem / G::W, x = rem - y * G::W;
            pixoff[tid] = ((b * G::HP + y + 1 - PR0) * G::WP + x + 1) * 32;
        }
        stage_input<HW, 8, 32>(tile, a.a, ci0, PR0, nrows, a.B, tid);
        // this wave's 64 pixels: lane (h, j) holds G[pixel 2*step + h][co0 + j]
        float gv[32];
#pragma unroll
        for (int st = 0; st < 32; ++st) {
            const int m = m0 + wave * 64 + 2 * st + h;
            gv[st] = (m < Mtot) ? a.g[(long)m * a.Cout + co0 + j] : 0.f;
        }
        __syncthreads();
        if (taps == 9) {
#pragma unroll
            for (int st = 0; st < 32; ++st) {
                const int po = pixoff[wave * 64 + 2 * st + h];
                const float g = gv[st];
                bsum += g;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float av = tile[po + ((tap / 3 - 1) * G::WP + (tap % 3 - 1)) * 32 + j];
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, g, acc[tap], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int st = 0; st < 32; ++st) {
                const int po = pixoff[wave * 64 + 2 * st + h];
                const float g = gv[st];
                bsum += g;
                const float av = tile[po + j];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, g, acc[0], 0, 0, 0);
            }
        }
    }

    // cross-wave reduction through LDS, one tap at a time
    float* red = tile;  // 4 waves x 1024 floats
    float* slab = a.slab + (long)blockIdx.x * a.slab_stride;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        if (tap < taps) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[tap][r];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = tid + 256 * q;
                const float sum = ((red[idx] + red[1024 + idx]) + red[2048 + idx]) + red[3072 + idx];
                const int r = idx >> 6, ln = idx & 63;
                const int ci = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
                const int co = ln & 31;
                slab[a.w_off + (long)(tap * a.a.w_rows + a.a.w_r0 + ci0 + ci) * a.Cout + co0 + co] = sum;
            }
        }
    }
    if (a.b_off >= 0 && ci_tile == 0) {
        bsum += __shfl_xor(bsum, 32);
        __syncthreads();
        if (lane < 32) red[wave * 32 + lane] = bsum;
        __syncthreads();
        if (tid < 32) slab[a.b_off + co0 + tid] = ((red[tid] + red[32 + tid]) + red[64 + tid]) + red[96 + tid];
    }
}

template <int HW>
int launch_wgrad_t(const WgradArgs& a, int nslab, hipStream_t st) {
    using G = Geo<HW>;
    constexpr size_t lds = (size_t)(G::NR * G::WP * 32) * sizeof(float) + TILE_PX * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_mfma_kernel<HW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            tdm_set_error("wgrad_mfma: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return 100 + (int)e;
        }
        attr_set = true;
    }
    const int nco = a.Cout / 32;
    hipLaunchKernelGGL((wgrad_mfma_kernel<HW>), dim3(nslab, a.nci * nco), dim3(256), lds, st, a);
    TDM_CHECK_LAUNCH("wgrad_mfma");
    return 0;
}

// out[off + i] = sum over the section's slabs, in a fixed order (deterministic).  A workgroup covers EPB elements x
// 256 / EPB slab groups: thread (e, q) adds slabs q, q + NQ, ... with 8 independent loads in flight, the groups meet in LDS.
// Wide, shallow sections (the MFMA weight-gradient slabs: thousands of elements x 64-256 slabs) use 64 x 4; the narrow, DEEP
// ones (bias / time-embedding partial rows of the elementwise producers: 32-128 elements x up to 1024 rows) use 16 x 16 —
// with 64 x 4 their one or two workgroups walked 256 dependent rounds of loads each and set the launch's time (32 us for
// 90 MB; the deep sections hold 4 MB of it).
__device__ __forceinline__ int reduce_epb(int nslab) { return nslab > 256 ? 16 : 64; }
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, long stride,
                                                           ReduceArgs ra, float* __restrict__ out) {
    __shared__ float sh[256];
    // 1-D grid of exactly the workgroups the sections need (a (max length, sections) grid launched 15,000 workgroups of
    // which 2,800 had work: the empty ones cost a third of the kernel's time)
    int si = 0;
    while (si + 1 < ra.nsec && (int)blockIdx.x >= ra.blk0[si + 1]) ++si;
    const ReduceSec s = ra.sec[si];
    if (s.stride_override != 0) stride = s.stride_override;
    if (s.vec4) {   // wide shallow sections with aligned geometry: 64 element QUADS x 4 slab quarters, 16-byte loads
        __shared__ float4 sh4[256];
        const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
        const int i4 = ((int)blockIdx.x - ra.blk0[si]) * 64 + e;
        float4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool in = i4 * 4 < s.len;
        if (in) {
            const float* p = slabs + s.off + s.src_delta + (long)i4 * 4;
            int k = q;
            for (; k + 12 < s.nslab; k += 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 v = *reinterpret_cast<const float4*>(p + (long)(k + 4 * u) * stride);
                    acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
                }
            }
            for (; k < s.nslab; k += 4) {
                const float4 v = *reinterpret_cast<const float4*>(p + (long)k * stride);
                acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
            }
        }
        float4 t;
        t.x = (acc[0].x + acc[1].x) + (acc[2].x + acc[3].x); t.y = (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y);
        t.z = (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z); t.w = (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w);
        sh4[threadIdx.x] = t;
        __syncthreads();
        if (q == 0 && in) {
            float4 r = sh4[e];
            for (int qq = 1; qq < 4; ++qq) { const float4 o = sh4[qq * 64 + e]; r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w; }   // fixed order
            const float sc = s.scale != 0.f ? s.scale : 1.f;
            r.x *= sc; r.y *= sc; r.z *= sc; r.w *= sc;
            float* const d = s.dst != nullptr ? s.dst : out + s.off;
            *reinterpret_cast<float4*>(d + (long)i4 * 4) = r;
        }
        return;
    }
    const int epb = reduce_epb(s.nslab), nq = 256 / epb;
    const int e = threadIdx.x % epb, q = threadIdx.x / epb;
    {
        const int i0 = ((int)blockIdx.x - ra.blk0[si]) * epb;
        const int i = i0 + e;
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = 0.f;
        if (i < s.len) {
            const float* p = slabs + s.off + s.src_delta + i;
            int k = q;
            for (; k + 7 * nq < s.nslab; k += 8 * nq) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] += p[(long)(k + nq * u) * stride];
            }
            for (; k < s.nslab; k += nq) acc[0] += p[(long)k * stride];
        }
        __syncthreads();
        sh[q * epb + e] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        __syncthreads();
        if (q == 0 && i < s.len) {
            float r = 0.f;
            for (int qq = 0; qq < nq; ++qq) r += sh[qq * epb + e];      // fixed order
            r *= (s.scale != 0.f ? s.scale : 1.f);
            float* const d = s.dst != nullptr ? s.dst : out + s.off;
            if (s.outer_w != nullptr) {
                for (int jn = 0; jn < s.outer_n; ++jn) d[i * s.outer_n + jn] = r * s.outer_w[jn];
            } else {
                d[i] = r;
            }
        }
    }
}

}  // namespace

int tdm_launch_conv(const ConvArgs& a, int hw, int N, bool dgrad, hipStream_t st) {
    for (int i = 0; i < a.nsrc; ++i) {
        TDM_REQUIRE(a.src[i].nch % CK == 0 && a.src[i].nch > 0, "conv: source %d channel count %d not a multiple of %d", i,
                    a.src[i].nch, CK);
        TDM_REQUIRE(a.src[i].taps == 9 || a.src[i].taps == 1, "conv: taps must be 9 or 1");
        TDM_REQUIRE((a.src[i].C % 4) == 0 && (a.src[i].c0 % 4) == 0, "conv: channel alignment");
    }
    TDM_REQUIRE(a.B > 0 && (long)a.B * hw * hw * 96 < 2147483647L, "conv: batch %d out of range", a.B);
#define TDM_CONV_CASE(HWv, NTv)                                                   \
    if (hw == HWv && N == NTv * 32)                                               \
        return dgrad ? launch_conv_t<HWv, NTv, true>(a, st) : launch_conv_t<HWv, NTv, false>(a, st);
    TDM_CONV_CASE(28, 1)
    TDM_CONV_CASE(28, 2)
    TDM_CONV_CASE(28, 3)
    TDM_CONV_CASE(14, 1)
    TDM_CONV_CASE(14, 2)
#undef TDM_CONV_CASE
    tdm_set_error("conv: unsupported geometry hw=%d N=%d", hw, N);
    return 1;
}

int tdm_launch_wgrad(const WgradArgs& a, int hw, int nslab, hipStream_t st) {
    TDM_REQUIRE(a.Cout % 32 == 0 && a.nci >= 1, "wgrad: Cout %d / nci %d", a.Cout, a.nci);
    TDM_REQUIRE(a.a.taps == 9 || a.a.taps == 1, "wgrad: taps must be 9 or 1");
    TDM_REQUIRE(nslab >= 1 && nslab <= TDM_UNET_MAX_SLABS, "wgrad: nslab %d", nslab);
    TDM_REQUIRE(a.B > 0 && (long)a.B * hw * hw * 96 < 2147483647L, "wgrad: batch %d out of range", a.B);
    if (hw == 28) return launch_wgrad_t<28>(a, nslab, st);
    if (hw == 14) return launch_wgrad_t<14>(a, nslab, st);
    tdm_set_error("wgrad: unsupported hw=%d", hw);
    return 1;
}

int tdm_launch_reduce(const float* slabs, long stride, const ReduceArgs& ra, float* out, hipStream_t st) {
    TDM_REQUIRE(ra.nsec >= 1 && ra.nsec <= TDM_MAX_SECS, "reduce: nsec %d", ra.nsec);
    int maxlen = 0;
    for (int i = 0; i < ra.nsec; ++i) maxlen = ra.sec[i].len > maxlen ? ra.sec[i].len : maxlen;
    (void)maxlen;
    ReduceArgs rb = ra;
    int nb = 0;
    static const bool allow_vec4 = [] { const char* e = getenv("TDM_REDUCE_VEC4"); return e == nullptr || atoi(e) != 0; }();
    for (int i = 0; i < ra.nsec; ++i) {
        ReduceSec& s = rb.sec[i];
        const long sstride = s.stride_override != 0 ? s.stride_override : stride;
        const float* dstp = s.dst != nullptr ? s.dst : out + s.off;
        s.vec4 = allow_vec4 && s.nslab <= 256 && s.nslab >= 4 && s.len >= 4096 && (s.len & 3) == 0 && (sstride & 3) == 0 && s.outer_w == nullptr &&
                 (((uintptr_t)(slabs + s.off + s.src_delta)) & 15) == 0 && (((uintptr_t)dstp) & 15) == 0;
        const int epb = s.vec4 ? 256 : (s.nslab > 256 ? 16 : 64);
        rb.blk0[i] = nb;
        nb += (s.len + epb - 1) / epb;
    }
    rb.blk0[ra.nsec] = nb;
    TDM_REQUIRE(nb >= 1, "reduce: empty sections");
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(nb), dim3(256), 0, st, slabs, stride, rb, out);
    TDM_CHECK_LAUNCH("reduce_slabs");
    return 0;
}
