// bf16x3 split-operand implicit-GEMM convolution for the SimpleUNet residual
// blocks (src/mnist.py:45-61) on v_mfma_f32_32x32x16_bf16.
//
// Why: in fp32 the Cin >= 32 convolutions are bound by the fp32 MFMA/VALU roof
// (157 TFLOP/s).  Splitting every fp32 operand x into
// hi = bf16(x), lo = bf16(x - hi) and accumulating  hi*hi + hi*lo + lo*hi  in
// fp32 keeps 16 mantissa bits per operand (relative error ~1e-5 per product,
// far inside the 1e-3 parity bound) at 3/16 of the fp32-MFMA cycle cost, which
// moves these layers off the compute roof towards the HBM roof.
//
// Same tiling as conv_mfma.hip: a workgroup owns 256 consecutive flat pixels,
// its haloed input is staged in "padded tall" coordinates; here each staged
// pixel is 80 bytes: 16 channels as bf16 hi (32 B), the same 16 as bf16 lo
// (32 B), 16 B pad (5 x 16 B pitch -> conflict-free ds_read_b128).  One K chunk
// = 16 channels = exactly one MFMA K step per tap.  Weights are pre-packed once
// per call by pack_weights_kernel into MFMA B-fragment order (hi and lo planes),
// for the forward and for the transposed convolution, so staging them is a
// linear 16-byte copy and the kernel itself is direction-agnostic.
#include "tdm_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int CK = 16;
constexpr int PIXB = 80;      // bytes per staged pixel
constexpr int TILE_PX = 256;

template <int HW> struct Geo;
template <> struct Geo<28> { static constexpr int H = 28, W = 28, HP = 30, WP = 30, NR = 15; };
template <> struct Geo<14> { static constexpr int H = 14, W = 14, HP = 16, WP = 16, NR = 26; };

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

template <int HW>
__device__ __forceinline__ int padded_row(int m) {
    using G = Geo<HW>;
    const int b = m / (G::H * G::W);
    const int y = (m - b * (G::H * G::W)) / G::W;
    return b * G::HP + y + 1;
}

__device__ __forceinline__ void split4(const float4 v, bf16x4& hi, bf16x4& lo) {
    hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
    lo[0] = (__bf16)(v.x - (float)hi[0]);
    lo[1] = (__bf16)(v.y - (float)hi[1]);
    lo[2] = (__bf16)(v.z - (float)hi[2]);
    lo[3] = (__bf16)(v.w - (float)hi[3]);
}

// Per-tile staging plan of one thread: element i (i < 8) is float4 #(tid + 256*i) of the
// [rows][padded cols][4 x float4] tile.  Everything that does not depend on the K chunk
// is computed once per (tile, source): the global float offset of the pixel (or -1 for a
// halo / out-of-batch position) and which of the (up to 3) images it belongs to.
struct StagePlan {
    int goff[8];   // float offset of (pixel, channel c0 + 4*(tid&3)) in the source tensor, -1 = zero
    int bsel;      // 2 bits per element: image index relative to the tile's first image
};

template <int HW>
__device__ __forceinline__ void make_plan(StagePlan& pl, const ConvSrc& s, int PR0, int nelem, int B, int tid) {
    using G = Geo<HW>;
    const int up = s.up;
    const int Hs = G::H >> up, Ws = G::W >> up;
    const int b0 = PR0 / G::HP;
    pl.bsel = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i;
        pl.goff[i] = -1;
        if (e < nelem) {
            const int pos = e >> 2;
            const int lr = pos / G::WP;
            const int pc = pos - lr * G::WP;
            const int PR = PR0 + lr;
            const int b = PR / G::HP;
            const int py = PR - b * G::HP;
            if (py >= 1 && py <= G::H && pc >= 1 && pc <= G::W && b < B) {
                const int y = (py - 1) >> up, x = (pc - 1) >> up;
                pl.goff[i] = ((b * Hs + y) * Ws + x) * s.C + s.c0 + (tid & 3) * 4;
                pl.bsel |= (b - b0) << (2 * i);
            }
        }
    }
}

template <int HW>
__device__ __forceinline__ void stage_input_split(char* tile, const ConvSrc& s, const StagePlan& pl, int chan0, int PR0,
                                                  int nelem, int B, int tid, bool skip_loads) {
    using G = Geo<HW>;
    float4 tb0 = make_float4(0.f, 0.f, 0.f, 0.f), tb1 = tb0, tb2 = tb0;
    if (s.tb != nullptr) {
        const int b0 = PR0 / G::HP;
        const float* tbp = s.tb + chan0 + (tid & 3) * 4;
        tb0 = *reinterpret_cast<const float4*>(tbp + (long)min(b0, B - 1) * s.tb_stride);
        tb1 = *reinterpret_cast<const float4*>(tbp + (long)min(b0 + 1, B - 1) * s.tb_stride);
        tb2 = *reinterpret_cast<const float4*>(tbp + (long)min(b0 + 2, B - 1) * s.tb_stride);
    }
    char* dst = tile + (tid >> 2) * PIXB + (tid & 3) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (tid + 256 * i < nelem) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!skip_loads && pl.goff[i] >= 0) {
                v = *reinterpret_cast<const float4*>(s.ptr + pl.goff[i] + chan0);
                const int sel = (pl.bsel >> (2 * i)) & 3;
                const float4 t4 = (sel == 0) ? tb0 : ((sel == 1) ? tb1 : tb2);
                v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w;
            }
            bf16x4 hi, lo;
            split4(v, hi, lo);
            *reinterpret_cast<bf16x4*>(dst + i * (64 * PIXB)) = hi;
            *reinterpret_cast<bf16x4*>(dst + i * (64 * PIXB) + 32) = lo;
        }
    }
}

template <int HW, int NT>
__global__ __launch_bounds__(256) void conv_bf16x3_kernel(ConvArgs a) {
    using G = Geo<HW>;
    constexpr int N = NT * 32;
    constexpr int TILE_B = G::NR * G::WP * PIXB;
    extern __shared__ float4 smem4[];
    char* tile = reinterpret_cast<char*>(smem4);
    char* wl = tile + TILE_B;

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
        aoff[mt] = (lr * G::WP + x + 1) * PIXB + h * 16;
    }

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const int nelem = nrows * G::WP * 4;
    for (int si = 0; si < a.nsrc; ++si) {
        const ConvSrc s = a.src[si];
        const int taps = s.taps;
        const int chunk_u16 = taps * NT * 1024;   // packed chunk: taps x NT x {hi,lo} x 64 lanes x 8
        StagePlan pl;
        make_plan<HW>(pl, s, PR0, nelem, a.B, tid);
        for (int kc = 0; kc < s.nch; kc += CK) {
            __syncthreads();
            stage_input_split<HW>(tile, s, pl, kc, PR0, nelem, a.B, tid, (a.ablate & 1) != 0);
            if (!(a.ablate & 2)) {
                const uint4* src = reinterpret_cast<const uint4*>(s.wp + (long)(s.wchunk0 + (kc >> 4)) * chunk_u16);
                uint4* dst = reinterpret_cast<uint4*>(wl);
                const int n16 = taps * NT * 128;
                for (int e = tid; e < n16; e += 256) dst[e] = src[e];
            }
            __syncthreads();

            auto do_tap = [&](int tap, int toff) {
                bf16x8 ah[2], al[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    ah[mt] = *reinterpret_cast<const bf16x8*>(tile + aoff[mt] + toff);
                    al[mt] = *reinterpret_cast<const bf16x8*>(tile + aoff[mt] + toff + 32);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const char* wb = wl + ((tap * NT + nt) * 2) * 1024 + lane * 16;
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(wb);
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(wb + 1024);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh, acc[mt][nt], 0, 0, 0);
                    }
                }
            };
            if (a.ablate & 4) continue;
            if (taps == 9) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) do_tap(tap, ((tap / 3 - 1) * G::WP + (tap % 3 - 1)) * PIXB);
            } else {
                do_tap(0, 0);
            }
        }
    }

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
                    if (a.relu) v = (v < 0.f) ? 0.f : v;
                    const long o = (long)m * N + co;
                    if (a.aux != nullptr) a.aux[o] = v;
                    if (a.res != nullptr) v += a.res[o];
                    a.out[o] = v;
                }
            }
        }
}

template <int HW, int NT>
int launch_t(const ConvArgs& a, hipStream_t st) {
    using G = Geo<HW>;
    constexpr size_t lds = (size_t)G::NR * G::WP * PIXB + (size_t)9 * NT * 2048;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16x3_kernel<HW, NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            tdm_set_error("conv_bf16x3: hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
            return 100 + (int)e;
        }
        attr_set = true;
    }
    const long Mtot = (long)a.B * G::H * G::W;
    const int ntiles = (int)((Mtot + TILE_PX - 1) / TILE_PX);
    hipLaunchKernelGGL((conv_bf16x3_kernel<HW, NT>), dim3(ntiles), dim3(256), lds, st, a);
    TDM_CHECK_LAUNCH("conv_bf16x3");
    return 0;
}

// ---------------------------------------------------------------------------
// bf16x3 weight gradient:  dW[tap][ci][co] = sum_p A[p + tap][ci] * G[p][co]
// GEMM with M = 32 ci, N = 32 co, K = pixels on v_mfma_f32_32x32x16_bf16.
// Both operands need 8 consecutive K (= pixel) values of one channel per lane,
// while the LDS tiles are [pixel][channel] (the natural NHWC order, coalesced from
// HBM): ds_read_b64_tr_b16 does that transpose in the LDS read — a 16-lane group
// reads a 4-pixel x 16-channel block and each lane receives one channel's 4
// pixels.  Two such reads form one MFMA operand fragment.  Each lane supplies its
// own row address, so the padded-tall halo tile (any row wrap, any tap shift)
// needs no second copy.  hi and lo planes are separate 64-B-pitch images, which
// makes every transposed read a contiguous 256-B, conflict-free access.
// 8 waves per workgroup: each owns 32 of the tile's 256 pixels (2 K steps) and all
// 9 taps; partial sums are reduced through LDS once at the end and written to
// this workgroup's slab (deterministic, no atomics).  Bias gradients are summed in
// exact fp32 while staging.
// ---------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_pair(const char* p0, const char* p1) {
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
    s16x8 r;
    r[0] = lo4[0]; r[1] = lo4[1]; r[2] = lo4[2]; r[3] = lo4[3];
    r[4] = hi4[0]; r[5] = hi4[1]; r[6] = hi4[2]; r[7] = hi4[3];
    return __builtin_bit_cast(bf16x8, r);
}

template <int HW>
__global__ __launch_bounds__(512) void wgrad_bf16x3_kernel(WgradArgs a) {
    using G = Geo<HW>;
    constexpr int NPX = G::NR * G::WP;          // staged (haloed) pixels
    constexpr int APL = NPX * 64;               // bytes of one A plane (32 ch x bf16 per pixel)
    constexpr int GPL = TILE_PX * 64;           // bytes of one G plane
    extern __shared__ float4 smem4[];
    char* Ahi = reinterpret_cast<char*>(smem4);
    char* Alo = Ahi + APL;
    char* Ghi = Alo + APL;
    char* Glo = Ghi + GPL;
    int* pixoff = reinterpret_cast<int*>(Glo + GPL);   // TILE_PX ints: staged-pixel index of each tile pixel

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci_tile = blockIdx.y % a.nci, co_tile = blockIdx.y / a.nci;
    const int ci0 = ci_tile * 32, co0 = co_tile * 32;
    const int Mtot = a.B * G::H * G::W;
    const int taps = a.a.taps;
    const ConvSrc& s = a.a;
    const int up = s.up;
    const int Hs = G::H >> up, Ws = G::W >> up;
    // transposed-read lane roles: 16-lane group g -> channel block (g&1), K half (g>>1);
    // inside the group lane 4q+pc addresses row q, channels 4pc..4pc+3 of the block
    const int g4 = lane >> 4, cb = g4 & 1, hh = g4 >> 1, q = (lane >> 2) & 3, pcq = lane & 3;
    const int colb = (cb * 16 + pcq * 4) * 2;   // byte offset of the lane's 4 channels inside a 64-B pixel row

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);   // this thread's 4 output channels (tid & 7), exact fp32

    for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        const int m0 = t * TILE_PX;
        const int mlast = min(m0 + TILE_PX - 1, Mtot - 1);
        const int PR0 = padded_row<HW>(m0) - 1;
        const int nrows = padded_row<HW>(mlast) - PR0 + 2;
        const int b0 = PR0 / G::HP;
        __syncthreads();   // previous tile fully consumed
        if (tid < TILE_PX) {
            const int m = min(m0 + tid, Mtot - 1);
            const int b = m / (G::H * G::W);
            const int rem = m - b * (G::H * G::W);
            const int y = rem / G::W, x = rem - y * G::W;
            pixoff[tid] = (b * G::HP + y + 1 - PR0) * G::WP + x + 1;
        }
        // ---- stage A: haloed input tile, 32 channels, fp32 -> bf16 hi / lo planes ----
        {
            const int c4 = tid & 7;
            float4 tb0 = make_float4(0.f, 0.f, 0.f, 0.f), tb1 = tb0, tb2 = tb0;
            if (s.tb != nullptr) {
                const float* tbp = s.tb + ci0 + c4 * 4;
                tb0 = *reinterpret_cast<const float4*>(tbp + (long)min(b0, a.B - 1) * s.tb_stride);
                tb1 = *reinterpret_cast<const float4*>(tbp + (long)min(b0 + 1, a.B - 1) * s.tb_stride);
                tb2 = *reinterpret_cast<const float4*>(tbp + (long)min(b0 + 2, a.B - 1) * s.tb_stride);
            }
            const int nelem = nrows * G::WP * 8;
#pragma unroll
            for (int i = 0; i < (NPX * 8 + 511) / 512; ++i) {
                const int e = tid + 512 * i;
                if (e < nelem) {
                    const int pos = e >> 3;
                    const int lr = pos / G::WP;
                    const int pc = pos - lr * G::WP;
                    const int PR = PR0 + lr;
                    const int b = PR / G::HP;
                    const int py = PR - b * G::HP;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (py >= 1 && py <= G::H && pc >= 1 && pc <= G::W && b < a.B) {
                        const int y = (py - 1) >> up, x = (pc - 1) >> up;
                        v = *reinterpret_cast<const float4*>(s.ptr + ((long)(b * Hs + y) * Ws + x) * s.C + s.c0 + ci0 +
                                                             c4 * 4);
                        const int sel = b - b0;
                        const float4 t4 = (sel == 0) ? tb0 : ((sel == 1) ? tb1 : tb2);
                        v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w;
                    }
                    bf16x4 hi, lo;
                    split4(v, hi, lo);
                    *reinterpret_cast<bf16x4*>(Ahi + pos * 64 + c4 * 8) = hi;
                    *reinterpret_cast<bf16x4*>(Alo + pos * 64 + c4 * 8) = lo;
                }
            }
            // ---- stage G: the tile's output gradient, 32 channels ----
#pragma unroll
            for (int i = 0; i < TILE_PX * 8 / 512; ++i) {
                const int e = tid + 512 * i;
                const int px = e >> 3;
                const int m = m0 + px;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < Mtot) v = *reinterpret_cast<const float4*>(a.g + (long)m * a.Cout + co0 + c4 * 4);
                bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;
                bf16x4 hi, lo;
                split4(v, hi, lo);
                *reinterpret_cast<bf16x4*>(Ghi + px * 64 + c4 * 8) = hi;
                *reinterpret_cast<bf16x4*>(Glo + px * 64 + c4 * 8) = lo;
            }
        }
        __syncthreads();
        // ---- this wave's 32 pixels: 2 K steps of 16 ----
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int p0 = wave * 32 + ks * 16 + hh * 8 + q;      // tile pixel of the lane's row, read 0
            const int gb0 = p0 * 64 + colb, gb1 = (p0 + 4) * 64 + colb;
            const bf16x8 gh = tr_pair(Ghi + gb0, Ghi + gb1);
            const bf16x8 gl = tr_pair(Glo + gb0, Glo + gb1);
            const int ab0 = pixoff[p0] * 64 + colb, ab1 = pixoff[p0 + 4] * 64 + colb;
            if (taps == 9) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int to = ((tap / 3 - 1) * G::WP + (tap % 3 - 1)) * 64;
                    const bf16x8 ah = tr_pair(Ahi + ab0 + to, Ahi + ab1 + to);
                    const bf16x8 al = tr_pair(Alo + ab0 + to, Alo + ab1 + to);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, gh, acc[tap], 0, 0, 0);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gl, acc[tap], 0, 0, 0);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gh, acc[tap], 0, 0, 0);
                }
            } else {
                const bf16x8 ah = tr_pair(Ahi + ab0, Ahi + ab1);
                const bf16x8 al = tr_pair(Alo + ab0, Alo + ab1);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, gh, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gl, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gh, acc[0], 0, 0, 0);
            }
        }
    }

    // ---- reduce the 8 waves' partial sums through LDS, one tap at a time ----
    float* red = reinterpret_cast<float*>(smem4);   // 8 waves x 1024 floats = 32 KB (inside the A planes)
    float* slab = a.slab + (long)blockIdx.x * a.slab_stride;
    const int hl = lane >> 5, jl = lane & 31;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        if (tap < taps) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[tap][r];
            __syncthreads();
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int idx = tid + 512 * k2;
                float sum = 0.f;
#pragma unroll
                for (int w8 = 0; w8 < 8; ++w8) sum += red[w8 * 1024 + idx];
                const int r = idx >> 6, ln = idx & 63;
                const int ci = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
                const int co = ln & 31;
                slab[a.w_off + (long)(tap * a.a.w_rows + a.a.w_r0 + ci0 + ci) * a.Cout + co0 + co] = sum;
            }
        }
    }
    (void)hl; (void)jl;
    if (a.b_off >= 0 && ci_tile == 0) {
        // threads with equal (tid & 7) hold partial sums of the same 4 channels
        __syncthreads();
        float4* r4 = reinterpret_cast<float4*>(red);
        r4[tid] = bsum;
        __syncthreads();
        if (tid < 8) {
            float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = tid; k < 512; k += 8) {
                const float4 v = r4[k];
                sacc.x += v.x; sacc.y += v.y; sacc.z += v.z; sacc.w += v.w;
            }
            *reinterpret_cast<float4*>(slab + a.b_off + co0 + tid * 4) = sacc;
        }
    }
}

template <int HW>
int launch_wgrad_bf16_t(const WgradArgs& a, int nslab, hipStream_t st) {
    using G = Geo<HW>;
    constexpr size_t lds = (size_t)2 * G::NR * G::WP * 64 + (size_t)2 * TILE_PX * 64 + TILE_PX * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16x3_kernel<HW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            tdm_set_error("wgrad_bf16x3: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return 100 + (int)e;
        }
        attr_set = true;
    }
    const int nco = a.Cout / 32;
    hipLaunchKernelGGL((wgrad_bf16x3_kernel<HW>), dim3(nslab, a.nci * nco), dim3(512), lds, st, a);
    TDM_CHECK_LAUNCH("wgrad_bf16x3");
    return 0;
}

// ---------------------------------------------------------------------------
// weight pre-pack: fp32 HWIO -> bf16 hi/lo in MFMA B-fragment order
//   forward : B[k = ci][n = co] = W[tap][ci][co]            chunks over ci
//   dgrad   : B[k = co][n = ci] = W[8-tap][ci][co] (3x3)    chunks over co
// element (chunk, tap, nt, part, lane, j):  n = nt*32 + (lane&31),  k = chunk*16 + 8*(lane>>5) + j
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ P, PackArgs pa,
                                                           unsigned short* __restrict__ out) {
    const PackDesc d = pa.d[blockIdx.y];
    const int K = d.dgrad ? d.cout : d.cin;       // contraction length
    const int Nn = d.dgrad ? d.cin : d.cout;      // output channels of this direction
    const int NT = Nn / 32;
    const int total = (K / 16) * d.taps * NT * 512;   // (hi, lo) pairs
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
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
        else x = P[d.src_off + (long)((d.taps == 9 ? 8 - tap : 0) * d.cin + n) * d.cout + k];
        const __bf16 hi = (__bf16)x;
        const __bf16 lo = (__bf16)(x - (float)hi);
        const long base = d.dst_off + ((long)((chunk * d.taps + tap) * NT + nt) * 2) * 512 + lane * 8 + jj;
        out[base] = __builtin_bit_cast(unsigned short, hi);
        out[base + 512] = __builtin_bit_cast(unsigned short, lo);
    }
}

}  // namespace

int tdm_launch_conv_bf16(const ConvArgs& a, int hw, int N, hipStream_t st) {
    for (int i = 0; i < a.nsrc; ++i) {
        TDM_REQUIRE(a.src[i].nch % CK == 0 && a.src[i].nch > 0, "conv_bf16: source %d channel count %d", i, a.src[i].nch);
        TDM_REQUIRE(a.src[i].taps == 9 || a.src[i].taps == 1, "conv_bf16: taps must be 9 or 1");
        TDM_REQUIRE((a.src[i].C % 4) == 0 && (a.src[i].c0 % 4) == 0, "conv_bf16: channel alignment");
        TDM_REQUIRE(a.src[i].wp != nullptr && (((uintptr_t)a.src[i].wp) & 15) == 0, "conv_bf16: packed weights missing");
    }
    TDM_REQUIRE(a.B > 0 && (long)a.B * hw * hw * 96 < 2147483647L, "conv_bf16: batch %d out of range", a.B);
    if (hw == 28 && N == 32) return launch_t<28, 1>(a, st);
    if (hw == 28 && N == 64) return launch_t<28, 2>(a, st);
    if (hw == 28 && N == 96) return launch_t<28, 3>(a, st);
    if (hw == 14 && N == 32) return launch_t<14, 1>(a, st);
    if (hw == 14 && N == 64) return launch_t<14, 2>(a, st);
    tdm_set_error("conv_bf16: unsupported geometry hw=%d N=%d", hw, N);
    return 1;
}

int tdm_launch_wgrad_bf16(const WgradArgs& a, int hw, int nslab, hipStream_t st) {
    TDM_REQUIRE(a.Cout % 32 == 0 && a.nci >= 1, "wgrad_bf16: Cout %d / nci %d", a.Cout, a.nci);
    TDM_REQUIRE(a.a.taps == 9 || a.a.taps == 1, "wgrad_bf16: taps must be 9 or 1");
    TDM_REQUIRE(nslab >= 1 && nslab <= TDM_UNET_MAX_SLABS, "wgrad_bf16: nslab %d", nslab);
    TDM_REQUIRE(a.B > 0 && (long)a.B * hw * hw * 96 < 2147483647L, "wgrad_bf16: batch %d out of range", a.B);
    if (hw == 28) return launch_wgrad_bf16_t<28>(a, nslab, st);
    if (hw == 14) return launch_wgrad_bf16_t<14>(a, nslab, st);
    tdm_set_error("wgrad_bf16: unsupported hw=%d", hw);
    return 1;
}

int tdm_launch_pack(const float* params, const PackArgs& pa, unsigned short* out, hipStream_t st) {
    TDM_REQUIRE(pa.n >= 1 && pa.n <= TDM_MAX_PACK, "pack: %d descriptors", pa.n);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(36, pa.n), dim3(256), 0, st, params, pa, out);
    TDM_CHECK_LAUNCH("pack_weights");
    return 0;
}
