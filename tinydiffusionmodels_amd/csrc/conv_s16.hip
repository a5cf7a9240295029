// bf16x3 implicit-GEMM convolution, transposed convolution and weight gradient
// over PRE-SPLIT ("S16", tdm_s16.h) activations — the loader-light successor of
// round 1's split-while-staging kernels (removed).  Padded-tall tiling, packed weights
// (conv_pack.hip: pack_weights_kernel), one arithmetic (hi*hi + hi*lo + lo*hi, fp32
// accumulate on v_mfma_f32_32x32x16_bf16); what changes:
//   * staging a K chunk is a 16-byte copy per piece (the S16 group of a pixel is
//     already [hi 32 B | lo 32 B], exactly the LDS pixel image);
//   * the MFMA operand roles are swapped (weights = A rows, pixels = B columns),
//     so each lane ends up with 4 consecutive output channels per register
//     quad of ITS pixel: float4 epilogue loads/stores and 8-byte S16 stores;
//   * one MFMA site per tap: a 1x1 source runs through the centre-tap site, so
//     the accumulators never need moving between alternative code paths;
//   * bias gradients are no longer the weight-gradient kernel's job (the
//     producers of the gradient tensors sum them in fp32, elementwise.hip).
#include "tdm_common.h"
#include "tdm_s16.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // a 16-byte piece held in registers

namespace {

constexpr int CK = 16;
constexpr int PIXB = 80;      // bytes per staged pixel: hi 32 | lo 32 | pad 16
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

// float offset of staged position `pos` (padded-tall) in source s, or -1 (zero)
template <int HW>
__device__ __forceinline__ int src_offset(const ConvSrc& s, int PR0, int pos, int B) {
    using G = Geo<HW>;
    const int lr = pos / G::WP;
    const int pc = pos - lr * G::WP;
    const int PR = PR0 + lr;
    const int b = PR / G::HP;
    const int py = PR - b * G::HP;
    if (py >= 1 && py <= G::H && pc >= 1 && pc <= G::W && b < B) {
        const int up = s.up;
        const int y = (py - 1) >> up, x = (pc - 1) >> up;
        return ((b * (G::H >> up) + y) * (G::W >> up) + x) * s.C + s.c0;
    }
    return -1;
}

// ---------------------------------------------------------------------------
// convolution / transposed convolution
// ---------------------------------------------------------------------------
#define TDM_PIN(x) asm volatile("" : "+s"(x))
// A pointer that went through TDM_PIN is no longer known to be a kernel argument, i.e. to point to global memory, and
// would be dereferenced with flat_* instructions (both memory counters, slower issue): every access casts it back.
#define TDM_GLOBAL __attribute__((address_space(1)))
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ T gload(const void* p) { return *(const TDM_GLOBAL T*)p; }
template <typename T> __device__ __forceinline__ void gstore(void* p, const T v) { *(TDM_GLOBAL T*)p = v; }
__device__ __forceinline__ float4 gload4(const float* p) { const f32x4 v = gload<f32x4>(p); return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void gstore4(float* p, const float4 v) { f32x4 t; t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w; gstore<f32x4>(p, t); }
__device__ __forceinline__ uint4 gload16(const void* p) { const u32x4 v = gload<u32x4>(p); return make_uint4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void gstore_s16_4(float* s16, long m, int C, int c, const float4 v) {   // tdm_store_s16_4, global
    tdm_bf16x4 hi, lo;
    tdm_split4(v, hi, lo);
    char* base = reinterpret_cast<char*>(s16 + m * C + (c & ~15)) + (c & 15) * 2;
    gstore<tdm_bf16x4>(base, hi);
    gstore<tdm_bf16x4>(base + 32, lo);
}
struct PinnedSrc {
    const float* ptr; const unsigned short* wp; int C, c0, nch, up, taps, wchunk0;
    __device__ __forceinline__ void load(const ConvSrc& s) {
        ptr = s.ptr; wp = s.wp; C = s.C; c0 = s.c0; nch = s.nch; up = s.up; taps = s.taps; wchunk0 = s.wchunk0;
        TDM_PIN(ptr); TDM_PIN(wp); TDM_PIN(C); TDM_PIN(c0); TDM_PIN(nch); TDM_PIN(up); TDM_PIN(taps); TDM_PIN(wchunk0);
    }
};
struct PinnedArgs {
    PinnedSrc s0, s1;
    int nsrc, relu, B, ablate, tb_out_stride;
    const float *bias, *res, *tb_out, *skip_bias;
    float *out, *aux, *out_s16, *sums, *skip_out, *out_s16_pre;
    unsigned char* mask_out;
    const unsigned char* relu_mask_in;
    const unsigned short* skip_wp;
    __device__ __forceinline__ explicit PinnedArgs(const ConvArgs& k) {
        s0.load(k.src[0]); s1.load(k.src[1]);
        nsrc = k.nsrc; relu = k.relu; B = k.B; ablate = TDM_ABLATE(k.ablate); tb_out_stride = k.tb_out_stride;
        bias = k.bias; res = k.res; tb_out = k.tb_out; skip_bias = k.skip_bias;
        out = k.out; aux = k.aux; out_s16 = k.out_s16; sums = k.sums; skip_out = k.skip_out; out_s16_pre = k.out_s16_pre;
        mask_out = k.mask_out; relu_mask_in = k.relu_mask_in; skip_wp = k.skip_wp;
        TDM_PIN(nsrc); TDM_PIN(relu); TDM_PIN(B); TDM_PIN(tb_out_stride);
#if TDM_DIAG_BUILD
        TDM_PIN(ablate);
#endif
        TDM_PIN(bias); TDM_PIN(res); TDM_PIN(tb_out); TDM_PIN(skip_bias);
        TDM_PIN(out); TDM_PIN(aux); TDM_PIN(out_s16); TDM_PIN(sums); TDM_PIN(skip_out); TDM_PIN(out_s16_pre);
        TDM_PIN(mask_out); TDM_PIN(relu_mask_in); TDM_PIN(skip_wp);
    }
    // field of source si (uniform): scalar selects, no indexed kernarg read
    __device__ __forceinline__ PinnedSrc src(int si) const {
        PinnedSrc r;
        r.ptr = si ? s1.ptr : s0.ptr; r.wp = si ? s1.wp : s0.wp; r.C = si ? s1.C : s0.C; r.c0 = si ? s1.c0 : s0.c0;
        r.nch = si ? s1.nch : s0.nch; r.up = si ? s1.up : s0.up; r.taps = si ? s1.taps : s0.taps;
        r.wchunk0 = si ? s1.wchunk0 : s0.wchunk0;
        return r;
    }
};

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// staged rows of a 256 MT pixel tile: pixel rows it can touch (+1 for an unaligned start) + one image seam (2 halo rows)
// + the top and bottom halo rows
template <int HW, int MT> struct ConvRows { static constexpr int NR = (HW - 1 + 256 * MT + HW - 1) / HW + 4; };
// registers of the fused 1x1 skip conv (ConvArgs::skip_out): exist only in the SKIP instantiation
template <int NT, bool SKIP> struct SkipState {};
template <int NT> struct SkipState<NT, true> { f32x16 acc2[NT]; uint4 psk; };

// 8 waves per workgroup, ONE 32-pixel M tile per wave.  tools/phase_probe.py showed every phase of the former 4-wave /
// 2-M-tile workgroup taking the same time with the CU to itself as with a second workgroup next to it: at two waves per
// SIMD the kernel is bound by the latency of its own dependent instruction chains (index math, LDS round trips,
// epilogue), not by any throughput.  Halving the per-wave work and the register budget (<= 128 VGPRs at N <= 64) puts
// four waves on every SIMD with the same 256-pixel tile and the same LDS images.
constexpr int CONV_THREADS = 512;
constexpr int NPIN = 4;   // 16-byte input pieces per thread and K chunk

__device__ __forceinline__ void gstore_s16_o(float* s16, unsigned o, int c, const float4 v) {   // o = pixel * C + c (elements)
    tdm_bf16x4 hi, lo;
    tdm_split4(v, hi, lo);
    char* base = reinterpret_cast<char*>(s16 + (o - (unsigned)(c & 15))) + (c & 15) * 2;
    gstore<tdm_bf16x4>(base, hi);
    gstore<tdm_bf16x4>(base + 32, lo);
}

// MT = M tiles (32 pixels each) per wave: the workgroup's tile is 256 MT pixels.  MT = 2 (28x28, N = 32 only): measured
// per-round time of this kernel is a + b x (MFMAs per wave) with a fixed a = 4.5 us (first-load latency, prologue,
// epilogue) and b set by the LDS fragment traffic — a wave reading ONE weight fragment pair for TWO pixel fragments halves
// the rounds (and the fixed cost paid per round) and cuts the fragment bytes per MFMA by a quarter, at the same 4 waves
// per SIMD (32 accumulator registers instead of 16; single prefetch set).
// MSE: the instantiation of rb4.conv2's forward launch in a train step (ConvArgs::o1_tgt): fused output conv + MSE backward;
// it carries neither the rank-1 residual nor the ReLU-backward sums (their registers would push the walk over the 128 budget).
// PH ("phase" form, rb4.conv1's forward: ConvArgs::up_phase): source 0 is a HALF-resolution tensor that the reference up-samples
// (nearest, x2: src/mnist.py:83) before a 3x3 convolution.  For an output pixel (2i + py, 2j + px) the three taps of a direction
// fall on only TWO source pixels — py = 0: rows {i-1, i} with weights {W[-1], W[0] + W[+1]}; py = 1: rows {i, i+1} with
// {W[-1] + W[0], W[+1]} — so the nine taps over the 4x up-sampled image are FOUR taps over the source image with weights
// pre-summed per phase (py, px) (conv_pack.hip, PackDesc::phase): 4/9 of the MFMAs and of the pixel-fragment reads for those
// channels, and the staged image is the 14x14 source itself (no up-sampled duplicate in LDS).  All 32 pixels of an MFMA must
// share the weights, hence the tile: 64 consecutive SOURCE positions x 4 phases; wave w owns phase w & 3 of source positions
// 32 (w >> 2) .. + 31, i.e. 32 same-parity output pixels.  Source 1 (h1, full resolution) runs through the ordinary nine taps
// with the same lanes, reading a 28x28-domain image staged around the tile's output rows.
// S2D ("space to depth", rb4.conv1's data gradient w.r.t. the up-sampled h3: ConvArgs::s2d): the transpose of the phase form.  The
// gradient of a source pixel (i, j) gathers the output gradient g over the 4x4 window g[2i-1 .. 2i+2][2j-1 .. 2j+2] with 16
// pre-summed weight matrices (PackDesc::phase = 2) — 512 instead of 1152 K elements per source pixel, and the result IS the
// 14x14 gradient (no pair-summed intermediate, no second pass to add rows).  g is read as its four parity sub-images
// g_pq[i'][j'] = g[2i'+p][2j'+q]: K chunk c stages sub-image c / nc0 (16 channels) in the ordinary 14x14 geometry and uses the
// FOUR taps of the 3x3 neighbourhood that sub-image contributes (rows {i, i+1} for p = 0, {i-1, i} for p = 1; columns likewise).
template <int HW, int NT, bool SKIP, bool PROBE, int MT = 1, bool MSE = false, bool PH = false, bool S2D = false>
__global__ __launch_bounds__(CONV_THREADS, (NT <= 2 ? 4 : 2)) void conv_s16_kernel(ConvArgs ka) {
    using G = Geo<HW>;
    using G2 = Geo<14>;                                                // geometry of the half-resolution source (PH)
    constexpr int TPX = PH ? 64 : TILE_PX * MT;                        // pixels of the workgroup's tile (PH: SOURCE positions)
    constexpr int NR28 = 16, NR14 = 10;                                // PH: staged rows of the two images (64 positions: <= 6 source rows, + seam, + halo)
    constexpr int NRv = PH ? NR28 : (MT == 1 ? G::NR : ConvRows<HW, MT>::NR);   // staged rows of the padded-tall image
    constexpr int NPINv = (NRv * G::WP * 4 + CONV_THREADS - 1) / CONV_THREADS;   // 16-byte input pieces per thread and K chunk
    constexpr int PHT = 16;                                            // PH: packed taps per chunk of source 0 (4 phases x 4)
    constexpr int WTAPS = PH ? PHT : 9;                                // weight fragments (per N tile) the LDS weight area holds
    static_assert(NRv <= 32, "the per-wave row table has 32 entries");
    static_assert(MT == 1 || (HW == 28 && NT == 1), "two M tiles per wave: built for the 28x28 N = 32 kernels");
    static_assert(!PH || (HW == 28 && NT == 1 && SKIP && MT == 1 && !MSE), "the phase form is built for rb4.conv1's forward");
    static_assert(!S2D || (HW == 14 && NT == 2 && !SKIP && MT == 1 && !MSE && !PH), "the space-to-depth form is built for rb4.conv1's data gradient");
    // The ~300-byte argument block does not stay in scalar registers by itself: the compiler re-reads a field from the
    // kernarg segment (s_load + s_waitcnt lgkmcnt(0), a scalar-cache round trip) next to almost every use — before each
    // prefetch load, around every uniform branch of the epilogue.  Everything the kernel uses is copied ONCE into
    // scalars the optimiser cannot rematerialise (opaque asm), about 60 SGPRs.
    PinnedArgs a(ka);
    // rank-1 residual (ConvArgs::r1_*): only the 28x28 N = 32 instantiation (rb1.conv2) carries its registers
    constexpr bool R1 = HW == 28 && NT == 1 && !SKIP;
    static_assert(!MSE || R1, "the fused MSE backward rides on the 28x28 N = 32 kernel's fused output conv");
    constexpr bool R1X = R1 && !MSE;
    const float* r1_x = R1X ? ka.r1_x : nullptr; const float* r1_w = ka.r1_w; const float* r1_b = ka.r1_b;
    if constexpr (R1X) { TDM_PIN(r1_x); TDM_PIN(r1_w); TDM_PIN(r1_b); }
    float* o1_out = R1 ? ka.o1_out : nullptr; const float* o1_w = ka.o1_w; const float* o1_b = ka.o1_b;   // fused 1x1 output conv (rb4.conv2)
    if constexpr (R1) { TDM_PIN(o1_out); TDM_PIN(o1_w); TDM_PIN(o1_b); }
    const float* o1_tgt = MSE ? ka.o1_tgt : nullptr; float* o1_deps = ka.o1_deps; float* o1_sums = ka.o1_sums;   // MSE at the source (training)
    int o1_dscale_bits = __float_as_int(ka.o1_dscale);
    if constexpr (MSE) { TDM_PIN(o1_tgt); TDM_PIN(o1_deps); TDM_PIN(o1_sums); TDM_PIN(o1_dscale_bits); }
    // rb4.conv1's data gradient (ConvArgs::dc_pair): only the 28x28 N = 96 instantiation carries it
    constexpr bool DCAT = HW == 28 && NT == 3 && !SKIP;
    float* dc_pair = DCAT ? ka.dc_pair : nullptr; float* dc_h1 = ka.dc_h1; const float* rk1_d = ka.rk1_d; const float* rk1_u = ka.rk1_u;
    if constexpr (DCAT) { TDM_PIN(dc_pair); TDM_PIN(dc_h1); TDM_PIN(rk1_d); TDM_PIN(rk1_u); }
    const float* s2_d = S2D ? ka.rk1_d : nullptr; const float* s2_u = ka.rk1_u;   // S2D: the skip path's rank-one share (4 pixels' d summed)
    if constexpr (S2D) { TDM_PIN(s2_d); TDM_PIN(s2_u); }
    constexpr int N = NT * 32;
    constexpr int TILE_B = NRv * G::WP * PIXB;
    extern __shared__ float4 smem4[];
    char* tile = reinterpret_cast<char*>(smem4);
    char* wl = tile + TILE_B;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: everything derived from it stays scalar
    const int h = lane >> 5, j = lane & 31;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    // phase probe (PROBE instantiation, launched when ablate & 16; tools/phase_probe.py): thread 0 stamps the shader clock
    // at phase boundaries into the int64 table [blockIdx.x][16] that the caller appended BEHIND the B*H*W*N floats of aux
    int nstamp = 0;
    auto stamp = [&]() {
        if constexpr (PROBE) {
            if ((a.ablate & 16) && tid == 0 && nstamp < 16)
                gstore<long long>(reinterpret_cast<long long*>(a.aux + (long)a.B * G::H * G::W * N) + (long)blockIdx.x * 16 + nstamp,
                                  (long long)__builtin_readcyclecounter());
            ++nstamp;
        }
    };
    stamp();
    const int Mtot = a.B * G::H * G::W;
    // PH: the tile is 64 SOURCE positions [m0, mlast] of the (B, 14, 14) raster; the 28x28-domain image is staged from the halo
    // row above output row 2 * i0 (ty0 = 2 * i0), the source image from the halo row above source row i0 (ty0s = i0)
    const int Mtile = PH ? a.B * G2::H * G2::W : Mtot;        // positions the tiles walk
    const int m0 = t * TPX;
    const int mlast = min(m0 + TPX - 1, Mtile - 1);
    const int tb0 = m0 / (PH ? G2::H * G2::W : G::H * G::W);  // image and row of the tile's first pixel:
    const int ty0s = PH ? (m0 - tb0 * (G2::H * G2::W)) / G2::W : 0;   // (PH: first source row)
    const int ty0 = PH ? 2 * ty0s : (m0 - tb0 * (G::H * G::W)) / G::W;   // the staged image starts at padded row PR0 = tb0 * HP + ty0
    const int PR0 = tb0 * G::HP + ty0;
    int nrows, nrows_s = 0;
    if constexpr (PH) {
        const int b1 = mlast / (G2::H * G2::W);
        const int i1 = (mlast - b1 * (G2::H * G2::W)) / G2::W;
        nrows = (b1 * G::HP + 2 * i1 + 2) - PR0 + 2;          // last output row 2 i1 + 1 sits at padded row b1 HP + 2 i1 + 2
        nrows_s = (b1 * G2::HP + i1 + 1) - (tb0 * G2::HP + ty0s) + 2;
    } else nrows = padded_row<HW>(mlast) - PR0 + 2;
    const int nelem = nrows * G::WP * 4;      // 16-byte pieces of one chunk
    const int nelem_s = nrows_s * G2::WP * 4; // (PH: of a source-0 chunk)
    const int mbase0 = PH ? m0 + (wave >> 2) * 32 : m0 + wave * (32 * MT);   // this wave's first M tile (scalar); its MT tiles are consecutive
    const int ph_y = PH ? (wave >> 1) & 1 : 0, ph_x = PH ? wave & 1 : 0;      // PH: this wave's phase

    int aoff[MT];   // LDS byte offset of this lane's pixel (centre tap) in the staged image, per M tile
    int aoff_s = 0; // PH: ... of its SOURCE pixel in the staged source image
    int m_lane = 0; // PH: output pixel (index into the (B, 28, 28) raster) of this lane, -1 past the end
    if constexpr (PH) {
        const int sp = mbase0 + j;                            // source position of this lane (pixel j of the wave's M tile)
        const int sc = min(sp, Mtile - 1);
        const int b = sc / (G2::H * G2::W);
        const int rem = sc - b * (G2::H * G2::W);
        const int i = rem / G2::W, jx = rem - i * G2::W;
        const int yo = 2 * i + ph_y, xo = 2 * jx + ph_x;
        aoff[0] = (((b - tb0) * G::HP + yo + 1 - ty0) * G::WP + xo + 1) * PIXB + h * 16;
        aoff_s = (((b - tb0) * G2::HP + i + 1 - ty0s) * G2::WP + jx + 1) * PIXB + h * 16;
        m_lane = sp < Mtile ? (b * G::H + yo) * G::W + xo : -1;
    } else
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        // (image, row, column) of the M tile's first pixel are scalar; a lane is q = x0 + j columns further: q / W by
        // multiply-shift (exact for q < W + 128), no per-lane constant divisions
        const int mbase = mbase0 + mt * 32;
        const int mb = min(mbase, Mtot - 1);
        const int b0 = mb / (G::H * G::W);
        const int rem0 = mb - b0 * (G::H * G::W);
        const int y0 = rem0 / G::W, x0 = rem0 - y0 * G::W;
        const int q = x0 + min(j, Mtot - 1 - mb);       // (pixels past the end: the last real one)
        const int dr = (q * (HW == 28 ? 2341 : 4682)) >> 16;
        const int x = q - dr * G::W;
        int y = y0 + dr, rowb = (b0 - tb0) * G::HP;
        if (y >= G::H) { y -= G::H; rowb += G::HP; }
        aoff[mt] = ((rowb + y + 1 - ty0) * G::WP + x + 1) * PIXB + h * 16;
    }

    // accumulators start from the conv bias (register quad g of N tile nt = channels nt*32 + 8g + 4h .. +3 of the lane's
    // pixel): the loads are issued first thing and the epilogue has nothing left to add
    f32x16 acc[MT][NT];
    auto init_acc = [&](f32x16* ac, const float* bias) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 bz = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bias != nullptr) bz = gload4(bias + nt * 32 + 8 * g + 4 * h);
                ac[nt][4 * g] = bz.x; ac[nt][4 * g + 1] = bz.y; ac[nt][4 * g + 2] = bz.z; ac[nt][4 * g + 3] = bz.w;
            }
    };
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) init_acc(acc[mt], a.bias);
    SkipState<MT * NT, SKIP> sk;
    if constexpr (SKIP) {
        sk.psk = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) init_acc(sk.acc2 + mt * NT, a.skip_bias);
    }

    // Software pipeline over K chunks, register-staged.  At N = 32 input pieces are prefetched TWO chunks ahead (two
    // register sets, 4 x 16 B per thread each), the L2-resident packed weights one chunk ahead: with a single chunk of
    // distance a workgroup's MFMA phase is shorter than the HBM latency under load and every chunk stalled.
    constexpr int WN = (WTAPS * NT * 128 + CONV_THREADS - 1) / CONV_THREADS;
    constexpr int STEP = CONV_THREADS / 4;   // staged positions between a thread's consecutive pieces
    char* const sdst = tile + (tid >> 2) * PIXB + (tid & 3) * 16;
    const int nc0 = S2D ? 2 : a.s0.nch >> 4;   // (S2D: 32 gradient channels, checked by the launcher)
    const int nchunks = S2D ? 4 * nc0 : nc0 + (a.nsrc > 1 ? (a.s1.nch >> 4) : 0);   // S2D: four parity sub-images of source 0
    int goffA[NPINv], goffB[NPINv];   // staging plans of the sources the two input sets were loaded from
    int planA = -1, planB = -1;
    uint4 pinA[NPINv], pinB[NPINv], pwt[WN];

    // Inputs and weights come through buffer descriptors: a piece that is padding (or past the end) gets an offset
    // beyond num_records and the hardware returns zeros — no exec-mask branch per load, no zero-initialised
    // destination, and (straight-line code) exact vmcnt(n) waits, so the second register set really stays in flight
    // while the first is staged.
    auto prefetch_in = [&](uint4 (&pin)[NPINv], int (&goff)[NPINv], int& plan_src, const int (&goff_other)[NPINv], int plan_other, int c) {
        const int si = S2D ? c / nc0 : ((c >= nc0) ? 1 : 0);   // (S2D: the plan is per parity sub-image)
        const int ch = S2D ? c - si * nc0 : (si ? c - nc0 : c);
        const PinnedSrc s = a.src(S2D ? 0 : si);
        const bool half = PH && si == 0;                      // PH: source 0 is staged at its own (14x14) resolution
        const int up = PH ? 0 : s.up, Hs = S2D ? 28 : (half ? G2::H : G::H >> up), Ws = S2D ? 28 : (half ? G2::W : G::W >> up);
        if (plan_src != si && plan_other == si) {   // the other register set already holds this source's plan
#pragma unroll
            for (int i = 0; i < NPINv; ++i) goff[i] = goff_other[i];
            plan_src = si;
        }
        if constexpr (PH) {
            if (plan_src != si) {
                // the same plan as below for either image of the phase form: geometry (HP, H, W, WP), first staged row and row
                // count are those of the source's own resolution
                const int HPv = half ? G2::HP : G::HP, Hv = half ? G2::H : G::H, Wv = half ? G2::W : G::W, WPv = half ? G2::WP : G::WP;
                const int ty0v = half ? ty0s : ty0, nrv = half ? nrows_s : nrows;
                int* const rowtab = reinterpret_cast<int*>(wl + WTAPS * NT * 2048 + NT * 2048) + wave * 32;
                if (lane < 32) {
                    int py = ty0v + lane, b = tb0;
                    if (py >= HPv) { py -= HPv; ++b; }
                    const bool ok = lane < nrv && py >= 1 && py <= Hv && b < a.B;
                    rowtab[lane] = ok ? __mul24(__mul24(__mul24(b, Hs) + (py - 1), Ws), s.C) * 4 : (int)0x80000000;
                }
                int lr = half ? (tid >> 2) / G2::WP : (tid >> 2) / G::WP;
                int pc = (tid >> 2) - lr * WPv;
                const int cbase = (s.c0 + (tid & 3) * 4) * 4;
                const int stp = half ? STEP % G2::WP : STEP % G::WP, str = half ? STEP / G2::WP : STEP / G::WP;
#pragma unroll
                for (int i = 0; i < NPINv; ++i) {
                    const int roff = rowtab[min(lr, 31)];
                    const bool ok = lr < 32 && roff >= 0 && pc >= 1 && pc <= Wv;
                    goff[i] = ok ? roff + __mul24(pc - 1, s.C) * 4 + cbase : (int)0x80000000;
                    pc += stp;
                    lr += str;
                    if (pc >= WPv) { pc -= WPv; ++lr; }
                }
                plan_src = si;
            }
        } else if constexpr (S2D) {
            if (plan_src != si) {
                // staged pixel (row py, column pc) of sub-image (p, q) is pixel (2 (py - 1) + p, 2 (pc - 1) + q) of the 28x28 tensor
                const int sp = si >> 1, sq = si & 1;
                int* const rowtab = reinterpret_cast<int*>(wl + WTAPS * NT * 2048) + wave * 32;
                if (lane < 32) {
                    int py = ty0 + lane, b = tb0;
                    if (py >= G::HP) { py -= G::HP; ++b; }
                    if (py >= G::HP) { py -= G::HP; ++b; }
                    const bool ok = lane < nrows && py >= 1 && py <= G::H && b < a.B;
                    rowtab[lane] = ok ? __mul24(__mul24(__mul24(b, 28) + 2 * (py - 1) + sp, 28), s.C) * 4 : (int)0x80000000;
                }
                int lr = (tid >> 2) / G::WP;
                int pc = (tid >> 2) - lr * G::WP;
                const int cbase = (s.c0 + (tid & 3) * 4) * 4;
#pragma unroll
                for (int i = 0; i < NPINv; ++i) {
                    const int roff = rowtab[lr];
                    const bool ok = roff >= 0 && pc >= 1 && pc <= G::W;
                    goff[i] = ok ? roff + __mul24(2 * (pc - 1) + sq, s.C) * 4 + cbase : (int)0x80000000;
                    pc += STEP % G::WP;
                    lr += STEP / G::WP;
                    if (pc >= G::WP) { pc -= G::WP; ++lr; }
                }
                plan_src = si;
            }
        } else
        if (plan_src != si) {
            // Staged position of piece e = tid + 512 i is (tid >> 2) + 128 i = (row lr, column pc) of the padded-tall image,
            // walked incrementally (no division per piece).  What depends on the ROW — image, validity, the three
            // multiplications of the source offset — is worked out once per row by lanes 0..31 of each wave and parked in
            // a wave-private LDS table; a piece then costs one LDS read, the column offset and a select.
            int* const rowtab = reinterpret_cast<int*>(wl + WTAPS * NT * 2048 + (SKIP ? NT * 2048 : 0)) + wave * 32;
            if (lane < 32) {
                int py = ty0 + lane, b = tb0;
                if (py >= G::HP) { py -= G::HP; ++b; }
                if (HW == 14 && py >= G::HP) { py -= G::HP; ++b; }   // 26 staged rows span up to three images at 14x14
                const bool ok = lane < nrows && py >= 1 && py <= G::H && b < a.B;
                rowtab[lane] = ok ? __mul24(__mul24(__mul24(b, Hs) + ((py - 1) >> up), Ws), s.C) * 4 : (int)0x80000000;
            }
            int lr = (tid >> 2) / G::WP;
            int pc = (tid >> 2) - lr * G::WP;
            const int cbase = (s.c0 + (tid & 3) * 4) * 4;
#pragma unroll
            for (int i = 0; i < NPINv; ++i) {
                const int roff = rowtab[lr];                       // (lr <= 31 for both geometries)
                const bool ok = roff >= 0 && pc >= 1 && pc <= G::W;
                goff[i] = ok ? roff + __mul24((pc - 1) >> up, s.C) * 4 + cbase : (int)0x80000000;   // byte offset, or out of range
                pc += STEP % G::WP;
                lr += STEP / G::WP;
                if (pc >= G::WP) { pc -= G::WP; ++lr; }
            }
            plan_src = si;
        }
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.ptr), 0, a.B * Hs * Ws * s.C * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < NPINv; ++i) {
            const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i], ch << 6, 0));
            pin[i] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    auto prefetch_w = [&](int c) {
        const int si = S2D ? 0 : ((c >= nc0) ? 1 : 0);
        const int ch = S2D ? c : (si ? c - nc0 : c);   // (S2D: packed chunk = (sub-image, channel chunk) = c)
        const PinnedSrc s = a.src(si);
        const int nbytes = s.taps * NT * 2048;   // one chunk of packed weights (PH, source 0: 16 phase taps; S2D: 4)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned short*>(s.wp) + (long)(s.wchunk0 + ch) * (s.taps * NT * 1024), 0, nbytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + CONV_THREADS * i) * 16, 0, 0));
            pwt[i] = make_uint4(v[0], v[1], v[2], v[3]);
        }
        if constexpr (SKIP) {   // NT * 128 pieces of the 1x1 weights of the same chunk
            const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<unsigned short*>(a.skip_wp) + (long)(s.wchunk0 + ch) * (NT * 1024), 0, NT * 2048, 0x00020000);
            const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, tid * 16, 0, 0));
            sk.psk = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    auto stage = [&](const uint4 (&pin)[NPINv], int nel) {
        __syncthreads();      // everyone finished reading the previous chunk's LDS image
#pragma unroll
        for (int i = 0; i < NPINv; ++i)
            if (tid + CONV_THREADS * i < nel) *reinterpret_cast<uint4*>(sdst + i * (STEP * PIXB)) = pin[i];
#pragma unroll
        for (int i = 0; i < WN; ++i)
            if (tid + CONV_THREADS * i < WTAPS * NT * 128) reinterpret_cast<uint4*>(wl)[tid + CONV_THREADS * i] = pwt[i];
        if constexpr (SKIP) {   // packed 1x1 weights of the chunk: NT x (hi 1 KB | lo 1 KB) behind the 3x3 weights
            if (tid < NT * 128) reinterpret_cast<uint4*>(wl + WTAPS * NT * 2048)[tid] = sk.psk;
        }
        __syncthreads();
    };
    auto compute = [&](int c) {
        if (a.ablate & 4) return;
        const int taps = (c >= nc0) ? a.s1.taps : a.s0.taps;
        // PH, chunk of the half-resolution source: sites 0..3 are the phase's four taps over the staged SOURCE image — site
        // 2 a' + b' reads source pixel (i + ph_y - 1 + a', j + ph_x - 1 + b') with packed tap (2 ph_y + ph_x) * 4 + 2 a' + b' —
        // and site 4 re-reads the source pixel itself (a' = 1 - ph_y, b' = 1 - ph_x) for the fused 1x1 skip conv alone
        const bool halfc = PH && c < nc0;
        const int pix0 = halfc ? aoff_s : aoff[0];
        const int s2p = S2D ? (c / nc0) >> 1 : 0, s2q = S2D ? (c / nc0) & 1 : 0;   // S2D: parity of the chunk's sub-image
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            // S2D: sub-image (p, q) contributes rows {0, +1} (p = 0) or {-1, 0} (p = 1) of the 3x3 neighbourhood, columns likewise
            const bool s2on = (s2p ? tp / 3 <= 1 : tp / 3 >= 1) && (s2q ? tp % 3 <= 1 : tp % 3 >= 1);
            if (S2D ? s2on : (PH ? (!halfc || tp <= 4) : (taps == 9 || tp == 4))) {   // a 1x1 source uses the centre-tap site with packed tap 0
                int toff = ((tp / 3 - 1) * G::WP + (tp % 3 - 1)) * PIXB;
                int wt = (taps == 9) ? tp : 0;
                if constexpr (S2D) wt = (tp / 3 - (s2p ? 0 : 1)) * 2 + (tp % 3 - (s2q ? 0 : 1));
                bool main_on = true;
                if constexpr (PH) {
                    if (halfc) {
                        const int ap = tp < 4 ? (tp >> 1) : 1 - ph_y, bp = tp < 4 ? (tp & 1) : 1 - ph_x;
                        toff = ((ph_y - 1 + ap) * G2::WP + (ph_x - 1 + bp)) * PIXB;
                        wt = (2 * ph_y + ph_x) * 4 + (tp & 3);
                        main_on = tp < 4;
                    }
                }
                bf16x8 ah[MT], al[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    ah[mt] = *reinterpret_cast<const bf16x8*>(tile + (PH ? pix0 : aoff[mt]) + toff);
                    al[mt] = *reinterpret_cast<const bf16x8*>(tile + (PH ? pix0 : aoff[mt]) + toff + 32);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (main_on) {
                    const char* wb = wl + ((wt * NT + nt) * 2) * 1024 + lane * 16;
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(wb);
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(wb + 1024);
                    // D[co][pixel]: weights are the A operand (one weight fragment pair feeds all MT pixel tiles)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al[mt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah[mt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah[mt], acc[mt][nt], 0, 0, 0);
                    }
                    }
                    if constexpr (SKIP) if (tp == 4) {   // the block's 1x1 skip conv reads exactly the centre-tap pixels
                        const char* sb = wl + WTAPS * NT * 2048 + (nt * 2) * 1024 + lane * 16;
                        const bf16x8 sh = *reinterpret_cast<const bf16x8*>(sb);
                        const bf16x8 sl = *reinterpret_cast<const bf16x8*>(sb + 1024);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            sk.acc2[mt * NT + nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sh, al[mt], sk.acc2[mt * NT + nt], 0, 0, 0);
                            sk.acc2[mt * NT + nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl, ah[mt], sk.acc2[mt * NT + nt], 0, 0, 0);
                            sk.acc2[mt * NT + nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sh, ah[mt], sk.acc2[mt * NT + nt], 0, 0, 0);
                        }
                    }
                }
            }
        }
    };

    const bool pf = !(a.ablate & 1);
    prefetch_in(pinA, goffA, planA, goffB, planB, 0);
    prefetch_w(0);
    if constexpr (NT == 1 && !SKIP && MT == 1) {
        if (nchunks > 1) prefetch_in(pinB, goffB, planB, goffA, planA, 1);
        stamp();
        for (int c = 0; c < nchunks; c += 2) {
            stage(pinA, nelem);                                           // chunk c
            stamp();
            if (c + 2 < nchunks && pf) prefetch_in(pinA, goffA, planA, goffB, planB, c + 2);
            if (c + 1 < nchunks && pf) prefetch_w(c + 1);
            compute(c);
            stamp();
            if (c + 1 >= nchunks) break;
            stage(pinB, nelem);                                           // chunk c + 1
            stamp();
            if (c + 3 < nchunks && pf) prefetch_in(pinB, goffB, planB, goffA, planA, c + 3);
            if (c + 2 < nchunks && pf) prefetch_w(c + 2);
            compute(c + 1);
            stamp();
        }
    } else {
        // N = 64 / 96 (and N = 32 with the fused skip conv): a second input register set does not fit the 128-register
        // budget of four waves per SIMD, so these widths prefetch one chunk ahead
        stamp();
        for (int c = 0; c < nchunks; ++c) {
            stage(pinA, (PH && c < nc0) ? nelem_s : nelem);
            stamp();
            if (c + 1 < nchunks && pf) { prefetch_in(pinA, goffA, planA, goffB, planB, c + 1); prefetch_w(c + 1); }
            compute(c);
            stamp();
        }
    }

    // Epilogue through LDS.  In the accumulator layout lane = pixel j of its M tile and register quad g of N tile
    // nt = channels nt*32 + 8g + 4h .. +3: storing from there writes 32-byte pieces of 32 different pixel rows per
    // instruction.  Each wave instead transposes 32 pixels x N channels through a private LDS block and walks it
    // row-major — 32 pixels x N floats are ONE contiguous range of every NHWC tensor involved — so each load / store
    // instruction of the epilogue (residual in; out, saved post-ReLU copy, S16 twin out) covers 1 KB of consecutive
    // addresses.
    // The walk's own inputs (residual, ReLU byte mask, time-bias row) are fetched into registers for the whole M tile
    // BEFORE its transpose — one exposed memory latency per tile, not one per 64-float4 pass (the passes used to be
    // separated by uniform branches, so the compiler waited for each pass's loads on the spot).  The conv bias is
    // already in the accumulators.  All offsets are 32-bit element offsets from scalar bases (tensors < 2^32 bytes).
    if (a.ablate & 8) return;
    nstamp = 6;
    // opaque copies: keeps the optimiser from computing the epilogue's per-lane offsets before the K loop and carrying
    // them through it (at N = 64 that cost spills inside the loop under the 128-register budget)
    int lane_e = lane, j_e = j, h_e = h;
    asm volatile("" : "+v"(lane_e), "+v"(j_e), "+v"(h_e));
    constexpr int EPI = N + 4;   // floats per staged pixel row: (N/4 + 1) x 16 B, an odd slot count
    constexpr int NIT = N / 8;   // passes per M tile: 32 * N/4 float4, 64 per instruction
    const bool bwd = !MSE && NT != 3 && a.relu_mask_in != nullptr;   // (N = 32 / 64 outputs only)
    constexpr int GI = 4;   // passes per group: 4 independent chains, 16 value registers
    struct Pre { float4 rt[GI]; unsigned mk[GI]; float rx[R1X ? GI : 1]; float tg[MSE ? GI : 1]; } p;   // inputs of ONE group of passes; rt: residual, else the time-bias row
    // branch-free: absent inputs get an empty descriptor (num_records 0 -> zeros), so the requests are one straight run
    // of loads (under uniform branches each request became a load + s_waitcnt vmcnt(0) + spill at N = 64)
    const bool use_res = a.res != nullptr;
    const bool use_tb = !use_res && a.tb_out != nullptr && a.out_s16 != nullptr;
    const __amdgpu_buffer_rsrc_t rs_rt = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(use_res ? a.res : a.tb_out), 0, use_res ? Mtot * N * 4 : (use_tb ? a.B * a.tb_out_stride * 4 : 0), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_mk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(a.relu_mask_in), 0, bwd ? Mtot * (N / 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(r1_x), 0, r1_x != nullptr ? Mtot * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_tg = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(o1_tgt), 0, o1_tgt != nullptr ? Mtot * 4 : 0, 0x00020000);
    float* T = reinterpret_cast<float*>(smem4) + wave * (32 * EPI);
    // The wave's MT M tiles go through the epilogue one after the other (same wave-private LDS block, same registers).
    static_for<0, MT>([&](auto mtc) {
    constexpr int mt = decltype(mtc)::value;
    const int mbase = mbase0 + mt * 32;
    // PH: the wave's 32 pixels are not consecutive — pixel px of the tile is output pixel m_lane of lane px; pass `it` of the
    // walk handles pixel 8 it + (lane >> 3) (N = 32: eight lanes per pixel), fetched once per pass (-1: past the end)
    int mpx[PH ? 4 : 1];
    if constexpr (PH) {
#pragma unroll
        for (int it = 0; it < 4; ++it) mpx[it] = __shfl(m_lane, it * 8 + (lane_e >> 3));
    }
    // S2D: the skip path's rank-one share of a SOURCE pixel is (sum of d over its four output pixels) * u[c]; lane j holds the
    // sum of pixel j of the tile, the walk fetches it per pass
    float d4_lane = 0.f;
    if constexpr (S2D) {
        if (s2_d != nullptr) {
            const int ms = min(mbase + j_e, Mtot - 1);
            const int b = ms / (G::H * G::W);
            const int rem = ms - b * (G::H * G::W);
            const int i = rem / G::W, jx = rem - i * G::W;
            const int m28 = (b * 28 + 2 * i) * 28 + 2 * jx;
            const f32x2 r0 = gload<f32x2>(s2_d + m28), r1 = gload<f32x2>(s2_d + m28 + 28);
            d4_lane = (r0[0] + r0[1]) + (r1[0] + r1[1]);
        }
    }
    const int img0 = PH ? min(mbase, Mtile - 1) / (G2::H * G2::W) : mbase / (G::H * G::W);   // image of the group's first pixel (scalar)
    const int mnext = (img0 + 1) * (G::H * G::W);   // a 32-pixel group touches at most two images
    const bool has_epi_in = use_res || use_tb || bwd || (R1X && r1_x != nullptr) || MSE;
    auto preload = [&](int g) {
#pragma unroll
        for (int it = 0; it < GI; ++it) {
            const int e = (g * GI + it) * 64 + lane_e;
            const int px = e / (N / 4), c = (e - px * (N / 4)) * 4;
            int m = min(mbase + px, Mtot - 1);   // clamped: the walk skips pixels past the end
            if constexpr (PH) m = max(mpx[(g * GI + it) & 3], 0);
            const int o = m * N + c;
            const int otb = (img0 + (m >= mnext ? 1 : 0)) * a.tb_out_stride + c;
            const f32x4 r = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_rt, (use_res ? o : otb) * 4, 0, 0));
            p.rt[it] = make_float4(r[0], r[1], r[2], r[3]);
            p.mk[it] = __builtin_amdgcn_raw_buffer_load_b8(rs_mk, o >> 2, 0, 0);
            if constexpr (R1X) p.rx[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_rx, m * 4, 0, 0));
            if constexpr (MSE) p.tg[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_tg, m * 4, 0, 0));
        }
    };
    if (!(DCAT && dc_pair != nullptr)) preload(0);   // (the d cat epilogue fetches its own two small inputs)
    stamp();                     // 6: tile inputs requested
    if constexpr (mt == 0) __syncthreads();   // every wave is done with the operand images
    stamp();                     // 7
    auto to_lds = [&](const f32x16* ac) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(T + j_e * EPI + nt * 32 + 8 * g + 4 * h_e) =
                    make_float4(ac[nt][4 * g], ac[nt][4 * g + 1], ac[nt][4 * g + 2], ac[nt][4 * g + 3]);
    };
    to_lds(acc[mt]);
    stamp();                     // 8: transpose written
    if constexpr (DCAT) if (dc_pair != nullptr) {
        // d cat from rb4.conv1's data gradient (ConvArgs::dc_pair).  The wave's block holds 32 consecutive pixels x 96
        // channels; tiles start at multiples of 32 and rows are 28 wide, so pixels 2k, 2k + 1 of the block are always a
        // horizontal pair of one image row.  Part A: 16 pairs x 16 channel quads (channels 0..63) -> the pair's SUM goes to
        // the half-width tensor; part B: 32 pixels x 8 quads (channels 64..95) -> dc_h1.  Both add the skip path's rank-one
        // share d[m] * u[c].  8 passes of one LDS read (two in part A), at most 2 fma per value and one 16-byte store.
        const int ca = (lane_e & 15) * 4, cb = 64 + (lane_e & 7) * 4;
        const float4 ua = gload4(rk1_u + ca), ub = gload4(rk1_u + cb);
        const bool whole = mbase + 32 <= Mtot;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int pp = it * 4 + (lane_e >> 4);                   // pair index inside the block
            const int m0p = mbase + 2 * pp;
            if (whole || m0p < Mtot) {                                // (Mtot is even: a pair is whole or absent)
                const f32x2 d2 = gload<f32x2>(rk1_d + m0p);
                const float ds = d2[0] + d2[1];
                const float4 v0 = *reinterpret_cast<const float4*>(T + (2 * pp) * EPI + ca);
                const float4 v1 = *reinterpret_cast<const float4*>(T + (2 * pp + 1) * EPI + ca);
                float4 o;
                o.x = fmaf(ds, ua.x, v0.x + v1.x); o.y = fmaf(ds, ua.y, v0.y + v1.y);
                o.z = fmaf(ds, ua.z, v0.z + v1.z); o.w = fmaf(ds, ua.w, v0.w + v1.w);
                gstore4(dc_pair + ((unsigned)(m0p >> 1) * 64 + ca), o);
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int px = it * 8 + (lane_e >> 3);
            const int m = mbase + px;
            if (whole || m < Mtot) {
                const float d = gload<float>(rk1_d + m);
                const float4 v = *reinterpret_cast<const float4*>(T + px * EPI + cb);
                float4 o;
                o.x = fmaf(d, ub.x, v.x); o.y = fmaf(d, ub.y, v.y); o.z = fmaf(d, ub.z, v.z); o.w = fmaf(d, ub.w, v.w);
                gstore4(dc_h1 + ((unsigned)m * 32 + (cb - 64)), o);
            }
        }
        return;
    }
    // Everything of p has been requested: wait for it HERE, once.  The requests sit under uniform branches, so without
    // this the compiler guards every pass's first use of p.rt / p.mk with s_waitcnt vmcnt(0) — which on gfx9
    // also waits for the previous pass's STORES (same counter).
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) only
    // The walk runs phase by phase over ALL passes of the tile (values in registers), not pass by pass: every uniform
    // "is this output wanted" branch then wraps NIT independent instruction chains that the scheduler can interleave.
    // Pass by pass, each chain (LDS read -> ReLU -> split -> store, ~30 dependent instructions) ran alone at one
    // instruction per ~10 cycles.  `full` (all 32 pixels exist) removes the per-lane_e bounds check from all but the last tile.
    float4 sacc[2][2];      // [slot][kind] partial sums of this lane_e's channel quad (bwd only)
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int kd = 0; kd < 2; ++kd) sacc[sl][kd] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 o1_gw = make_float4(0.f, 0.f, 0.f, 0.f);   // fused MSE backward (o1_tgt): sum d * value of this lane's channel quad ...
    float o1_gb = 0.f, o1_gl = 0.f;                    // ... sum d, sum (eps - tgt)^2 (lanes with lane % 8 == 0 only)
    auto walk = [&](auto full_c, auto group_c) {
        constexpr bool FULL = decltype(full_c)::value;
        constexpr int I0 = decltype(group_c)::value * GI;
        float4 v[GI];
        unsigned o[GI];
        bool ok[GI];
#pragma unroll
        for (int k = 0; k < GI; ++k) {
            const int e = (I0 + k) * 64 + lane_e;
            const int px = e / (N / 4), c = (e - px * (N / 4)) * 4;
            v[k] = *reinterpret_cast<const float4*>(T + px * EPI + c);
            o[k] = (unsigned)(mbase + px) * N + c;
            ok[k] = FULL || mbase + px < Mtot;
            if constexpr (PH) { o[k] = (unsigned)max(mpx[(I0 + k) & 3], 0) * N + c; ok[k] = mpx[(I0 + k) & 3] >= 0; }
        }
        if constexpr (S2D) if (s2_d != nullptr) {   // FIRST (the saved copy `aux` = the complete gradient): + (d summed over the source pixel's four output pixels) * u[c]
            const float4 u4 = gload4(s2_u + (lane_e & (N / 4 - 1)) * 4);   // (a lane's channel quad is the same in every pass)
#pragma unroll
            for (int k = 0; k < GI; ++k) {
                const float dq = __shfl(d4_lane, ((I0 + k) * 64 + lane_e) / (N / 4));
                v[k].x = fmaf(dq, u4.x, v[k].x); v[k].y = fmaf(dq, u4.y, v[k].y); v[k].z = fmaf(dq, u4.z, v[k].z); v[k].w = fmaf(dq, u4.w, v[k].w);
            }
        }
        if (a.relu) {
#pragma unroll
            for (int k = 0; k < GI; ++k) {
                v[k].x = fmaxf(v[k].x, 0.f); v[k].y = fmaxf(v[k].y, 0.f); v[k].z = fmaxf(v[k].z, 0.f); v[k].w = fmaxf(v[k].w, 0.f);
            }
        }
        if (a.aux != nullptr && !(a.ablate & 16)) {
#pragma unroll
            for (int k = 0; k < GI; ++k) if (ok[k]) gstore4(a.aux + o[k], v[k]);
        }
        if (a.mask_out != nullptr) {
#pragma unroll
            for (int k = 0; k < GI; ++k)
                if (ok[k])
                    gstore<unsigned char>(a.mask_out + (o[k] >> 2),
                                          (unsigned char)((v[k].x > 0.f ? 1 : 0) | (v[k].y > 0.f ? 2 : 0) |
                                                          (v[k].z > 0.f ? 4 : 0) | (v[k].w > 0.f ? 8 : 0)));
        }
        if (a.res != nullptr) {
#pragma unroll
            for (int k = 0; k < GI; ++k) {
                const float4 rz = p.rt[k];
                v[k].x += rz.x; v[k].y += rz.y; v[k].z += rz.z; v[k].w += rz.w;
            }
        }
        if constexpr (R1X) if (r1_x != nullptr) {   // rank-1 residual: the one-input-channel skip conv, recomputed (same fma as conv_first)
#pragma unroll
            for (int k = 0; k < GI; ++k) {
                const int c = (((I0 + k) * 64 + lane_e) % (N / 4)) * 4;
                const float4 w4 = gload4(r1_w + c), b4 = gload4(r1_b + c);
                v[k].x += fmaf(p.rx[k], w4.x, b4.x); v[k].y += fmaf(p.rx[k], w4.y, b4.y);
                v[k].z += fmaf(p.rx[k], w4.z, b4.z); v[k].w += fmaf(p.rx[k], w4.w, b4.w);
            }
        }
        if (bwd && a.out_s16_pre != nullptr) {   // the unmasked gradient's S16 twin (ConvArgs::out_s16_pre)
#pragma unroll
            for (int k = 0; k < GI; ++k) {
                const int c = (((I0 + k) * 64 + lane_e) % (N / 4)) * 4;
                if (ok[k]) gstore_s16_o(a.out_s16_pre, o[k], c, v[k]);
            }
        }
        if (bwd) {   // ReLU backward of the tensor this gradient belongs to + the sums its bias gradients need
#pragma unroll
            for (int k = 0; k < GI; ++k) {
                const int m = mbase + ((I0 + k) * 64 + lane_e) / (N / 4);
                const unsigned mk = p.mk[k];
                const float4 u = v[k];
                const float4 mv = make_float4((mk & 1u) ? u.x : 0.f, (mk & 2u) ? u.y : 0.f, (mk & 4u) ? u.z : 0.f,
                                              (mk & 8u) ? u.w : 0.f);
                const bool s0 = ok[k] && m < mnext, s1 = ok[k] && m >= mnext;   // slot 0: image of the first pixel; slot 1: the next
                sacc[0][0].x += s0 ? u.x : 0.f; sacc[0][0].y += s0 ? u.y : 0.f; sacc[0][0].z += s0 ? u.z : 0.f; sacc[0][0].w += s0 ? u.w : 0.f;
                sacc[0][1].x += s0 ? mv.x : 0.f; sacc[0][1].y += s0 ? mv.y : 0.f; sacc[0][1].z += s0 ? mv.z : 0.f; sacc[0][1].w += s0 ? mv.w : 0.f;
                sacc[1][0].x += s1 ? u.x : 0.f; sacc[1][0].y += s1 ? u.y : 0.f; sacc[1][0].z += s1 ? u.z : 0.f; sacc[1][0].w += s1 ? u.w : 0.f;
                sacc[1][1].x += s1 ? mv.x : 0.f; sacc[1][1].y += s1 ? mv.y : 0.f; sacc[1][1].z += s1 ? mv.z : 0.f; sacc[1][1].w += s1 ? mv.w : 0.f;
                v[k] = mv;
            }
        }
        if (a.out != nullptr) {
#pragma unroll
            for (int k = 0; k < GI; ++k) if (ok[k]) gstore4(a.out + o[k], v[k]);
        }
        if constexpr (R1) if (o1_out != nullptr) {   // the model's 1x1 output conv: 8 consecutive lanes hold a pixel's 32 channels
            const float4 w4 = gload4(o1_w + (lane_e & 7) * 4);
            const float ob = gload<float>(o1_b);
#pragma unroll
            for (int k = 0; k < GI; ++k) {
                float d = ((v[k].x * w4.x + v[k].y * w4.y) + v[k].z * w4.z) + v[k].w * w4.w;   // (conv_out_kernel's association)
                d += __shfl_xor(d, 1);
                d += __shfl_xor(d, 2);
                d += __shfl_xor(d, 4);
                if ((lane_e & 7) == 0 && ok[k]) gstore<float>(o1_out + (o[k] >> 5), d + ob);
                if constexpr (MSE) if (ok[k]) {   // every lane of the pixel holds d after the butterfly
                    const float df = (d + ob) - p.tg[k];
                    const float dl = df * __int_as_float(o1_dscale_bits);
                    o1_gw.x += dl * v[k].x; o1_gw.y += dl * v[k].y; o1_gw.z += dl * v[k].z; o1_gw.w += dl * v[k].w;
                    if ((lane_e & 7) == 0) {
                        o1_gb += dl; o1_gl += df * df;
                        gstore<float>(o1_deps + (o[k] >> 5), dl);
                    }
                }
            }
        }
        if (a.out_s16 != nullptr) {
            if (a.tb_out != nullptr) {
#pragma unroll
                for (int k = 0; k < GI; ++k) {
                    float4 tz = p.rt[k];
                    if (a.res != nullptr) {   // both a residual and a time-bias row (no UNet launch; C-ABI layer tests): fetch in place
                        const int e = (I0 + k) * 64 + lane_e;
                        const int px = e / (N / 4), c = (e - px * (N / 4)) * 4;
                        const int m = min(mbase + px, Mtot - 1);
                        tz = gload4(a.tb_out + (unsigned)((img0 + (m >= mnext ? 1 : 0)) * a.tb_out_stride + c));
                    }
                    v[k].x += tz.x; v[k].y += tz.y; v[k].z += tz.z; v[k].w += tz.w;
                }
            }
#pragma unroll
            for (int k = 0; k < GI; ++k) {
                const int c = (((I0 + k) * 64 + lane_e) % (N / 4)) * 4;
                if (ok[k]) gstore_s16_o(a.out_s16, o[k], c, v[k]);
            }
        }
    };
    static_assert(NIT % GI == 0, "pass groups");
    // (scheduling barrier between groups: interleaving TWO groups costs more registers than the N = 64 budget has)
    // (the inputs of group g + 1 are requested when group g is done: one group's worth of registers; the N = 64 budget
    //  has no room for a whole tile's — they were spilled as they arrived, one memory round trip each)
    auto groups = [&](auto full_c) {
        static_for<0, NIT / GI>([&](auto g) {
            constexpr int gi = decltype(g)::value;
            if constexpr (gi > 0) {
                // (a launch without epilogue inputs — rb4's data gradient — requests nothing and must not wait either: the
                //  vmcnt(0) would sit behind the previous group's stores, ~2 k cycles per group: tools/ws_probe.py)
                if (has_epi_in) {
                    preload(gi);
                    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): also retires the previous group's stores
                }
            }
            walk(full_c, g);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    if (mbase + 32 <= (PH ? Mtile : Mtot)) groups(std::true_type{});
    else groups(std::false_type{});
    if constexpr (MSE) {   // lanes with equal lane % 8 hold the same channel quad: butterfly over bits 3..5
        float4 r = o1_gw;
        float rb = o1_gb, rl = o1_gl;
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
            r.x += __shfl_xor(r.x, off); r.y += __shfl_xor(r.y, off); r.z += __shfl_xor(r.z, off); r.w += __shfl_xor(r.w, off);
            rb += __shfl_xor(rb, off); rl += __shfl_xor(rl, off);
        }
        if (lane_e < 8 && mbase < Mtot) {
            float* const row = o1_sums + ((unsigned)mbase >> 5) * 40u;
            gstore4(row + lane_e * 4, r);
            if (lane_e == 0) { gstore<float>(row + 32, rb); gstore<float>(row + 33, rl); }
        }
    }
    if (bwd && a.sums != nullptr) {
        // lanes with equal lane_e % (N/4) hold the same channel quad of different pixels: butterfly over the rest
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int kd = 0; kd < 2; ++kd) {
                float4 r = sacc[sl][kd];
#pragma unroll
                for (int off = N / 4; off < 64; off <<= 1) {
                    r.x += __shfl_xor(r.x, off); r.y += __shfl_xor(r.y, off);
                    r.z += __shfl_xor(r.z, off); r.w += __shfl_xor(r.w, off);
                }
                if (lane_e < N / 4 && mbase < Mtot) {
                    const unsigned grp = (unsigned)mbase >> 5;
                    gstore4(a.sums + (unsigned)(((grp * 2 + sl) * 2 + kd) * N + lane_e * 4), r);
                }
            }
    }
    stamp();                     // 9
    stamp();                     // 10 (second M tile of the former 4-wave layout: none)
    if constexpr (SKIP) {   // second accumulator: skip_out = 1x1 conv (+ its bias, already accumulated), same transposed walk
        to_lds(sk.acc2 + mt * NT);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = it * 64 + lane_e;
            const int px = e / (N / 4), c = (e - px * (N / 4)) * 4;
            int m = mbase + px;
            if constexpr (PH) m = mpx[it & 3] >= 0 ? mpx[it & 3] : Mtot;
            const float4 v = *reinterpret_cast<const float4*>(T + px * EPI + c);
            if (m < Mtot) gstore4(a.skip_out + ((unsigned)m * N + c), v);
        }
    }
    });   // M tiles
    stamp();   // stores issued
    if constexpr (PROBE) if (a.ablate & 16) __builtin_amdgcn_s_waitcnt(0);   // stores acknowledged
    stamp();
}

template <int HW, int NT, bool SKIP, int MT = 1>
int launch_conv_t(const ConvArgs& a, hipStream_t st) {
    using G = Geo<HW>;
    constexpr int NRv = MT == 1 ? G::NR : ConvRows<HW, MT>::NR;
    constexpr size_t lds_op = (size_t)NRv * G::WP * PIXB + (size_t)9 * NT * 2048 + (SKIP ? (size_t)NT * 2048 : 0) +
                              (size_t)(CONV_THREADS / 64) * 32 * sizeof(int);   // operand images + the waves' row tables
    constexpr size_t lds_epi = (size_t)(CONV_THREADS / 64) * 32 * (NT * 32 + 4) * sizeof(float);   // per-wave transpose blocks
    constexpr size_t lds = lds_op > lds_epi ? lds_op : lds_epi;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s16_kernel<HW, NT, SKIP, false, MT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 110000);
        if (e != hipSuccess) {
            tdm_set_error("conv_s16: hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
            return 100 + (int)e;
        }
        attr_set = true;
    }
    const long Mtot = (long)a.B * G::H * G::W;
    const int ntiles = (int)((Mtot + TILE_PX * MT - 1) / (TILE_PX * MT));
    // ablate & 32 (probe): one workgroup per CU (LDS request > half of 160 KB) — phase times without a co-resident workgroup
    const size_t lds_req = (TDM_ABLATE(a.ablate) & 32) ? (size_t)110000 : lds;
#if TDM_DIAG_BUILD
    if constexpr (((HW == 28 && NT == 1) || (HW == 14 && NT == 2)) && !SKIP && MT == 1) {
        if (a.ablate & 16) {   // the instrumented instantiation (diagnostics only)
            static bool probe_attr = false;
            if (!probe_attr) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s16_kernel<HW, NT, SKIP, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 110000);
                probe_attr = true;
            }
            hipLaunchKernelGGL((conv_s16_kernel<HW, NT, SKIP, true>), dim3(ntiles), dim3(CONV_THREADS), lds_req, st, a);
            TDM_CHECK_LAUNCH("conv_s16(probe)");
            return 0;
        }
    }
#endif
    if constexpr (HW == 28 && NT == 1 && !SKIP && MT == 1) {
        if (a.o1_tgt != nullptr) {   // rb4.conv2 of a train step: output conv + MSE backward in the epilogue
            static bool mse_attr = false;
            if (!mse_attr) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s16_kernel<HW, NT, SKIP, false, 1, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 110000);
                if (e != hipSuccess) {
                    tdm_set_error("conv_s16: hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
                    return 100 + (int)e;
                }
                mse_attr = true;
            }
            hipLaunchKernelGGL((conv_s16_kernel<HW, NT, SKIP, false, 1, true>), dim3(ntiles), dim3(CONV_THREADS), lds_req, st, a);
            TDM_CHECK_LAUNCH("conv_s16(mse)");
            return 0;
        }
    }
    hipLaunchKernelGGL((conv_s16_kernel<HW, NT, SKIP, false, MT>), dim3(ntiles), dim3(CONV_THREADS), lds_req, st, a);
    TDM_CHECK_LAUNCH("conv_s16");
    return 0;
}

// rb4.conv1's forward in the phase form (ConvArgs::up_phase; the kernel's PH instantiation)
int launch_conv_phase(const ConvArgs& a, hipStream_t st) {
    using G = Geo<28>;
    constexpr size_t lds_op = (size_t)16 * G::WP * PIXB + (size_t)16 * 2048 + 2048 + (size_t)(CONV_THREADS / 64) * 32 * sizeof(int);
    constexpr size_t lds_epi = (size_t)(CONV_THREADS / 64) * 32 * (32 + 4) * sizeof(float);
    constexpr size_t lds = lds_op > lds_epi ? lds_op : lds_epi;
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s16_kernel<28, 1, true, false, 1, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 110000);
        if (e != hipSuccess) {
            tdm_set_error("conv_s16(phase): hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
            return 100 + (int)e;
        }
        attr_set = true;
    }
    const long Msrc = (long)a.B * 196;
    const int ntiles = (int)((Msrc + 63) / 64);
    hipLaunchKernelGGL((conv_s16_kernel<28, 1, true, false, 1, false, true>), dim3(ntiles), dim3(CONV_THREADS), lds, st, a);
    TDM_CHECK_LAUNCH("conv_s16(phase)");
    return 0;
}

// rb4.conv1's data gradient w.r.t. the up-sampled source at source resolution (ConvArgs::s2d; the kernel's S2D instantiation)
int launch_conv_s2d(const ConvArgs& a, hipStream_t st) {
    using G = Geo<14>;
    constexpr size_t lds_op = (size_t)G::NR * G::WP * PIXB + (size_t)9 * 2 * 2048 + (size_t)(CONV_THREADS / 64) * 32 * sizeof(int);
    constexpr size_t lds_epi = (size_t)(CONV_THREADS / 64) * 32 * (64 + 4) * sizeof(float);
    constexpr size_t lds = lds_op > lds_epi ? lds_op : lds_epi;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s16_kernel<14, 2, false, false, 1, false, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 110000);
        if (e != hipSuccess) {
            tdm_set_error("conv_s16(s2d): hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
            return 100 + (int)e;
        }
        attr_set = true;
    }
    const long Mtot = (long)a.B * 196;
    const int ntiles = (int)((Mtot + TILE_PX - 1) / TILE_PX);
    hipLaunchKernelGGL((conv_s16_kernel<14, 2, false, false, 1, false, false, true>), dim3(ntiles), dim3(CONV_THREADS), lds, st, a);
    TDM_CHECK_LAUNCH("conv_s16(s2d)");
    return 0;
}

// ---------------------------------------------------------------------------
// weight gradient  dW[tap][ci][co] = sum_p A[p + tap][ci] * G[p][co]  (A, G are S16)
// ---------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 tr_pair(const char* p0, const char* p1) {
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
    s16x8 r;
    r[0] = lo4[0]; r[1] = lo4[1]; r[2] = lo4[2]; r[3] = lo4[3];
    r[4] = hi4[0]; r[5] = hi4[1]; r[6] = hi4[2]; r[7] = hi4[3];
    return __builtin_bit_cast(bf16x8, r);
}

// ---------------------------------------------------------------------------
// weight gradient, producer / consumer form (default).  The kernel above runs load -> LDS write -> MFMA as three phases
// that all 8 waves go through together (barriers), so the matrix pipe idles while a tile is staged (~16 % busy).  Here
// a workgroup is 16 waves: waves 8..15 PRODUCE — they fetch tile k + 2 into registers and write tile k + 1 into the
// other half of a double-buffered LDS image — while waves 0..7 CONSUME tile k (transposed LDS reads + MFMA).  One
// barrier per 128-pixel tile; the staging work (index math, 47 KB of slow ds_write_b128) runs under the MFMAs.
// ---------------------------------------------------------------------------
constexpr int W2_TP = 128;   // pixels per tile
template <int HW> struct W2Geo;
template <> struct W2Geo<28> { static constexpr int NRW = 10; };   // staged rows: 128 pixels + halo + one image seam
template <> struct W2Geo<14> { static constexpr int NRW = 15; };

template <int HW, bool SK2>
__global__ __launch_bounds__(1024) void wgrad2_s16_kernel(WgradArgs a) {
    using G = Geo<HW>;
    constexpr int NPX = W2Geo<HW>::NRW * G::WP;
    constexpr int APL = NPX * 64;            // one plane (hi or lo) of the activation image
    constexpr int GPL = W2_TP * 64;
    constexpr int G2O = 2 * APL + 2 * GPL;                     // second gradient image (fused 1x1 skip), SK2 only
    constexpr int PXO = G2O + (SK2 ? 2 * GPL : 0);            // staged-pixel table
    constexpr int BUF = PXO + W2_TP * (int)sizeof(int);
    extern __shared__ float4 smem4[];
    char* const lds = reinterpret_cast<char*>(smem4);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup -> (slab slot, channel-tile combination).  The gridDim.y combinations of one slot walk the SAME pixel tiles
    // (same gradient tiles for the ci tiles, same activation tiles for the co tiles); in grid order they are gridDim.x
    // workgroups apart, i.e. — one 1024-thread workgroup per CU — in different rounds when the grid exceeds the chip
    // (rb4.conv1's up(h3) part: 256 x 2: the 51 MB gradient tensor came from HBM twice).  Dispatch order L is therefore
    // re-dealt: XCD L % 8 runs its slots' combinations back to back, so they are co-resident and share through that L2.
    int comb = blockIdx.y, bxs = blockIdx.x;
    if ((gridDim.x & 7) == 0) {
        const int L = blockIdx.x + gridDim.x * blockIdx.y;
        const int k = L >> 3;
        comb = k % (int)gridDim.y;
        bxs = (L & 7) + 8 * (k / (int)gridDim.y);        // (xcd_remap below turns this into the XCD's contiguous slot range)
    }
    const int ci_tile = comb % a.nci, co_tile = comb / a.nci;
    const int ci0 = ci_tile * 32, co0 = co_tile * 32;
    const int Mtot = a.B * G::H * G::W;
    const int ntiles = (Mtot + W2_TP - 1) / W2_TP;
    const int taps = a.a.taps;
    const int step = gridDim.x;
    // Workgroups of one XCD take NEIGHBOURING tiles at the same time (slot = contiguous range per XCD): a 128-pixel tile
    // stages 10 rows for 4.6 rows of payload, and with blockIdx-ordered tiles the 2.2x halo re-reads went to eight
    // different L2s, i.e. to HBM — the kernel was bandwidth-bound on its own halo.
    const int slot = xcd_remap(bxs, gridDim.x);
    const int nk = (ntiles - slot + step - 1) / step;   // tiles of this workgroup: slot + k * step
    float* const slab = a.slab + (long)slot * a.slab_stride;

    if (wave >= 8) {
        // ------------------------------- producers -------------------------------
        const int ptid = tid - 512;
        const ConvSrc& s = a.a;
        const float* a_ptr = s.ptr; const float* g_ptr = a.g;
        int a_C = s.C, a_c0 = s.c0, a_up = s.up, g_C = a.Cout, nB = a.B;
        TDM_PIN(a_ptr); TDM_PIN(g_ptr); TDM_PIN(a_C); TDM_PIN(a_c0); TDM_PIN(a_up); TDM_PIN(g_C); TDM_PIN(nB);
        const int Hs = G::H >> a_up, Ws = G::W >> a_up;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_ptr), 0, nB * Hs * Ws * a_C * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g_ptr), 0, Mtot * g_C * 4, 0x00020000);
        const float* g2_ptr = a.g2;
        TDM_PIN(g2_ptr);
        const __amdgpu_buffer_rsrc_t rsG2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g2_ptr), 0, SK2 ? Mtot * g_C * 4 : 0, 0x00020000);
        // staging role: piece8 = ptid & 7 -> 16-channel group (piece8 >> 2), 16-byte piece of the group (piece8 & 3:
        // 0,1 = hi halves, 2,3 = lo halves); destination plane / offset inside a 64-byte pixel row
        const int piece8 = ptid & 7, grp = piece8 >> 2, pq = piece8 & 3;
        const int dcol = grp * 32 + (pq & 1) * 16;
        const int dplane_a = (pq >= 2) ? APL : 0, dplane_g = (pq >= 2) ? GPL : 0;
        const int a_col = a_c0 + ci0 + grp * 16 + pq * 4;   // float column of this thread's piece inside a pixel row of A
        const int g_col = co0 + grp * 16 + pq * 4;
        constexpr int NA = (NPX * 8 + 511) / 512;
        constexpr int NG = W2_TP * 8 / 512;
        struct Stage { u32x4 pa[NA]; u32x4 pg[NG]; u32x4 pg2[SK2 ? NG : 1]; int pix; int nelem; };
        Stage s0, s1;
        // Per-tile index work is what bounds this kernel once the phases overlap (the staging waves' address arithmetic
        // competes with the MFMA waves for vector issue), so everything that does not depend on the tile is computed
        // ONCE: piece e = ptid + 512 i sits at staged position (ptid >> 3) + 64 i = (row lr_i, column pc_i); its column
        // offset and column validity are tile-invariant.  Per tile only the staged ROWS change: lanes 0..NRW-1 of each
        // producer wave work out (offset, valid) of one row each and park the pair in a wave-private LDS table; a piece
        // then costs one 8-byte LDS read, an add and a select.
        constexpr int NRW = W2Geo<HW>::NRW;
        int* const rowtab = reinterpret_cast<int*>(lds + 2 * BUF) + (wave - 8) * 32;   // [16 rows][offset, nelem-limit flag]
        int lr_i[NA], col_i[NA];
        unsigned colok = 0u;
        {
            int lr = (ptid >> 3) / G::WP;
            int pc = (ptid >> 3) - lr * G::WP;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                lr_i[i] = min(lr, NRW - 1) * 8;                     // byte offset of the row's table entry
                col_i[i] = (((pc - 1) >> a_up) * a_C + a_col) * 4;   // bytes
                colok |= (pc >= 1 && pc <= G::W && lr < NRW) ? (1u << i) : 0u;
                pc += 64 % G::WP;
                lr += 64 / G::WP;
                if (pc >= G::WP) { pc -= G::WP; ++lr; }
            }
        }
        auto prefetch = [&](Stage& st, int k) {
            const int t = min(slot + k * step, ntiles - 1);   // (past the end: re-read the last tile, never staged)
            const int m0 = t * W2_TP;
            const int mlast = min(m0 + W2_TP - 1, Mtot - 1);
            const int tb0 = m0 / (G::H * G::W);
            const int rem0 = m0 - tb0 * (G::H * G::W);
            const int ty0 = rem0 / G::W;
            const int x0 = rem0 - ty0 * G::W;
            const int PR0 = tb0 * G::HP + ty0;
            const int nrows = min(padded_row<HW>(mlast) - PR0 + 2, NRW);
            st.nelem = nrows * G::WP * 8;
            // row table: staged row r = lane is padded row PR0 + r of the tall image
            if (lane < 16) {
                int py = ty0 + lane, b = tb0;
                if (py >= G::HP) { py -= G::HP; ++b; }
                if (HW == 14 && py >= G::HP) { py -= G::HP; ++b; }
                const bool ok = lane < nrows && py >= 1 && py <= G::H && b < nB;
                const int off = __mul24(__mul24(__mul24(b, Hs) + ((py - 1) >> a_up), Ws), a_C) * 4;
                *reinterpret_cast<int2*>(rowtab + lane * 2) = make_int2(ok ? off : (int)0x80000000, 0);
            }
            // staged-pixel index of tile pixel `ptid` (< 128): q = x0 + ptid columns past the start of row ty0
            st.pix = 0;
            if (ptid < W2_TP) {
                const int q = x0 + min(ptid, mlast - m0);   // (pixels past the end: the last real one; their G rows read as zeros)
                const int dr = (q * (HW == 28 ? 2341 : 4682)) >> 16;   // q / W for q < W + 128 (checked exhaustively)
                const int x = q - dr * G::W;
                int y = ty0 + dr, seam = 0;
                if (y >= G::H) { y -= G::H; seam = G::HP; }             // into the next image: + one padded image of rows
                st.pix = (seam + y + 1 - ty0) * G::WP + x + 1;
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int roff = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(rowtab) + lr_i[i]);
                const bool ok = ((colok >> i) & 1u) != 0u;
                st.pa[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? (roff + col_i[i]) : (int)0x80000000, 0, 0));
            }
#pragma unroll
            for (int i = 0; i < NG; ++i) {   // pixels past the end are past num_records: zeros
                const int m = m0 + ((ptid + 512 * i) >> 3);
                st.pg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, (__mul24(m, g_C) + g_col) * 4, 0, 0));
                if constexpr (SK2)
                    st.pg2[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG2, (__mul24(m, g_C) + g_col) * 4, 0, 0));
            }
        };
        auto write = [&](const Stage& st, int buf) {
            char* const base = lds + buf * BUF;
            int* const pixoff = reinterpret_cast<int*>(base + PXO);
            if (ptid < W2_TP) pixoff[ptid] = st.pix;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int e = ptid + 512 * i;
                if (e < st.nelem) *reinterpret_cast<u32x4*>(base + dplane_a + dcol + (e >> 3) * 64) = st.pa[i];
            }
#pragma unroll
            for (int i = 0; i < NG; ++i)
            {
                *reinterpret_cast<u32x4*>(base + 2 * APL + dplane_g + dcol + ((ptid + 512 * i) >> 3) * 64) = st.pg[i];
                if constexpr (SK2) *reinterpret_cast<u32x4*>(base + G2O + dplane_g + dcol + ((ptid + 512 * i) >> 3) * 64) = st.pg2[i];
            }
        };
        prefetch(s0, 0);
        prefetch(s1, 1);
        write(s0, 0);
        __syncthreads();                       // tile 0 staged
        const bool idle = a.b_off == -3;   // probe: consumers only
        for (int k = 0; k < nk; k += 2) {
            // consumers: tile k from buffer 0
            if (!idle) prefetch(s0, k + 2);
            if (k + 1 < nk && !idle) write(s1, 1);
            __syncthreads();
            if (k + 1 >= nk) break;
            // consumers: tile k + 1 from buffer 1
            if (!idle) prefetch(s1, k + 3);
            if (k + 2 < nk && !idle) write(s0, 0);
            __syncthreads();
        }
        // the consumers' final reduction: same barrier count
        const int nbar = (taps == 9) ? 10 : 2;
        for (int i = 0; i < nbar; ++i) __syncthreads();
        return;
    }

    // ------------------------------- consumers -------------------------------
    const int g4 = lane >> 4, cb = g4 & 1, hh = g4 >> 1, q = (lane >> 2) & 3, pcq = lane & 3;
    const int colb = (cb * 16 + pcq * 4) * 2;
    const int tgrp = wave >> 2, wq = wave & 3;
    const int tp0 = tgrp * 5, ntap = tgrp ? 4 : 5;
    f32x16 acc[5];
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    // (ONE MFMA site per accumulator: tap ranges as compile-time constants would fold every tap offset into the reads'
    //  offset fields, but the three instantiations then disagree on where the accumulators live — 388 bytes of spills)
    auto kstep = [&](const char* base, int ks, int t0, int nt) {   // 16 pixels ks*16.., taps t0 .. t0+nt-1 -> acc[0..nt-1]
        const char* Ahi = base; const char* Alo = base + APL;
        const char* Ghi = base + 2 * APL; const char* Glo = Ghi + GPL;
        const int* pixoff = reinterpret_cast<const int*>(base + PXO);
        const int p0 = ks * 16 + hh * 8 + q;
        const int gb0 = p0 * 64 + colb, gb1 = (p0 + 4) * 64 + colb;
        const bf16x8 gh = tr_pair(Ghi + gb0, Ghi + gb1);
        const bf16x8 gl = tr_pair(Glo + gb0, Glo + gb1);
        const int ab0 = pixoff[p0] * 64 + colb, ab1 = pixoff[p0 + 4] * 64 + colb;
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            if (d < nt) {
                const int tp = t0 + d;
                const int to = ((tp / 3 - 1) * G::WP + (tp % 3 - 1)) * 64;
                const bf16x8 ah = tr_pair(Ahi + ab0 + to, Ahi + ab1 + to);
                const bf16x8 al = tr_pair(Alo + ab0 + to, Alo + ab1 + to);
                acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, gh, acc[d], 0, 0, 0);
                acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gl, acc[d], 0, 0, 0);
                acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gh, acc[d], 0, 0, 0);
            }
        }
        if constexpr (SK2) {
            if (nt == 4) {   // tap group 1: its fifth accumulator takes the block's 1x1 skip conv (centre-tap pixels x G2)
                const bf16x8 g2h = tr_pair(base + G2O + gb0, base + G2O + gb1);
                const bf16x8 g2l = tr_pair(base + G2O + GPL + gb0, base + G2O + GPL + gb1);
                const bf16x8 ah = tr_pair(Ahi + ab0, Ahi + ab1);
                const bf16x8 al = tr_pair(Alo + ab0, Alo + ab1);
                acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, g2h, acc[4], 0, 0, 0);
                acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, g2l, acc[4], 0, 0, 0);
                acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, g2h, acc[4], 0, 0, 0);
            }
        }
    };
    // ONE kstep site for every accumulator: with compute(0) / compute(1) (and the 1x1 path) as separate inlined sites the
    // compiler gave each site its own accumulator registers and copied all 80 of them between sites every tile, each
    // copy waiting out the MFMA that produced it (s_nop 9..11).  Buffer, K-step range and tap range are runtime values.
    const int ks0 = (taps == 9) ? wq * 2 : wave;   // 3x3: 2 of the tile's 8 K-steps; 1x1: the 8 waves split them
    const int nks = (taps == 9) ? 2 : 1;
    const int kt0 = (taps == 9) ? tp0 : 4, knt = (taps == 9) ? ntap : 1;
    const bool cidle = a.b_off == -2;   // probe: producers only
    __syncthreads();                           // tile 0 staged
    for (int k = 0; k < nk; ++k) {
        const char* base = lds + (k & 1) * BUF;
        if (!cidle)
            for (int jk = 0; jk < nks; ++jk) kstep(base, ks0 + jk, kt0, knt);
        __syncthreads();
    }

    // partial sums of the waves that share a tap -> LDS -> fixed-order sum -> this workgroup's slab
    float* red = reinterpret_cast<float*>(smem4);   // 8 waves x 1024 floats
    if (taps == 9) {
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[d][r];
            __syncthreads();
            // tap d of group 0 (waves 0-3) and tap 5 + d of group 1 (waves 4-7; d < 4): 2 x 1024 outputs
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                const int o = tid + 512 * k2;
                const int grp2 = o >> 10, idx = o & 1023;
                if (grp2 == 0 || d < 4 || SK2) {
                    const float* rb = red + grp2 * 4096 + idx;
                    const float sum = (rb[0] + rb[1024]) + (rb[2048] + rb[3072]);
                    const int r = idx >> 6, ln = idx & 63;
                    const int ci = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
                    const int co = ln & 31;
                    const int wt = grp2 * 5 + d;
                    if (grp2 == 1 && d == 4) slab[a.w_off2 + (long)(a.a.w_r0 + ci0 + ci) * a.Cout + co0 + co] = sum;   // 1x1 skip
                    else slab[a.w_off + (long)(wt * a.a.w_rows + a.a.w_r0 + ci0 + ci) * a.Cout + co0 + co] = sum;
                }
            }
        }
    } else {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[0][r];
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
            slab[a.w_off + (long)(a.a.w_r0 + ci0 + ci) * a.Cout + co0 + co] = sum;
        }
    }
}


// ---------------------------------------------------------------------------
// weight gradient of rb4.conv1's up-sampled source in the parity form (WgradArgs::s2d; the transpose of the "PH" / "S2D" conv
// kernels above).  dW[ky][kx] = sum_p up(h3)[p + (ky-1, kx-1)] (x) g[p] over the 28x28 pixels p; with up(h3)[y][x] = h3[y>>1][x>>1] the
// same sum is, per parity sub-image g_pq[i'][j'] = g[2i'+p][2j'+q] of the output gradient, a 14x14 weight gradient with FOUR taps:
//     dWeff[(p,a),(q,b)] = sum_{i',j'} h3[i' - (a - p)][j' - (b - q)] (x) g_pq[i'][j'],        a, b in {0, 1}
// and dW[ky][kx] is the sum of the four dWeff whose (p, a) reaches ky ((0,1),(1,1) -> 0; (0,0),(1,1) -> 1; (0,0),(1,0) -> 2) and
// whose (q, b) reaches kx: 16 x 196 instead of 9 x 784 tap-pixel products per image.  Same producer / consumer structure as
// wgrad2_s16_kernel: a tile is 64 positions of the 14x14 raster; producers stage the haloed h3 image ONCE and the four parity
// sub-images of g (a stride-2 gather); consumer wave w owns sub-image w >> 1 and row offset a = w & 1, i.e. two effective taps
// (b = 0, 1) over all four K steps of a tile — no K split, so the final step only combines the 16 effective taps into the nine
// (ky, kx) slots (fixed order) and writes them to the workgroup's slab.
// ---------------------------------------------------------------------------
constexpr int W3_TP = 64;     // positions per tile
constexpr int W3_NRW = 10;    // staged rows of the 14x14 image: <= 6 rows of positions + halo + one image seam
__global__ __launch_bounds__(1024) void wgrad_s2d_kernel(WgradArgs a) {
    using G = Geo<14>;
    constexpr int NPX = W3_NRW * G::WP;
    constexpr int APL = NPX * 64;            // one plane (hi or lo) of the activation image
    constexpr int GPL = W3_TP * 64;          // one plane of ONE parity sub-image of the gradient tile
    constexpr int PXO = 2 * APL + 8 * GPL;   // staged-pixel table
    constexpr int BUF = PXO + W3_TP * (int)sizeof(int);
    extern __shared__ float4 smem4[];
    char* const lds = reinterpret_cast<char*>(smem4);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci_tile = blockIdx.y % a.nci, co_tile = blockIdx.y / a.nci;
    const int ci0 = ci_tile * 32, co0 = co_tile * 32;
    const int Mtot = a.B * G::H * G::W;
    const int ntiles = (Mtot + W3_TP - 1) / W3_TP;
    const int step = gridDim.x;
    const int slot = xcd_remap(blockIdx.x, gridDim.x);
    const int nk = (ntiles - slot + step - 1) / step;   // tiles of this workgroup: slot + k * step
    float* const slab = a.slab + (long)slot * a.slab_stride;

    if (wave >= 8) {
        // ------------------------------- producers -------------------------------
        const int ptid = tid - 512;
        const float* a_ptr = a.a.ptr; const float* g_ptr = a.g;
        int a_C = a.a.C, a_c0 = a.a.c0, g_C = a.Cout, nB = a.B;
        TDM_PIN(a_ptr); TDM_PIN(g_ptr); TDM_PIN(a_C); TDM_PIN(a_c0); TDM_PIN(g_C); TDM_PIN(nB);
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_ptr), 0, nB * G::H * G::W * a_C * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g_ptr), 0, nB * 784 * g_C * 4, 0x00020000);
        const int piece8 = ptid & 7, grp = piece8 >> 2, pq = piece8 & 3;
        const int dcol = grp * 32 + (pq & 1) * 16;
        const int dplane_a = (pq >= 2) ? APL : 0, dplane_g = (pq >= 2) ? GPL : 0;
        const int a_col = a_c0 + ci0 + grp * 16 + pq * 4;   // float column of this thread's piece inside a pixel row of A
        const int g_col = co0 + grp * 16 + pq * 4;
        constexpr int NA = (NPX * 8 + 511) / 512;
        struct Stage { u32x4 pa[NA]; u32x4 pg[4]; int pix; int nelem; };
        Stage s0, s1;
        int* const rowtab = reinterpret_cast<int*>(lds + 2 * BUF) + (wave - 8) * 32;
        int lr_i[NA], col_i[NA];
        unsigned colok = 0u;
        {
            int lr = (ptid >> 3) / G::WP;
            int pc = (ptid >> 3) - lr * G::WP;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                lr_i[i] = min(lr, W3_NRW - 1) * 4;                   // byte offset of the row's table entry
                col_i[i] = ((pc - 1) * a_C + a_col) * 4;             // bytes
                colok |= (pc >= 1 && pc <= G::W && lr < W3_NRW) ? (1u << i) : 0u;
                pc += 64 % G::WP;
                lr += 64 / G::WP;
                if (pc >= G::WP) { pc -= G::WP; ++lr; }
            }
        }
        auto prefetch = [&](Stage& st, int k) {
            const int t = min(slot + k * step, ntiles - 1);   // (past the end: re-read the last tile, never staged)
            const int m0 = t * W3_TP;
            const int mlast = min(m0 + W3_TP - 1, Mtot - 1);
            const int tb0 = m0 / (G::H * G::W);
            const int rem0 = m0 - tb0 * (G::H * G::W);
            const int ty0 = rem0 / G::W;
            const int x0 = rem0 - ty0 * G::W;
            const int PR0 = tb0 * G::HP + ty0;
            const int nrows = min(padded_row<14>(mlast) - PR0 + 2, W3_NRW);
            st.nelem = nrows * G::WP * 8;
            if (lane < 16) {   // row table: staged row r = lane is padded row PR0 + r of the tall image
                int py = ty0 + lane, b = tb0;
                if (py >= G::HP) { py -= G::HP; ++b; }
                const bool ok = lane < nrows && py >= 1 && py <= G::H && b < nB;
                rowtab[lane] = ok ? __mul24(__mul24(__mul24(b, G::H) + (py - 1), G::W), a_C) * 4 : (int)0x80000000;
            }
            // tile position ptid >> 3 (< 64): its staged-pixel index (threads ptid < 64 keep the one of position ptid) and
            // the 28x28 pixel of its parity-(0, 0) gradient sample
            const int pos = ptid >> 3;
            auto locate = [&](int tp, int& pixidx, int& m28) {
                const int q = x0 + min(tp, mlast - m0);              // columns past the start of row ty0
                const int dr = (q * 4682) >> 16;                     // q / 14 for q < 14 + 128
                const int x = q - dr * G::W;
                int y = ty0 + dr, bb = tb0, seam = 0;
                if (y >= G::H) { y -= G::H; seam = G::HP; ++bb; }
                pixidx = (seam + y + 1 - ty0) * G::WP + x + 1;
                m28 = (bb * 28 + 2 * y) * 28 + 2 * x;
            };
            int pixg, m28;
            locate(pos, pixg, m28);
            st.pix = 0;
            if (ptid < W3_TP) { int m28_; locate(ptid, st.pix, m28_); }
            const bool live = m0 + pos <= mlast;                     // positions past the end: gradient rows read as zeros
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int roff = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(rowtab) + lr_i[i]);
                const bool ok = ((colok >> i) & 1u) != 0u && roff >= 0;
                st.pa[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? (roff + col_i[i]) : (int)0x80000000, 0, 0));
            }
#pragma unroll
            for (int par = 0; par < 4; ++par) {                      // sub-image (p, q) = (par >> 1, par & 1)
                const int m = m28 + (par >> 1) * 28 + (par & 1);
                st.pg[par] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, live ? (__mul24(m, g_C) + g_col) * 4 : (int)0x80000000, 0, 0));
            }
        };
        auto write = [&](const Stage& st, int buf) {
            char* const base = lds + buf * BUF;
            int* const pixoff = reinterpret_cast<int*>(base + PXO);
            if (ptid < W3_TP) pixoff[ptid] = st.pix;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int e = ptid + 512 * i;
                if (e < st.nelem) *reinterpret_cast<u32x4*>(base + dplane_a + dcol + (e >> 3) * 64) = st.pa[i];
            }
#pragma unroll
            for (int par = 0; par < 4; ++par)
                *reinterpret_cast<u32x4*>(base + 2 * APL + par * (2 * GPL) + dplane_g + dcol + (ptid >> 3) * 64) = st.pg[par];
        };
        prefetch(s0, 0);
        prefetch(s1, 1);
        write(s0, 0);
        __syncthreads();                       // tile 0 staged
        for (int k = 0; k < nk; k += 2) {
            prefetch(s0, k + 2);
            if (k + 1 < nk) write(s1, 1);
            __syncthreads();
            if (k + 1 >= nk) break;
            prefetch(s1, k + 3);
            if (k + 2 < nk) write(s0, 0);
            __syncthreads();
        }
        __syncthreads();                       // the consumers' combination step: one barrier
        return;
    }

    // ------------------------------- consumers -------------------------------
    const int g4 = lane >> 4, cb = g4 & 1, hh = g4 >> 1, q = (lane >> 2) & 3, pcq = lane & 3;
    const int colb = (cb * 16 + pcq * 4) * 2;
    const int par = wave >> 1, ap = wave & 1;              // sub-image (p, q) = (par >> 1, par & 1); row offset index a
    const int dy = ap - (par >> 1), qx = par & 1;          // h3 row read: i' - dy; columns: j' - (b - q), b = 0, 1
    f32x16 acc[2];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    __syncthreads();                           // tile 0 staged
    for (int k = 0; k < nk; ++k) {
        const char* base = lds + (k & 1) * BUF;
        const char* Ahi = base; const char* Alo = base + APL;
        const char* Ghi = base + 2 * APL + par * (2 * GPL); const char* Glo = Ghi + GPL;
        const int* pixoff = reinterpret_cast<const int*>(base + PXO);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int p0 = ks * 16 + hh * 8 + q;
            const int gb0 = p0 * 64 + colb, gb1 = (p0 + 4) * 64 + colb;
            const bf16x8 gh = tr_pair(Ghi + gb0, Ghi + gb1);
            const bf16x8 gl = tr_pair(Glo + gb0, Glo + gb1);
            const int ab0 = pixoff[p0] * 64 + colb, ab1 = pixoff[p0 + 4] * 64 + colb;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int to = ((-dy) * G::WP - (b - qx)) * 64;
                const bf16x8 ah = tr_pair(Ahi + ab0 + to, Ahi + ab1 + to);
                const bf16x8 al = tr_pair(Alo + ab0 + to, Alo + ab1 + to);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, gh, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gl, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gh, acc[b], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // the 16 effective taps -> LDS -> nine (ky, kx) sums in fixed order -> this workgroup's slab
    float* red = reinterpret_cast<float*>(smem4);   // [16 effective taps = par * 4 + a * 2 + b][1024]
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((par * 4 + ap * 2 + b) << 10) + r * 64 + lane] = acc[b][r];
    __syncthreads();
    // (ky or kx) <- the two (parity, offset) pairs that reach it, as parity * 2 + offset
    const int src0[3] = {1, 0, 0}, src1[3] = {3, 3, 2};
    for (int o = tid; o < 9 * 1024; o += 512) {
        const int tap9 = o >> 10, idx = o & 1023;
        const int ky = tap9 / 3, kx = tap9 - ky * 3;
        float sum = 0.f;
#pragma unroll
        for (int yi = 0; yi < 2; ++yi)
#pragma unroll
            for (int xi = 0; xi < 2; ++xi) {
                const int yr = yi ? src1[ky] : src0[ky], xr = xi ? src1[kx] : src0[kx];
                const int e = (((yr >> 1) * 2 + (xr >> 1)) * 4) + (yr & 1) * 2 + (xr & 1);
                sum += red[(e << 10) + idx];
            }
        const int r = idx >> 6, ln = idx & 63;
        const int ci = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        const int co = ln & 31;
        slab[a.w_off + (long)(tap9 * a.a.w_rows + a.a.w_r0 + ci0 + ci) * a.Cout + co0 + co] = sum;
    }
}

int launch_wgrad_s2d(const WgradArgs& a, int nslab, hipStream_t st) {
    using G = Geo<14>;
    constexpr size_t buf = (size_t)2 * W3_NRW * G::WP * 64 + 8 * W3_TP * 64 + W3_TP * sizeof(int);
    constexpr size_t lds = 2 * buf + 8 * 32 * sizeof(int);
    static_assert(lds >= 16 * 1024 * sizeof(float), "the combination step needs 64 KB");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_s2d_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            tdm_set_error("wgrad_s2d: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return 100 + (int)e;
        }
        attr_set = true;
    }
    const int nco = a.Cout / 32;
    hipLaunchKernelGGL(wgrad_s2d_kernel, dim3(nslab, a.nci * nco), dim3(1024), lds, st, a);
    TDM_CHECK_LAUNCH("wgrad_s2d");
    return 0;
}

template <int HW, bool SK2>
int launch_wgrad2_t(const WgradArgs& a, int nslab, hipStream_t st) {
    using G = Geo<HW>;
    constexpr size_t lds = (size_t)2 * (2 * W2Geo<HW>::NRW * G::WP * 64 + (SK2 ? 4 : 2) * W2_TP * 64 + W2_TP * sizeof(int)) + 8 * 32 * sizeof(int);   // + row tables
    static_assert(lds >= 8 * 1024 * sizeof(float), "final reduction needs 32 KB");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad2_s16_kernel<HW, SK2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            tdm_set_error("wgrad2_s16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return 100 + (int)e;
        }
        attr_set = true;
    }
    const int nco = a.Cout / 32;
    hipLaunchKernelGGL((wgrad2_s16_kernel<HW, SK2>), dim3(nslab, a.nci * nco), dim3(1024), lds, st, a);
    TDM_CHECK_LAUNCH("wgrad2_s16");
    return 0;
}

// out_s16[m][c] = split(in[m][c] + tb[b][c])   (generic entry points / tests)
__global__ __launch_bounds__(256) void to_s16_kernel(const float* __restrict__ in, const float* __restrict__ tb,
                                                     int tb_stride, float* __restrict__ out, long M, int HWpix, int C) {
    const int C4 = C >> 2;
    const long total = M * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / C4;
        const int c = (int)(i - m * C4) * 4;
        float4 v = reinterpret_cast<const float4*>(in)[i];
        if (tb != nullptr) {
            const float4 t4 = *reinterpret_cast<const float4*>(tb + (m / HWpix) * tb_stride + c);
            v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w;
        }
        tdm_store_s16_4(out, m, C, c, v);
    }
}

}  // namespace

int tdm_launch_conv_s16(const ConvArgs& a, int hw, int N, hipStream_t st) {
    if (a.s2d) {   // rb4.conv1's data gradient w.r.t. up(h3), at 14x14: source = the 28x28 output gradient, 4 taps per parity sub-image
        TDM_REQUIRE(hw == 14 && N == 64 && a.nsrc == 1 && a.src[0].taps == 4 && a.src[0].up == 0 && a.src[0].nch % CK == 0 && a.src[0].nch > 0 &&
                    (a.src[0].C % 16) == 0 && (a.src[0].c0 % 16) == 0 && a.src[0].tb == nullptr && a.src[0].wp != nullptr &&
                    (((uintptr_t)a.src[0].wp) & 15) == 0 && (a.out != nullptr || a.aux != nullptr) && a.skip_out == nullptr && a.o1_out == nullptr && a.r1_x == nullptr &&
                    a.dc_pair == nullptr && (a.rk1_d == nullptr || a.rk1_u != nullptr) && a.src[0].nch == 32,
                    "conv_s16: the space-to-depth form is built for rb4.conv1's data gradient (hw 14, N 64, one 28x28 source, fp32 output)");
        TDM_REQUIRE(a.B > 0 && (long)a.B * 784 < TDM_S16_MAX_PIXELS && (long)a.B * 784 * a.src[0].C * 4 < 2147483647L,
                    "conv_s16: batch %d out of range", a.B);
        return launch_conv_s2d(a, st);
    }
    if (a.up_phase) {   // rb4.conv1's forward: source 0 half-resolution with 16 phase taps per chunk, source 1 full resolution
        TDM_REQUIRE(hw == 28 && N == 32 && a.nsrc == 2 && a.src[0].taps == 16 && a.src[1].taps == 9 && a.src[0].up == 1 && a.src[1].up == 0 &&
                    a.skip_out != nullptr && a.skip_wp != nullptr && a.skip_bias != nullptr && a.res == nullptr && a.relu_mask_in == nullptr &&
                    a.o1_out == nullptr && a.r1_x == nullptr && a.dc_pair == nullptr,
                    "conv_s16: the phase form is built for rb4.conv1's forward (hw 28, N 32, up-sampled source + full-resolution source, fused skip)");
        for (int i = 0; i < 2; ++i) {
            TDM_REQUIRE(a.src[i].nch % CK == 0 && a.src[i].nch > 0 && (a.src[i].C % 16) == 0 && (a.src[i].c0 % 16) == 0 && a.src[i].tb == nullptr &&
                        a.src[i].wp != nullptr && (((uintptr_t)a.src[i].wp) & 15) == 0, "conv_s16(phase): source %d layout", i);
        }
        TDM_REQUIRE(a.B > 0 && (long)a.B * hw * hw < TDM_S16_MAX_PIXELS, "conv_s16: batch %d out of range", a.B);
        return launch_conv_phase(a, st);
    }
    for (int i = 0; i < a.nsrc; ++i) {
        TDM_REQUIRE(a.src[i].nch % CK == 0 && a.src[i].nch > 0, "conv_s16: source %d channel count %d", i, a.src[i].nch);
        TDM_REQUIRE(a.src[i].taps == 9 || a.src[i].taps == 1, "conv_s16: taps must be 9 or 1");
        TDM_REQUIRE((a.src[i].C % 16) == 0 && (a.src[i].c0 % 16) == 0, "conv_s16: S16 sources need 16-channel groups");
        TDM_REQUIRE(a.src[i].tb == nullptr, "conv_s16: the time bias is pre-added by the producer of an S16 tensor");
        TDM_REQUIRE(a.src[i].wp != nullptr && (((uintptr_t)a.src[i].wp) & 15) == 0, "conv_s16: packed weights missing");
    }
    TDM_REQUIRE(a.out != nullptr || a.out_s16 != nullptr || a.o1_out != nullptr || a.dc_pair != nullptr, "conv_s16: no output");
    TDM_REQUIRE(a.dc_pair == nullptr || (hw == 28 && N == 96 && a.dc_h1 != nullptr && a.rk1_d != nullptr && a.rk1_u != nullptr &&
                                         a.out == nullptr && a.out_s16 == nullptr && a.res == nullptr && a.relu_mask_in == nullptr &&
                                         ((long)a.B * 784) % 2 == 0),
                "conv_s16: the paired d cat epilogue is built for the 28x28 N = 96 data gradient alone");
    TDM_REQUIRE(a.o1_tgt == nullptr || (a.o1_out != nullptr && a.o1_deps != nullptr && a.o1_sums != nullptr &&
                                        a.r1_x == nullptr && a.relu_mask_in == nullptr && (TDM_ABLATE(a.ablate) & 16) == 0),
                "conv_s16: the fused MSE backward needs the fused output conv, a deps buffer and the partial rows (and neither "
                "the rank-1 residual nor a ReLU backward)");
    TDM_REQUIRE(a.o1_out == nullptr || (hw == 28 && N == 32 && a.skip_out == nullptr && a.o1_w != nullptr && a.o1_b != nullptr),
                "conv_s16: the fused output conv is built for the 28x28 N = 32 kernel");
    TDM_REQUIRE(a.r1_x == nullptr || (hw == 28 && N == 32 && a.skip_out == nullptr && a.res == nullptr && a.r1_w != nullptr && a.r1_b != nullptr),
                "conv_s16: the rank-1 residual is built for the 28x28 N = 32 kernel without another residual");
    // 32-bit addressing limits of the S16 kernels: staged-row offsets are built with __mul24 on the pixel index
    // (24-bit operands: B * hw * hw < 2^23) and held as SIGNED byte offsets with 0x80000000 = "padding"
    // (every source / epilogue tensor < 2^31 bytes; implied by the pixel bound for <= 64 channels, checked for the rest)
    TDM_REQUIRE(a.B > 0 && (long)a.B * hw * hw < TDM_S16_MAX_PIXELS,
                "conv_s16: batch %d out of range (B * %d * %d must stay below 2^23 pixels: 32-bit S16 addressing)", a.B, hw, hw);
    for (int i = 0; i < a.nsrc; ++i)
        TDM_REQUIRE((long)a.B * (hw >> a.src[i].up) * (hw >> a.src[i].up) * a.src[i].C * 4 < 2147483647L,
                    "conv_s16: source %d of batch %d exceeds 2^31 bytes", i, a.B);
    if (a.skip_out != nullptr) {   // fused 1x1 skip conv: built for the one geometry that uses it (rb4: 96 -> 32 @ 28x28)
        TDM_REQUIRE(hw == 28 && N == 32 && a.skip_wp != nullptr && a.skip_bias != nullptr, "conv_s16: fused skip needs hw=28, N=32");
        for (int i = 0; i < a.nsrc; ++i) TDM_REQUIRE(a.src[i].taps == 9, "conv_s16: fused skip rides on 3x3 sources");
        return launch_conv_t<28, 1, true>(a, st);
    }
    if (hw == 28 && N == 32) return launch_conv_t<28, 1, false>(a, st);
    if (hw == 28 && N == 64) return launch_conv_t<28, 2, false>(a, st);
    if (hw == 28 && N == 96) return launch_conv_t<28, 3, false>(a, st);
    if (hw == 14 && N == 32) return launch_conv_t<14, 1, false>(a, st);
    if (hw == 14 && N == 64) return launch_conv_t<14, 2, false>(a, st);
    tdm_set_error("conv_s16: unsupported geometry hw=%d N=%d", hw, N);
    return 1;
}

int tdm_launch_wgrad_s16(const WgradArgs& a, int hw, int nslab, hipStream_t st) {
    if (a.s2d) {   // rb4.conv1's up-sampled source in the parity form: activation at 14x14 (up = 0), gradient at 28x28
        TDM_REQUIRE(hw == 14 && a.a.taps == 9 && a.a.up == 0 && a.Cout % 32 == 0 && a.nci >= 1 && (a.a.C % 16) == 0 && (a.a.c0 % 16) == 0 &&
                    a.a.tb == nullptr && a.g2 == nullptr && nslab >= 1 && nslab <= TDM_UNET_MAX_SLABS,
                    "wgrad_s16: the parity form is built for rb4.conv1's up-sampled source (hw 14, 3x3, no fused 1x1)");
        TDM_REQUIRE(a.B > 0 && (long)a.B * 784 < TDM_S16_MAX_PIXELS && (long)a.B * 196 * a.a.C * 4 < 2147483647L &&
                    (long)a.B * 784 * a.Cout * 4 < 2147483647L, "wgrad_s16: batch %d out of range", a.B);
        return launch_wgrad_s2d(a, nslab, st);
    }
    TDM_REQUIRE(a.Cout % 32 == 0 && a.nci >= 1, "wgrad_s16: Cout %d / nci %d", a.Cout, a.nci);
    TDM_REQUIRE(a.a.taps == 9 || a.a.taps == 1, "wgrad_s16: taps must be 9 or 1");
    TDM_REQUIRE((a.a.C % 16) == 0 && (a.a.c0 % 16) == 0 && a.a.tb == nullptr, "wgrad_s16: S16 source layout");
    TDM_REQUIRE(nslab >= 1 && nslab <= TDM_UNET_MAX_SLABS, "wgrad_s16: nslab %d", nslab);
    TDM_REQUIRE(a.B > 0 && (long)a.B * hw * hw < TDM_S16_MAX_PIXELS,
                "wgrad_s16: batch %d out of range (B * %d * %d must stay below 2^23 pixels: 32-bit S16 addressing)", a.B, hw, hw);
    TDM_REQUIRE((long)a.B * (hw >> a.a.up) * (hw >> a.a.up) * a.a.C * 4 < 2147483647L && (long)a.B * hw * hw * a.Cout * 4 < 2147483647L,
                "wgrad_s16: a tensor of batch %d exceeds 2^31 bytes", a.B);
    TDM_REQUIRE(a.g2 == nullptr || a.a.taps == 9, "wgrad_s16: the fused 1x1 gradient rides on a 3x3 launch");
    if (hw == 28) return a.g2 ? launch_wgrad2_t<28, true>(a, nslab, st) : launch_wgrad2_t<28, false>(a, nslab, st);
    if (hw == 14) return a.g2 ? launch_wgrad2_t<14, true>(a, nslab, st) : launch_wgrad2_t<14, false>(a, nslab, st);
    tdm_set_error("wgrad_s16: unsupported hw=%d", hw);
    return 1;
}

int tdm_launch_to_s16(const float* in, const float* tb, int tb_stride, float* out, long M, int HWpix, int C,
                      hipStream_t st) {
    TDM_REQUIRE(C % 16 == 0, "to_s16: C=%d must be a multiple of 16", C);
    long g = (M * (C / 4) + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(to_s16_kernel, dim3((unsigned)g), dim3(256), 0, st, in, tb, tb_stride, out, M, HWpix, C);
    TDM_CHECK_LAUNCH("to_s16");
    return 0;
}
