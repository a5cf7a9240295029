// Softmax self-attention of nn.TransformerEncoderLayer (src/shakespeare.py:108-111; no mask,
// dropout on the probabilities in train mode), forward and backward, on the bf16 matrix cores
// with split operands ("bf16x3": x = hi + lo, hi*hi + hi*lo + lo*hi, fp32 accumulation — the
// arithmetic of the linear layers in gemm_bf16.hip, ~1e-5 relative).  attn_mfma.hip does the
// same products on v_mfma_f32_32x32x2_f32 (exact fp32, 16x the matrix-pipe time per FLOP); this
// file is the default, that one the cross-check (tdm_set_attn_mode).
//
// Same transposed formulation as attn_mfma.hip — a lane owns one query (forward, dQ) or one key
// (dK/dV) and its accumulator registers run over the other index:
//   S^T[key][query] = K Q^T      A = K rows [key][d] from LDS (ds_read_b128 along d), B = Q registers
//   O^T[d][query]   = V^T P^T    A = V^T from a token-major LDS image through ds_read_b64_tr_b16,
//                                B = the P^T accumulator registers, packed to bf16 as they are
// v_mfma_f32_32x32x16_bf16 contracts k-slot (h, i), i = 0..7, of lane (j, h) in A with the same
// slot in B.  The lane's accumulator registers r = 8 ks + i hold rows (r & 3) + 8 (r >> 2) + 4 h,
// so feeding registers 8 ks .. 8 ks + 7 straight back as the B operand of k-step ks works if the
// A operand's slot (h, i) holds exactly that row: the token-major ("tr") images are therefore
// STAGED with their rows permuted — row (key) kappa of a 32-row chunk sits at position
// 16 (kappa >> 4) + 8 ((kappa >> 2) & 1) + (kappa & 3) + 4 ((kappa >> 3) & 1) — and no lane
// exchange or LDS round trip is needed for P.  A tensor that is needed both ways (K in dQ; Q and
// dO in dK/dV) is staged as two images: rows with a 16-byte-odd pitch (conflict-free b128 reads
// along d) and the 64-byte-pitch token-major image the transposed reads want.
// A workgroup = 4 waves = 128 queries (keys) of one (batch, head); the other side is streamed
// through LDS in blocks of 64 rows; any L, head_dim in {8, 16, 32, 64}.
#include <math.h>
#include <algorithm>
#include "tdm_common.h"
#include "tdm_transformer.h"
#include "tdm_s16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#ifdef TDM_DIAG
__device__ int g_attn_abl_dev = 0;   // diagnostics: 1 = no chunk compute, 2 = no staging, 4 = no stores (tools/time_attn.py --ablate)
#define ATTN_ABL (g_attn_abl_dev)
#else
#define ATTN_ABL 0
#endif

namespace {

constexpr int RB = 64;    // streamed rows per LDS block
constexpr int QB = 128;   // owned rows per workgroup (32 per wave)

template <int HD> struct Cfg {
    static constexpr int HDP = HD < 16 ? 16 : HD;        // d padded to a whole k-step (zeros)
    static constexpr int KS = HDP / 16;                  // k-steps of a contraction over d
    static constexpr int NT = HD > 32 ? 2 : 1;           // 32-wide d tiles of the transposed outputs
    static constexpr int RP = HDP * 2 + 16;              // row-image pitch (bytes): odd number of 16-B slots
    static constexpr int ROWPL = RB * RP;                // one plane (hi or lo) of a row image
    static constexpr int TRPL = NT * RB * 64;            // one plane of a token-major image: [d block][position][32 d]
    static constexpr int ROWIMG = 2 * ROWPL, TRIMG = 2 * TRPL;
    static constexpr int TP = HD * 4 + 16;               // fp32 row pitch of a wave's epilogue tile (odd number of 16-B slots)
    static constexpr int EPI = 4 * 32 * TP;              // the four waves' epilogue tiles (reuse the staging area after the loop)
    static constexpr int PRO = 4 * 2 * 32 * RP + 512;    // the four waves' own-row images (both planes) + 32 floats each, before the loop
};

__device__ __forceinline__ bf16x8 tr_pair(const char* p0, const char* p1) {
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
    s16x8 r;
    r[0] = lo4[0]; r[1] = lo4[1]; r[2] = lo4[2]; r[3] = lo4[3];
    r[4] = hi4[0]; r[5] = hi4[1]; r[6] = hi4[2]; r[7] = hi4[3];
    return __builtin_bit_cast(bf16x8, r);
}

// position of row kappa (0..31 within its chunk) in a token-major image (see the header)
__device__ __forceinline__ int tr_pos(int kappa) {
    return (kappa & 16) + 8 * ((kappa >> 2) & 1) + (kappa & 3) + 4 * ((kappa >> 3) & 1);
}

__device__ __forceinline__ void split8(const float4 a, const float4 b, bf16x8& hi, bf16x8& lo) {
    tdm_bf16x4 h0, l0, h1, l1;
    tdm_split4(a, h0, l0);
    tdm_split4(b, h1, l1);
    hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h0[2]; hi[3] = h0[3]; hi[4] = h1[0]; hi[5] = h1[1]; hi[6] = h1[2]; hi[7] = h1[3];
    lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l0[2]; lo[3] = l0[3]; lo[4] = l1[0]; lo[5] = l1[1]; lo[6] = l1[2]; lo[7] = l1[3];
}

// stage rows [r0, r0 + 64) of a [rows][HD] fp32 slice (row stride ld) as split bf16: row image (rowimg != nullptr) and /
// or token-major image (trimg != nullptr); rows >= nrows are zeros.  Columns d >= HD of either image are never written:
// the kernels clear LDS once.
template <int HD>
__device__ __forceinline__ void stage_block(char* rowimg, char* trimg, const float* __restrict__ src, long ld, int r0,
                                            int nrows, int tid) {
    using C = Cfg<HD>;
#pragma unroll
    for (int e = tid; e < RB * (HD / 4); e += 256) {
        const int rr = e / (HD / 4), d4 = e - rr * (HD / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + rr < nrows) v = *reinterpret_cast<const float4*>(src + (long)(r0 + rr) * ld + d4 * 4);
        tdm_bf16x4 hi, lo;
        tdm_split4(v, hi, lo);
        if (rowimg != nullptr) {
            *reinterpret_cast<tdm_bf16x4*>(rowimg + rr * C::RP + d4 * 8) = hi;
            *reinterpret_cast<tdm_bf16x4*>(rowimg + C::ROWPL + rr * C::RP + d4 * 8) = lo;
        }
        if (trimg != nullptr) {
            const int pos = (rr & 32) + tr_pos(rr & 31);
            const int off = (((d4 * 4) >> 5) * RB + pos) * 64 + ((d4 * 4) & 31) * 2;
            *reinterpret_cast<tdm_bf16x4*>(trimg + off) = hi;
            *reinterpret_cast<tdm_bf16x4*>(trimg + C::TRPL + off) = lo;
        }
    }
}

// stage_block in two halves: the global loads of the NEXT 64-row block are requested into registers before the MFMA chunks of the
// current one and written to LDS (split, both images) behind them — the block's HBM / L2 latency runs under the compute instead of
// between two barriers (L = 128: two blocks per head, every second staging was a bare wait).  Same values, same LDS images.
template <int HD> struct BlockRegs { float4 v[(RB * (HD / 4) + 255) / 256]; };
template <int HD>
__device__ __forceinline__ void load_block(BlockRegs<HD>& p, const float* __restrict__ src, long ld, int r0, int nrows, int tid) {
#pragma unroll
    for (int i = 0; i < (RB * (HD / 4) + 255) / 256; ++i) {
        const int e = tid + 256 * i;
        const int rr = e / (HD / 4), d4 = e - rr * (HD / 4);
        p.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < RB * (HD / 4) && r0 + rr < nrows) p.v[i] = *reinterpret_cast<const float4*>(src + (long)(r0 + rr) * ld + d4 * 4);
    }
}
template <int HD>
__device__ __forceinline__ void write_block(char* rowimg, char* trimg, const BlockRegs<HD>& p, int tid) {
    using C = Cfg<HD>;
#pragma unroll
    for (int i = 0; i < (RB * (HD / 4) + 255) / 256; ++i) {
        const int e = tid + 256 * i;
        if (e >= RB * (HD / 4)) break;
        const int rr = e / (HD / 4), d4 = e - rr * (HD / 4);
        tdm_bf16x4 hi, lo;
        tdm_split4(p.v[i], hi, lo);
        if (rowimg != nullptr) {
            *reinterpret_cast<tdm_bf16x4*>(rowimg + rr * C::RP + d4 * 8) = hi;
            *reinterpret_cast<tdm_bf16x4*>(rowimg + C::ROWPL + rr * C::RP + d4 * 8) = lo;
        }
        if (trimg != nullptr) {
            const int pos = (rr & 32) + tr_pos(rr & 31);
            const int off = (((d4 * 4) >> 5) * RB + pos) * 64 + ((d4 * 4) & 31) * 2;
            *reinterpret_cast<tdm_bf16x4*>(trimg + off) = hi;
            *reinterpret_cast<tdm_bf16x4*>(trimg + C::TRPL + off) = lo;
        }
    }
}

// B-operand registers of one owned row: k-step ks holds d = 16 ks + 8 h .. + 7, split
template <int HD>
__device__ __forceinline__ void load_breg(bf16x8 (&hi)[Cfg<HD>::KS], bf16x8 (&lo)[Cfg<HD>::KS], const float* __restrict__ row,
                                          int h, bool valid) {
#pragma unroll
    for (int ks = 0; ks < Cfg<HD>::KS; ++ks) {
        const int d0 = 16 * ks + 8 * h;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (valid && d0 < HD) { a = *reinterpret_cast<const float4*>(row + d0); b = *reinterpret_cast<const float4*>(row + d0 + 4); }
        split8(a, b, hi[ks], lo[ks]);
    }
}

// acc[row = image row c32 + j][col = the lane's own row] = sum_d img[c32 + j][d] * reg[d]
template <int HD>
__device__ __forceinline__ f32x16 rows_dot_reg(const char* rowimg, int c32, int j, int h, const bf16x8 (&rh)[Cfg<HD>::KS],
                                               const bf16x8 (&rl)[Cfg<HD>::KS]) {
    using C = Cfg<HD>;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        const char* p = rowimg + (c32 + j) * C::RP + ks * 32 + h * 16;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(p);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(p + C::ROWPL);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, rl[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, rh[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, rh[ks], acc, 0, 0, 0);
    }
    return acc;
}

// out[t][d = 32 t + j][col] += sum_{rows of chunk c32} img[row][32 t + j] * w[row][col], w = accumulator-layout registers
template <int HD>
__device__ __forceinline__ void accum_T_times(f32x16 (&out)[Cfg<HD>::NT], const char* trimg, int c32, int lane, const f32x16& w) {
    using C = Cfg<HD>;
    const int g4 = lane >> 4, cb = g4 & 1, hh = g4 >> 1, q = (lane >> 2) & 3, pcq = lane & 3;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 wh, wl;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = w[8 * ks + i];
            const __bf16 xh = (__bf16)x;
            wh[i] = xh;
            wl[i] = (__bf16)(x - (float)xh);
        }
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            const char* p = trimg + ((t * RB) + c32 + ks * 16 + hh * 8 + q) * 64 + (cb * 16 + pcq * 4) * 2;
            const bf16x8 ah = tr_pair(p, p + 4 * 64);
            const bf16x8 al = tr_pair(p + C::TRPL, p + C::TRPL + 4 * 64);
            out[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, out[t], 0, 0, 0);
            out[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, out[t], 0, 0, 0);
            out[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, out[t], 0, 0, 0);
        }
    }
}

// write the transposed accumulator tiles as rows: dst_row[32 t + 8 q + 4 h + (0..3)] = out[t][4 q + ..] * mul
// dst_row (fp32) and / or dst16 (the S16 twin of the same [rows][ld] tensor: row m, columns col0 + d) may be null
template <int HD>
__device__ __forceinline__ void store_rows(float* __restrict__ dst_row, float* __restrict__ dst16, long m, int ld, int col0,
                                           const f32x16 (&out)[Cfg<HD>::NT], int h, float mul) {
#pragma unroll
    for (int t = 0; t < Cfg<HD>::NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int d = 32 * t + 8 * q + 4 * h;
            if (d < HD) {
                const float4 v = make_float4(out[t][4 * q] * mul, out[t][4 * q + 1] * mul, out[t][4 * q + 2] * mul,
                                             out[t][4 * q + 3] * mul);
                if (dst_row != nullptr) *reinterpret_cast<float4*>(dst_row + d) = v;
                if (dst16 != nullptr) tdm_store_s16_4(dst16, m, ld, col0 + d, v);
            }
        }
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// B-operand registers of the wave's 32 own rows (lane (j, h): row j), loaded COALESCED — HD/4 lanes per row, 16 B each —
// split, passed through a wave-private LDS row image and read back as fragments.  (load_breg's row-per-lane global loads
// touch 32 rows x 64 B per instruction: the dQ kernel's 32 such loads per lane were a third of its time.)
// DOT: also dots[row] = sum_d src[row][d] * other[row][d] (exact fp32; the D_i of the backward), other has row stride ldo.
template <int HD, bool DOT>
__device__ __forceinline__ void load_breg_lds(bf16x8 (&hi)[Cfg<HD>::KS], bf16x8 (&lo)[Cfg<HD>::KS], char* img, float* dots,
                                              const float* __restrict__ src, long ld, const float* __restrict__ other, long ldo,
                                              int nvalid, int lane) {
    using C = Cfg<HD>;
    constexpr int LPR = HD / 4, RPP = 64 / LPR, NP = 32 / RPP;
    float4 v[NP], w[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int rr = p * RPP + lane / LPR, d4 = lane % LPR;
        v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        w[p] = v[p];
        if (rr < nvalid) {
            v[p] = *reinterpret_cast<const float4*>(src + (long)rr * ld + d4 * 4);
            if (DOT) w[p] = *reinterpret_cast<const float4*>(other + (long)rr * ldo + d4 * 4);
        }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int rr = p * RPP + lane / LPR, d4 = lane % LPR;
        tdm_bf16x4 h4, l4;
        tdm_split4(v[p], h4, l4);
        *reinterpret_cast<tdm_bf16x4*>(img + rr * C::RP + d4 * 8) = h4;
        *reinterpret_cast<tdm_bf16x4*>(img + 32 * C::RP + rr * C::RP + d4 * 8) = l4;
        if (DOT) {
            float s = fmaf(v[p].x, w[p].x, fmaf(v[p].y, w[p].y, fmaf(v[p].z, w[p].z, v[p].w * w[p].w)));
#pragma unroll
            for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o);
            if (d4 == 0) dots[rr] = s;
        }
    }
    wave_lds_fence();
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        const char* p = img + j * C::RP + ks * 32 + h * 16;
        if (16 * ks + 8 * h < HD) {
            hi[ks] = *reinterpret_cast<const bf16x8*>(p);
            lo[ks] = *reinterpret_cast<const bf16x8*>(p + 32 * C::RP);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) { hi[ks][i] = (__bf16)0.f; lo[ks][i] = (__bf16)0.f; }
        }
    }
    wave_lds_fence();
}

// the transposed accumulator tiles of the wave's 32 own rows -> rows, through a wave-private LDS tile: every store
// instruction then writes whole 32-byte (fp32) / 16-byte (S16 piece) runs, HD/8 lanes per row.  (store_rows' row-per-lane
// stores — 64 rows x 16 B, or x 8 B for the S16 twin — were a quarter of the backward kernels' time.)
// dst: fp32 tensor [rows][ld] or nullptr, dst16: its S16 twin or nullptr; m0 = the wave's first row, col0 = first column.
template <int HD>
__device__ __forceinline__ void store_rows_lds(char* tile, float* __restrict__ dst, float* __restrict__ dst16, long m0, int nvalid,
                                               int ld, int col0, const f32x16 (&out)[Cfg<HD>::NT], int lane, float mul) {
    using C = Cfg<HD>;
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < C::NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int d = 32 * t + 8 * q + 4 * h;
            if (d < HD)
                *reinterpret_cast<float4*>(tile + j * C::TP + d * 4) =
                    make_float4(out[t][4 * q] * mul, out[t][4 * q + 1] * mul, out[t][4 * q + 2] * mul, out[t][4 * q + 3] * mul);
        }
    wave_lds_fence();
    constexpr int LPR = HD / 8, RPP = 64 / LPR, NP = (32 + RPP - 1) / RPP;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int rr = p * RPP + lane / LPR, d8 = lane % LPR;
        if (rr < 32 && rr < nvalid) {
            const float4 a = *reinterpret_cast<const float4*>(tile + rr * C::TP + d8 * 32);
            const float4 b = *reinterpret_cast<const float4*>(tile + rr * C::TP + d8 * 32 + 16);
            const int c = col0 + 8 * d8;
            if (dst != nullptr) {
                float* r = dst + (m0 + rr) * ld + c;
                *reinterpret_cast<float4*>(r) = a;
                *reinterpret_cast<float4*>(r + 4) = b;
            }
            if (dst16 != nullptr) {
                bf16x8 vh, vl;
                split8(a, b, vh, vl);
                char* g = reinterpret_cast<char*>(dst16 + (m0 + rr) * ld + (c & ~15)) + (c & 8) * 2;
                *reinterpret_cast<bf16x8*>(g) = vh;
                *reinterpret_cast<bf16x8*>(g + 32) = vl;
            }
        }
    }
    wave_lds_fence();
}

__device__ __forceinline__ void clear_lds(char* lds, int bytes, int tid) {
    for (int e = tid * 16; e < bytes; e += 256 * 16) *reinterpret_cast<float4*>(lds + e) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------ forward
template <int HD>
__global__ __launch_bounds__(256, 2) void attn_fwd_bf16_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                            float* __restrict__ o16, float* __restrict__ lse, int L, int D,
                                                            int H, float scale, DropArgs dr) {
    using C = Cfg<HD>;
    extern __shared__ float4 sm4[];
    char* Kr = reinterpret_cast<char*>(sm4);
    char* Vt = Kr + C::ROWIMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / H, hh = bh - b * H;
    const int qw0 = blockIdx.y * QB + wave * 32;
    const int qi = qw0 + j;
    const bool qvalid = qi < L;
    const float* base = qkv + (long)b * L * 3 * D + hh * HD;
    if (HD < 32) clear_lds(Kr, C::ROWIMG + C::TRIMG, tid);

    BlockRegs<HD> pk, pv;   // the next block of K / V rows, in flight
    load_block<HD>(pk, base + D, 3L * D, 0, L, tid);
    load_block<HD>(pv, base + 2 * D, 3L * D, 0, L, tid);
    bf16x8 qh[C::KS], ql[C::KS];
    load_breg<HD>(qh, ql, base + (long)qi * 3 * D, h, qvalid);
    f32x16 acc_o[C::NT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[t][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    for (int k0 = 0; k0 < L; k0 += RB) {
        __syncthreads();
        if (!(ATTN_ABL & 2)) {
        write_block<HD>(Kr, nullptr, pk, tid);
        write_block<HD>(nullptr, Vt, pv, tid);
        if (k0 + RB < L) {
            load_block<HD>(pk, base + D, 3L * D, k0 + RB, L, tid);
            load_block<HD>(pv, base + 2 * D, 3L * D, k0 + RB, L, tid);
        }
        }
        __syncthreads();
        if (qw0 >= L) continue;   // wave-uniform: this wave has no query rows
        const int nchunk = (ATTN_ABL & 1) ? 0 : min(RB / 32, (L - k0 + 31) / 32);
        for (int c = 0; c < nchunk; ++c) {
            f32x16 s = rows_dot_reg<HD>(Kr, c * 32, j, h, qh, ql);
            float mloc = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + c * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                s[r] = key < L ? s[r] * scale : -INFINITY;
                mloc = fmaxf(mloc, s[r]);
            }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
            const float m_new = fmaxf(m, mloc);
            const float corr = __expf(m - m_new);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __expf(s[r] - m_new);
                psum += p;
                s[r] = p;
            }
            if (dr.thr != 0u) {
                const unsigned long long rowbase = ((unsigned long long)bh * L + qi) * L + k0 + c * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    s[r] = tdm_keep(dr, rowbase + (r & 3) + 8 * (r >> 2)) ? s[r] * dr.scale : 0.f;
            }
            psum += __shfl_xor(psum, 32);
            l = l * corr + psum;
            m = m_new;
#pragma unroll
            for (int t = 0; t < C::NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[t][r] *= corr;
            accum_T_times<HD>(acc_o, Vt, c * 32, lane, s);
        }
    }
    __syncthreads();   // every wave is done with the staged images: their space holds the epilogue tiles now
    if (qw0 < L && !(ATTN_ABL & 4)) {
        store_rows_lds<HD>(Kr + wave * 32 * C::TP, o, o16, (long)b * L + qw0, min(32, L - qw0), D, hh * HD, acc_o, lane, 1.f / l);
        if (qvalid && h == 0) lse[(long)bh * L + qi] = m + logf(l);
    }
}

// ------------------------------------------------------------------ backward, dQ (+ D_i = dO_i . O_i)
template <int HD>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_bf16_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                               const float* __restrict__ lse, const float* __restrict__ dO,
                                                               float* __restrict__ dqkv, float* __restrict__ dqkv16,
                                                               float* __restrict__ Dvec, int L, int D, int H, float scale,
                                                               DropArgs dr) {
    using C = Cfg<HD>;
    extern __shared__ float4 sm4[];
    char* Kr = reinterpret_cast<char*>(sm4);
    char* Kt = Kr + C::ROWIMG;
    char* Vr = Kt + C::TRIMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / H, hh = bh - b * H;
    const int qw0 = blockIdx.y * QB + wave * 32;
    const int qi = qw0 + j;
    const bool qvalid = qi < L;
    const float* base = qkv + (long)b * L * 3 * D + hh * HD;
    BlockRegs<HD> pk, pv;   // the next block of K / V rows, in flight (requested before the own-row loads: they overlap)
    load_block<HD>(pk, base + D, 3L * D, 0, L, tid);
    load_block<HD>(pv, base + 2 * D, 3L * D, 0, L, tid);
    bf16x8 qh[C::KS], ql[C::KS], gh[C::KS], gl[C::KS];
    float Di = 0.f;   // D_i = dO_i . O_i, exact fp32 as in the fp32 kernels: it multiplies every probability of the row
    {
        char* img = Kr + wave * (2 * 32 * C::RP);
        float* dots = reinterpret_cast<float*>(Kr + 4 * 2 * 32 * C::RP) + wave * 32;
        const int nv = max(0, min(32, L - qw0));
        load_breg_lds<HD, false>(qh, ql, img, nullptr, base + (long)qw0 * 3 * D, 3L * D, nullptr, 0, nv, lane);
        const long r0 = ((long)b * L + qw0) * D + hh * HD;
        load_breg_lds<HD, true>(gh, gl, img, dots, dO + r0, (long)D, o + r0, (long)D, nv, lane);
        if (qvalid) Di = dots[j];
    }
    __syncthreads();
    if (HD < 32) clear_lds(Kr, 2 * C::ROWIMG + C::TRIMG, tid);
    const float lse_i = qvalid ? lse[(long)bh * L + qi] : 0.f;
    f32x16 acc_dq[C::NT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_dq[t][r] = 0.f;

    for (int k0 = 0; k0 < L; k0 += RB) {
        __syncthreads();
        if (!(ATTN_ABL & 2)) {
        write_block<HD>(Kr, Kt, pk, tid);
        write_block<HD>(Vr, nullptr, pv, tid);
        if (k0 + RB < L) {
            load_block<HD>(pk, base + D, 3L * D, k0 + RB, L, tid);
            load_block<HD>(pv, base + 2 * D, 3L * D, k0 + RB, L, tid);
        }
        }
        __syncthreads();
        if (qw0 >= L) continue;
        const int nchunk = (ATTN_ABL & 1) ? 0 : min(RB / 32, (L - k0 + 31) / 32);
        for (int c = 0; c < nchunk; ++c) {
            f32x16 s = rows_dot_reg<HD>(Kr, c * 32, j, h, qh, ql);
            const f32x16 dp = rows_dot_reg<HD>(Vr, c * 32, j, h, gh, gl);
            const unsigned long long rowbase = ((unsigned long long)bh * L + qi) * L + k0 + c * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + c * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float p = key < L ? __expf(s[r] * scale - lse_i) : 0.f;
                float dpv = dp[r];
                if (dr.thr != 0u) dpv = tdm_keep(dr, rowbase + (r & 3) + 8 * (r >> 2)) ? dpv * dr.scale : 0.f;
                s[r] = p * (dpv - Di);
            }
            accum_T_times<HD>(acc_dq, Kt, c * 32, lane, s);
        }
    }
    __syncthreads();
    if (qw0 < L && !(ATTN_ABL & 4)) {
        store_rows_lds<HD>(Kr + wave * 32 * C::TP, dqkv, dqkv16, (long)b * L + qw0, min(32, L - qw0), 3 * D, hh * HD, acc_dq, lane, scale);
        if (qvalid && h == 0) Dvec[(long)bh * L + qi] = Di;
    }
}

// ------------------------------------------------------------------ backward, dK and dV
template <int HD>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_bf16_kernel(const float* __restrict__ qkv, const float* __restrict__ lse,
                                                                const float* __restrict__ dO, const float* __restrict__ Dvec,
                                                                float* __restrict__ dqkv, float* __restrict__ dqkv16, int L,
                                                                int D, int H, float scale, DropArgs dr) {
    using C = Cfg<HD>;
    extern __shared__ float4 sm4[];
    char* Qr = reinterpret_cast<char*>(sm4);
    char* Qt = Qr + C::ROWIMG;
    char* Gr = Qt + C::TRIMG;
    char* Gt = Gr + C::ROWIMG;
    float* Ls = reinterpret_cast<float*>(Gt + C::TRIMG);   // [64] lse (+inf beyond L: exp(s - inf) = 0)
    float* Ds = Ls + RB;                                   // [64] D_i
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / H, hh = bh - b * H;
    const int kw0 = blockIdx.y * QB + wave * 32;
    const int kj = kw0 + j;
    const bool kvalid = kj < L;
    const float* base = qkv + (long)b * L * 3 * D + hh * HD;
    bf16x8 kh[C::KS], kl[C::KS], vh[C::KS], vl[C::KS];
    {
        char* img = Qr + wave * (2 * 32 * C::RP);
        const int nv = max(0, min(32, L - kw0));
        load_breg_lds<HD, false>(kh, kl, img, nullptr, base + (long)kw0 * 3 * D + D, 3L * D, nullptr, 0, nv, lane);
        load_breg_lds<HD, false>(vh, vl, img, nullptr, base + (long)kw0 * 3 * D + 2 * D, 3L * D, nullptr, 0, nv, lane);
    }
    __syncthreads();
    if (HD < 32) clear_lds(Qr, 2 * C::ROWIMG + 2 * C::TRIMG, tid);
    f32x16 acc_dk[C::NT], acc_dv[C::NT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_dk[t][r] = 0.f; acc_dv[t][r] = 0.f; }

    for (int i0 = 0; i0 < L; i0 += RB) {
        __syncthreads();
        if (!(ATTN_ABL & 2)) {
        stage_block<HD>(Qr, Qt, base, 3L * D, i0, L, tid);
        stage_block<HD>(Gr, Gt, dO + (long)b * L * D + hh * HD, (long)D, i0, L, tid);
        }
        if (tid < RB) {
            const int ig = i0 + tid;
            Ls[tid] = ig < L ? lse[(long)bh * L + ig] : INFINITY;
            Ds[tid] = ig < L ? Dvec[(long)bh * L + ig] : 0.f;
        }
        __syncthreads();
        if (kw0 >= L) continue;
        const int nchunk = (ATTN_ABL & 1) ? 0 : min(RB / 32, (L - i0 + 31) / 32);
        for (int c = 0; c < nchunk; ++c) {
            f32x16 s = rows_dot_reg<HD>(Qr, c * 32, j, h, kh, kl);     // S[query][key]
            f32x16 dp = rows_dot_reg<HD>(Gr, c * 32, j, h, vh, vl);    // dP[query][key]
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 l4 = *reinterpret_cast<const float4*>(Ls + c * 32 + 8 * q + 4 * h);
                const float4 d4 = *reinterpret_cast<const float4*>(Ds + c * 32 + 8 * q + 4 * h);
                const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int r = 4 * q + jj;
                    const float p = __expf(s[r] * scale - lq[jj]);
                    float pd = p, dpv = dp[r];
                    if (dr.thr != 0u) {
                        const int query = i0 + c * 32 + 8 * q + 4 * h + jj;
                        const bool keep = tdm_keep(dr, ((unsigned long long)bh * L + query) * L + kj);
                        pd = keep ? p * dr.scale : 0.f;
                        dpv = keep ? dpv * dr.scale : 0.f;
                    }
                    s[r] = pd;
                    dp[r] = p * (dpv - dq[jj]);
                }
            }
            accum_T_times<HD>(acc_dv, Gt, c * 32, lane, s);
            accum_T_times<HD>(acc_dk, Qt, c * 32, lane, dp);
        }
    }
    __syncthreads();
    if (kw0 < L && !(ATTN_ABL & 4)) {
        char* tile = Qr + wave * 32 * C::TP;
        const int nv = min(32, L - kw0);
        store_rows_lds<HD>(tile, dqkv, dqkv16, (long)b * L + kw0, nv, 3 * D, D + hh * HD, acc_dk, lane, scale);
        store_rows_lds<HD>(tile, dqkv, dqkv16, (long)b * L + kw0, nv, 3 * D, 2 * D + hh * HD, acc_dv, lane, 1.f);
    }
}

template <int HD>
int attn_bf16_launch(int which, const float* qkv, const float* o, const float* lse, const float* dO, float* out, float* out16,
                     float* aux, long B, int L, int D, int H, DropArgs dr, hipStream_t st) {
    using C = Cfg<HD>;
    const float scale = 1.0f / sqrtf((float)HD);
    dim3 grid((unsigned)(B * H), (L + QB - 1) / QB);
    constexpr size_t lds_f = std::max(C::ROWIMG + C::TRIMG, C::EPI),
                     lds_q = std::max(2 * C::ROWIMG + C::TRIMG, std::max(C::EPI, C::PRO)),
                     lds_kv = std::max(2 * C::ROWIMG + 2 * C::TRIMG + 2 * RB * (int)sizeof(float), std::max(C::EPI, C::PRO));
    static bool attr_set = false;
    if (!attr_set) {   // head_dim 64: the dK/dV kernel stages ~69 KB
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_bf16_kernel<HD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_bf16_kernel<HD>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_bf16_kernel<HD>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
        if (e != hipSuccess) {
            tdm_set_error("attention (bf16): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return 100 + (int)e;
        }
        attr_set = true;
    }
    if (which == 0) {
        hipLaunchKernelGGL((attn_fwd_bf16_kernel<HD>), grid, dim3(256), lds_f, st, qkv, out, out16, aux, L, D, H, scale, dr);
        TDM_CHECK_LAUNCH("attn_fwd_bf16");
    } else if (which == 1) {
        hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<HD>), grid, dim3(256), lds_q, st, qkv, o, lse, dO, out, out16, aux, L, D, H, scale, dr);
        TDM_CHECK_LAUNCH("attn_bwd_dq_bf16");
    } else {
        hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<HD>), grid, dim3(256), lds_kv, st, qkv, lse, dO, aux, out, out16, L, D, H, scale, dr);
        TDM_CHECK_LAUNCH("attn_bwd_dkv_bf16");
    }
    return 0;
}

}  // namespace

// diagnostics (effective in -DTDM_DIAG builds only)
extern "C" int tdm_attn_set_ablate(int bits) {
#ifdef TDM_DIAG
    return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_abl_dev), &bits, sizeof(int)) == hipSuccess ? 0 : 1;
#else
    (void)bits;
    return 0;
#endif
}

// which: 0 forward (out = O, aux = lse), 1 dQ (out = dqkv, aux = D written), 2 dK/dV (out = dqkv, aux = D read);
// out16 != nullptr: the S16 twin of `out` is written too (forward: `out` stays required; backward: `out` may be nullptr)
int tdm_launch_attn_bf16(int which, int hd, const float* qkv, const float* o, const float* lse, const float* dO, float* out,
                         float* out16, float* aux, long B, int L, int D, int H, DropArgs dr, hipStream_t st) {
    TDM_REQUIRE((D % 4) == 0 && (((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)out16) & 15) == 0, "attention: 16-byte alignment");
    TDM_REQUIRE(out != nullptr || (which != 0 && out16 != nullptr), "attention: no output");
    TDM_REQUIRE(out16 == nullptr || (D % 16) == 0, "attention: an S16 output needs D %% 16 == 0");
    switch (hd) {
        case 8: return attn_bf16_launch<8>(which, qkv, o, lse, dO, out, out16, aux, B, L, D, H, dr, st);
        case 16: return attn_bf16_launch<16>(which, qkv, o, lse, dO, out, out16, aux, B, L, D, H, dr, st);
        case 32: return attn_bf16_launch<32>(which, qkv, o, lse, dO, out, out16, aux, B, L, D, H, dr, st);
        case 64: return attn_bf16_launch<64>(which, qkv, o, lse, dO, out, out16, aux, B, L, D, H, dr, st);
    }
    tdm_set_error("attention: head_dim %d not supported (8, 16, 32, 64)", hd);
    return 1;
}
