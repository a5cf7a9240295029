// Fused two-GEMM chain of the transformer denoiser's feed-forward block (src/shakespeare.py:108-111:
// nn.TransformerEncoderLayer's linear1 -> ReLU -> dropout -> linear2, 256 -> 2048 -> 256) and of its data gradient,
// with the 2048-wide hidden tile living in REGISTERS only:
//
//     Z^T[f][tok] = sum_k Wa[f][k] X[tok][k]          (K = 256; f walks the hidden width in blocks of 32)
//     P^T         = mid(Z^T)                           forward: + bias, ReLU, dropout;  backward: ReLU / dropout gate
//     Y^T[o][tok] += sum_f Wb[o][f] P^T[f][tok]        (o = 0..255)
//
// One workgroup = 128 tokens = 8 waves x 16 tokens (two waves per SIMD), v_mfma_f32_16x16x32_bf16:
//   * the wave's X fragments (16 tokens x 256 k, hi and lo bf16 = 64 registers) are loaded once and stay: they are the B
//     operand of every first-product MFMA — no LDS traffic for the activation at all;
//   * the first product's two 16 x 16 accumulators of a hidden block (lane = token, registers = hidden units) ARE the B
//     operand of the second product (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's
//     operand"): it sums over the accumulators' ROW index, so the hidden activation never goes through LDS, let alone
//     HBM.  The k order that layout imposes is absorbed by the row -> hidden-unit assignment of the FIRST product's A
//     lanes (accumulator 0 row q = unit 8 (q >> 2) + (q & 3), accumulator 1 the same + 4): lane group g then holds the 8
//     consecutive units 8 g .. 8 g + 7 of its token — the second product's A operand is the natural 16-byte piece of Wb,
//     and the split (hi / lo) fragments are exactly the 16-byte pieces of the S16 row, stored as they are when the hidden
//     tensor is wanted for the weight gradients (through a wave-private LDS tile, so that a store covers whole 128-byte
//     lines instead of 32-byte pieces of 16 rows);
//   * the weights stream through a 4-slot LDS ring (32 KB per slot: 32 hidden rows x 1 KB of Wa, or 256 rows x 128 B of
//     Wb) filled by `buffer_load ... lds` three items ahead, counted vmcnt waits, one barrier per item; the 16-byte pieces
//     of a row are XOR-swizzled on the SOURCE address, and the swizzle is folded into per-lane base addresses so that every
//     fragment read is base + immediate (conflict-free ds_read_b128, no address arithmetic in the loop);
//   * the first product runs one hidden block AHEAD of the second, so the mid-op's vector work (ReLU, dropout hash, split)
//     sits beside MFMAs of the same wave, and the SIMD's other wave fills what is left.
// (A first version ran ONE wave per SIMD on 32x32x16 with 32 tokens per wave and the whole register file: correct, but
//  issue-bound — ~450 vector instructions and 16 LDS-DMA issues per block next to 96 MFMAs in one in-order stream, 221 us
//  forward at 32,768 tokens against 256 us for the two unfused launches, and its row-per-lane hidden stores cost another
//  70 us.)
// Arithmetic = the unfused kernels': bf16x3 (hi*lo + lo*hi + hi*hi per step, fp32 accumulate) or plain bf16.
#include "tdm_common.h"
#include "tdm_transformer.h"
#include "tdm_s16.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace tdm_chain {

constexpr int DK = 256;          // K of the first product = rows of Wb = width of Y
constexpr int TOK = 128;         // tokens per workgroup
constexpr int WAVES = 8, WTOK = 16;
constexpr int SLOT = 32768;      // one ring item
constexpr int NSLOT = 4;
constexpr int TILE_OFF = NSLOT * SLOT;        // wave-private store-transpose tiles: 16 tokens x (128 + 16) bytes
constexpr int TROW = 144, TILE = WTOK * TROW;
constexpr int BIAS_OFF = TILE_OFF + WAVES * TILE;   // bias_a copy (F floats)
constexpr int DPW = 4;           // DMA wave-instructions per wave and item (32 x 1 KB / 8 waves)

#define TDM_LDS(p) ((__attribute__((address_space(3))) void*)(p))
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct ChainArgs {
    const float* X16;      // [M][256] S16
    const float* Wa16;     // [F][256] S16
    const float* Wb16;     // [256][F] S16
    const float* bias_a;   // [F] or nullptr
    const float* bias_b;   // [256] or nullptr
    float* Y;              // [M][256] fp32
    float* mid16;          // [M][F] S16 (the hidden activation / its gradient) or nullptr
    unsigned* mask;        // [ceil(M/16)][ceil(F/128)][64] sign-mask words: MODE 1 writes, MODE 2 reads
    int M, F;
    float gate_scale;
    DropArgs drop_mid, drop_out;
    // hidden-range segments (few token tiles — a small batch — would leave most CUs idle): workgroup (tile, seg) runs the hidden
    // blocks [seg, seg + 1) * (F / 32) / nseg and writes its RAW partial Y to ypart[seg][M][256]; ffn_combine_kernel sums them
    // and applies bias_b / the output dropout.  nseg == 1: the kernel finishes Y itself.  (F / 32) % (4 nseg) == 0.
    int nseg;
    float* ypart;
    int ablate;            // timing diagnostics (tools/time_ffn.py --ablate; results are wrong when set): 1 no DMA after the prologue,
                           // 2 no first-product MFMAs, 4 no second-product MFMAs, 8 no mid-op arithmetic, 16 no barriers, 32 no stores
};

__device__ __forceinline__ bf16x8 pack8(const float* v) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (__bf16)v[e];
    return r;
}

// MODE 0: forward, nothing saved (inference, no dropout); 1: forward, hidden S16 + sign masks saved (training); 2: data gradient
// v ^ K computed where it is used (volatile: never hoisted into a live range across the main loop)
template <int K>
__device__ __forceinline__ int xor_now(int v) {
    int r;
    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r) : "n"(K), "v"(v));
    return r;
}

template <int NPROD, int MODE>
__global__ __launch_bounds__(512, 2) void ffn_chain_kernel(ChainArgs a) {
    extern __shared__ float4 chain_smem4[];
    char* const lds = reinterpret_cast<char*>(chain_smem4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int tile_id = (int)blockIdx.x / a.nseg, seg = (int)blockIdx.x % a.nseg;
    const int tblk = tile_id * WAVES + wave;              // the wave's 16-token block
    const int tok = tblk * WTOK + c;
    const bool tok_ok = tok < a.M;
    const bool blk_ok = tblk * WTOK < a.M;                // wave-uniform: the block has at least one token
    // this workgroup's hidden blocks: fb0 .. fb0 + NFB - 1 of the F / 32 (fb0 a multiple of 4: whole sign-mask words)
    const int F = a.F, NFB_all = F >> 5, NFB = NFB_all / a.nseg, fb0 = seg * NFB, total = 2 * NFB;
    const int NMW = (NFB_all + 3) >> 2;                   // sign-mask words per token block (all segments)
    const int NMWs = (NFB + 3) >> 2, mw0 = fb0 >> 2;      // ... of this segment, and its first word
#ifdef TDM_DIAG
    int abl = a.ablate;            // (diagnostic builds only: TDM_BUILD_DEFINES=-DTDM_DIAG python -m tinydiffusionmodels_amd.build)
    asm volatile("" : "+s"(abl));
#else
    constexpr int abl = 0;
#endif
    constexpr int NST = MODE == 0 ? 0 : 2;                // stores per hidden block and wave counted in vmcnt (the sign-mask
                                                          // word every fourth block is NOT counted: waits are then stricter)
    // dropout keys with the step's salt folded in ONCE (tdm_keep would re-read the salt word from memory per element: a
    // global load + vmcnt(0) inside the loop drains the ring)
    DropArgs dmid = a.drop_mid, dout = a.drop_out;
    if (dmid.thr != 0u && dmid.salt != nullptr) { dmid.key = tdm_salted_key(dmid.key, *dmid.salt); }
    if (dout.thr != 0u && dout.salt != nullptr) { dout.key = tdm_salted_key(dout.key, *dout.salt); }
    dmid.salt = nullptr; dout.salt = nullptr;
    if (dmid.thr == 0u) dmid.scale = 1.f;
    unsigned kmid = dmid.key, kout = dout.key;
    asm volatile("" : "+s"(kmid), "+s"(kout));   // pinned: not recomputed per use
    dmid.key = kmid; dout.key = kout;

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X16), 0, (int)((long)a.M * DK * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wa16), 0, F * DK * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wb16), 0, DK * F * 4, 0x00020000);

    // ---- the wave's activation fragments: lane (token c, group g) holds X[tok][32 ks + 8 g .. + 7], hi and lo
    bf16x8 xh[8], xl[NPROD == 3 ? 8 : 1];
    {
        const int xo = tok_ok ? tok * (DK * 4) + (g >> 1) * 64 + (g & 1) * 16 : (int)0x80000000;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            xh[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsX, xo, ks * 128, 0));
            if constexpr (NPROD == 3) xl[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsX, xo, ks * 128 + 32, 0));
        }
    }
    // MODE 2: the wave's sign masks, one byte per lane and hidden block, four blocks per word: all NMW (<= 16) words up front;
    // the loop consumes word 0 and shifts the array down every fourth block (static indices only: a runtime-indexed
    // register array would live in scratch)
    unsigned mw[MODE == 2 ? 16 : 1];
    if constexpr (MODE == 2) {
        const unsigned* mp = a.mask + ((long)tblk * NMW + mw0) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 16; ++q) mw[q] = (blk_ok && q < NMWs) ? mp[q * 64] : 0u;
    }
    // forward: bias_a to LDS (read 8 values per hidden block and lane)
    if constexpr (MODE != 2) {
        float* bl = reinterpret_cast<float*>(lds + BIAS_OFF);
        for (int i = tid; i < F; i += 512) bl[i] = a.bias_a != nullptr ? a.bias_a[i] : 0.f;
    }

    // ---- DMA plan.  Wa item: one 1 KB row per wave-instruction, physical piece q of row R holds logical piece q ^ fA(R) (low 4
    // bits), fA(R) = 4 (R >> 3) + (R & 3).  Wb item: 8 rows of 128 B per wave-instruction, physical piece q of row R holds
    // logical piece q ^ ((R >> 1) & 7).
    // ONE address register per stream: the other instructions' offsets are derived at the issue point (row R + i of the Wa item
    // differs by i KB — a scalar — and by i in the low two bits of the XOR key; 8-row group i of the Wb item by 8 i rows — a
    // scalar — and by 4 (i & 1) in the key), so six registers stay free for the accumulators
    static_assert(DPW == 4 && DK == 256, "derived DMA offsets assume 4 instructions per wave and 1 KB Wa rows");
    const int voA0 = (wave * DPW) * (DK * 4) + ((lane ^ ((wave >> 1) << 2)) << 4);
    const int voB0 = (wave * DPW * 8 + (lane >> 3)) * (F * 4) + (((lane & 7) ^ (lane >> 4)) << 4);
    // Ring items in CONSUMPTION order (the first product runs one hidden block ahead of the second): q = 0: Wa(0);
    // q = 2 fb + 1: Wa(fb + 1); q = 2 fb + 2: Wb(fb); the last item, q = 2 NFB - 1, is Wb(NFB - 1).  Item q lives in slot q & 3
    // and is requested three items ahead.
    auto issue = [&](int q) {
        if ((abl & 1) && q > 3) return;
        char* const dst = lds + (q & (NSLOT - 1)) * SLOT + wave * (DPW * 1024);
        const int fbq = (q - 1) >> 1;
        const bool isA = q == 0 || (((q - 1) & 1) == 0 && fbq + 1 < NFB);
        if (isA) {
            const int fa = fb0 + (q == 0 ? 0 : fbq + 1);
#pragma unroll
            for (int i = 0; i < DPW; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, TDM_LDS(dst + i * 1024), 16, i == 0 ? voA0 : (i == 1 ? xor_now<16>(voA0) : i == 2 ? xor_now<32>(voA0) : xor_now<48>(voA0)),
                                                         fa * (32 * DK * 4) + i * (DK * 4), 0, 0);
        } else {
            const int fbb = fb0 + (((q - 1) & 1) == 0 ? fbq : (q >> 1) - 1);
#pragma unroll
            for (int i = 0; i < DPW; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, TDM_LDS(dst + i * 1024), 16, (i & 1) == 0 ? voB0 : xor_now<64>(voB0),
                                                         fbb * 128 + i * (8 * F * 4), 0, 0);
        }
    };

    // mid16 / mask stores through buffer descriptors (a lane without a token gets an offset past num_records and the
    // hardware drops its store: no exec-mask branch, every wave issues the same stores)
    const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(a.mid16, 0, MODE != 0 ? (int)((long)a.M * F * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(a.mask, 0, MODE == 1 ? (int)(((a.M + 15) / 16) * (long)NMW * 256) : 0, 0x00020000);
    // store instruction q in {0, 1} of a block: lane -> token 8 q + (lane & 7), 16-byte piece lane >> 3 of the block's 128 bytes
    const int st_tok = tblk * WTOK + (lane & 7);
    int mvo[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) mvo[q] = st_tok + 8 * q < a.M ? (st_tok + 8 * q) * (F * 4) + (lane >> 3) * 16 : (int)0x80000000;
    const int kvo = blk_ok ? tblk * (NMW * 256) + lane * 4 : (int)0x80000000;
    char* const tile = lds + TILE_OFF + wave * TILE;
    char* const tile_w = tile + c * TROW + (g >> 1) * 64 + (g & 1) * 16;           // this lane's hi piece; lo at + 32
    const char* const tile_r = tile + (lane & 7) * TROW + (lane >> 3) * 16;        // + 8 q rows

    // ---- fragment read addresses (byte offsets inside a slot).  First product: lane (r = c, g) reads row RA(r) = 8 (r >> 2) +
    // (r & 3) (accumulator 0; accumulator 1: + 4 rows) and logical piece 8 ks + 4 (g >> 1) + (g & 1) (+ 2: lo); the XOR with
    // fA = r moves bit 3 between the steps of a pair: even steps use base_e + 128 ks, odd steps base_o + 128 (ks - 1).
    const int RA = ((c >> 2) << 3) | (c & 3);
    const int pA = ((g >> 1) << 2) | (g & 1);              // 0, 1, 4, 5
    const int a_hi = (pA ^ (c & 7)) << 4, a_lo = ((pA | 2) ^ (c & 7)) << 4;
    const int f3 = (c >> 3) & 1;
    const int offAe = RA * 1024 + f3 * 128, offAo = RA * 1024 + (1 - f3) * 128;
    const int swB = (c >> 1) & 7;
    const int offBh = c * 128 + ((pA ^ swB) << 4), offBl = c * 128 + (((pA | 2) ^ swB) << 4);

    f32x4 accY[16];
#pragma unroll
    for (int ob = 0; ob < 16; ++ob) accY[ob] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Mid-op of the previous block's Z^T (8 values per lane: z0[0..3], z1[0..3] = hidden units 32 fb + 8 g + 0..7 of token c),
    // one element per step of the first product
    struct Mid { float v[8]; unsigned bits; };
    auto mid_elem = [&](Mid& m, const f32x4& z0, const f32x4& z1, int e, unsigned ebase, unsigned m8) {
        const float z = e < 4 ? z0[e & 3] : z1[e & 3];
        if constexpr (MODE == 1) {
            // idx < 2^32 (tdm_ffn_chain_ok), so tdm_keep's hash is hash32(idx ^ key): the same integers as tdm_dropout.h
            unsigned x = (ebase + (unsigned)e) ^ dmid.key;
            x ^= x >> 16; x *= 0x7feb352dU;
            x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
            const bool on = (z > 0.f) & (x >= dmid.thr);
            m.v[e] = on ? z * dmid.scale : 0.f;
            m.bits |= on ? (1u << e) : 0u;       // (z > 0, scale >= 1: the stored value is nonzero exactly when `on`)
        } else if constexpr (MODE == 0) {
            m.v[e] = z < 0.f ? 0.f : z;
        } else {
            m.v[e] = ((m8 >> e) & 1u) ? z * a.gate_scale : 0.f;
        }
    };
    auto bias_acc = [&](int fbn, f32x4& z0, f32x4& z1) {
        if constexpr (MODE != 2) {
            // (read as bf16x8: next to LDS-DMA in flight a float-typed LDS read makes hipcc wait vmcnt(0) — type-based
            //  alias analysis against the DMA's LDS write — and drain the ring every block; tools/isa_loopwaits.py)
            const char* bl = lds + BIAS_OFF + ((fb0 + fbn) * 32 + 8 * g) * 4;
            z0 = __builtin_bit_cast(f32x4, *reinterpret_cast<const bf16x8*>(bl));
            z1 = __builtin_bit_cast(f32x4, *reinterpret_cast<const bf16x8*>(bl + 16));
        } else {
            z0 = f32x4{0.f, 0.f, 0.f, 0.f};
            z1 = z0;
        }
    };
    // first product of hidden block fbn out of ring slot `slot`; element ks of the PREVIOUS block's mid-op rides behind step ks
    auto gemm1 = [&](int fbn, int slot, auto with_mid, Mid& m, const f32x4& pz0, const f32x4& pz1, unsigned ebase, unsigned m8,
                     f32x4& z0, f32x4& z1) {
        constexpr bool WITH_MID = decltype(with_mid)::value;
        bias_acc(fbn, z0, z1);
        const char* const se = lds + slot * SLOT + offAe;
        const char* const so = lds + slot * SLOT + offAo;
        // fragments of step ks + 1 are requested BEFORE the MFMAs of step ks (hipcc's own order issues them behind the step's
        // last MFMA: the wave then sits out an LDS round trip per step with the matrix pipe idle)
        bf16x8 f0h, f1h, f0l, f1l;
        auto frag = [&](int ks, bf16x8& r0h, bf16x8& r1h, bf16x8& r0l, bf16x8& r1l) {
            const char* const sb = ((ks & 1) ? so : se) + (ks & ~1) * 128;
            r0h = *reinterpret_cast<const bf16x8*>(sb + a_hi);
            r1h = *reinterpret_cast<const bf16x8*>(sb + a_hi + 4096);
            if constexpr (NPROD == 3) {
                r0l = *reinterpret_cast<const bf16x8*>(sb + a_lo);
                r1l = *reinterpret_cast<const bf16x8*>(sb + a_lo + 4096);
            }
        };
        frag(0, f0h, f1h, f0l, f1l);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            bf16x8 n0h = f0h, n1h = f1h, n0l = f0l, n1l = f1l;
            if (ks + 1 < 8) frag(ks + 1, n0h, n1h, n0l, n1l);
            __builtin_amdgcn_sched_barrier(0);
            if (abl & 2) { asm volatile("" :: "v"(f0h), "v"(f1h)); }
            else {
                if constexpr (NPROD == 3) {
                    z0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0h, xl[ks], z0, 0, 0, 0);
                    z1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1h, xl[ks], z1, 0, 0, 0);
                    z0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0l, xh[ks], z0, 0, 0, 0);
                    z1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1l, xh[ks], z1, 0, 0, 0);
                }
                z0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0h, xh[ks], z0, 0, 0, 0);
                z1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1h, xh[ks], z1, 0, 0, 0);
            }
            if constexpr (WITH_MID) { if (!(abl & 8)) mid_elem(m, pz0, pz1, ks, ebase, m8); }
            __builtin_amdgcn_sched_barrier(0);
            f0h = n0h; f1h = n1h; f0l = n0l; f1l = n1l;
        }
    };

    issue(0);
    if (total > 1) issue(1);
    if (total > 2) issue(2);
    wait_vm<0>();
    __syncthreads();      // (bias copy visible; everything issued so far has landed — the X fragments are needed now anyway)
    if (total > 3) issue(3);
    Mid mid;
    mid.bits = 0;
    unsigned wbits = 0;   // MODE 1: sign bits of up to four blocks
    f32x4 zc0, zc1;
    gemm1(0, 0, std::false_type{}, mid, accY[0], accY[0], 0u, 0u, zc0, zc1);

    for (int fb = 0; fb < NFB; ++fb) {
        // ===== phase A': first product of block fb + 1 (item 2 fb + 1) beside the mid-op of block fb.  On the last block the
        // product is taken over a stale slot and discarded: the phase stays one straight run of MFMAs + vector work
        const int qA = 2 * fb + 1;
        if (fb == 0 || qA + 2 >= total) wait_vm<0>();
        else wait_vm<2 * DPW + NST>();
        if (!(abl & 16)) __builtin_amdgcn_s_barrier();
        if (qA + 3 < total) issue(qA + 3);
        mid.bits = 0;
        unsigned m8 = 0;
        if constexpr (MODE == 2) m8 = (mw[0] >> ((fb & 3) * 8)) & 0xffu;
        const unsigned ebase = (unsigned)tok * (unsigned)F + (unsigned)((fb0 + fb) * 32 + 8 * g);   // flat index of the lane's first hidden unit
        f32x4 zn0, zn1;
        gemm1(fb + 1 < NFB ? fb + 1 : fb, qA & (NSLOT - 1), std::true_type{}, mid, zc0, zc1, ebase, m8, zn0, zn1);
        bf16x8 ph = pack8(mid.v), pl;
        {
            float d[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = mid.v[e] - (float)ph[e];
            pl = pack8(d);
        }
        if constexpr (MODE != 0) {
            // the split fragments are the 16-byte hi / lo pieces of the S16 row: through the wave's LDS tile, so that each of
            // the two store instructions covers 8 whole 128-byte lines (plain bf16 arithmetic: lo is stored, not multiplied)
            *reinterpret_cast<bf16x8*>(tile_w) = ph;
            *reinterpret_cast<bf16x8*>(tile_w + 32) = pl;
            const u32x4 s0 = __builtin_bit_cast(u32x4, *reinterpret_cast<const bf16x8*>(tile_r));
            const u32x4 s1 = __builtin_bit_cast(u32x4, *reinterpret_cast<const bf16x8*>(tile_r + 8 * TROW));
            if (!(abl & 32)) {
                __builtin_amdgcn_raw_buffer_store_b128(s0, rsM, mvo[0], (fb0 + fb) * 128, 0);
                __builtin_amdgcn_raw_buffer_store_b128(s1, rsM, mvo[1], (fb0 + fb) * 128, 0);
            } else asm volatile("" :: "v"(s0), "v"(s1));
            if constexpr (MODE == 1) {
                wbits |= mid.bits << ((fb & 3) * 8);
                if ((fb & 3) == 3 || fb + 1 == NFB) {
                    __builtin_amdgcn_raw_buffer_store_b32(wbits, rsK, kvo, ((fb0 + fb) >> 2) * 256, 0);
                    wbits = 0;
                }
            }
        }
        if constexpr (MODE == 2) {
            if ((fb & 3) == 3) {
#pragma unroll
                for (int q = 0; q < 15; ++q) mw[q] = mw[q + 1];
            }
        }
        // ===== phase B: Y^T += Wb(fb) . P^T   (item 2 fb + 2; the last block's is item 2 NFB - 1, already waited for above)
        const int qB = fb + 1 < NFB ? 2 * fb + 2 : total - 1;
        if (qB + 2 >= total) wait_vm<0>();
        else wait_vm<2 * DPW + 2 * NST>();
        if (!(abl & 16)) __builtin_amdgcn_s_barrier();
        if (qB + 3 < total && fb + 1 < NFB) issue(qB + 3);
        {
            const char* const sb = lds + (qB & (NSLOT - 1)) * SLOT;
            // fragments of output blocks ob .. ob + PFD, requested PFD blocks ahead
            constexpr int PFD = 2, NW = PFD + 1;
            bf16x8 wh[NW], wl[NW];
#pragma unroll
            for (int q = 0; q < PFD; ++q) {
                wh[q] = *reinterpret_cast<const bf16x8*>(sb + q * 2048 + offBh);
                if constexpr (NPROD == 3) wl[q] = *reinterpret_cast<const bf16x8*>(sb + q * 2048 + offBl);
            }
#pragma unroll
            for (int ob = 0; ob < 16; ++ob) {
                if (ob + PFD < 16) {
                    wh[(ob + PFD) % NW] = *reinterpret_cast<const bf16x8*>(sb + (ob + PFD) * 2048 + offBh);
                    if constexpr (NPROD == 3) wl[(ob + PFD) % NW] = *reinterpret_cast<const bf16x8*>(sb + (ob + PFD) * 2048 + offBl);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (abl & 4) { asm volatile("" :: "v"(wh[ob % NW])); }
                else {
                    if constexpr (NPROD == 3) {
                        accY[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ob % NW], pl, accY[ob], 0, 0, 0);
                        accY[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ob % NW], ph, accY[ob], 0, 0, 0);
                    }
                    accY[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ob % NW], ph, accY[ob], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        zc0 = zn0; zc1 = zn1;
    }

    // ---- epilogue: register r of block ob = output 16 ob + 4 g + r of the lane's token
    if (!tok_ok) return;
    if (a.nseg > 1) {   // a hidden-range segment: the raw partial sums (ffn_combine_kernel finishes Y)
        float* const prow = a.ypart + ((long)seg * a.M + tok) * DK;
#pragma unroll
        for (int ob = 0; ob < 16; ++ob)
            *reinterpret_cast<float4*>(prow + ob * 16 + 4 * g) = make_float4(accY[ob][0], accY[ob][1], accY[ob][2], accY[ob][3]);
        return;
    }
    float* const yrow = a.Y + (long)tok * DK;
#pragma unroll
    for (int ob = 0; ob < 16; ++ob) {
        const int o = ob * 16 + 4 * g;
        float4 y = make_float4(accY[ob][0], accY[ob][1], accY[ob][2], accY[ob][3]);
        if (a.bias_b != nullptr) {
            const float4 b = *reinterpret_cast<const float4*>(a.bias_b + o);
            y.x += b.x; y.y += b.y; y.z += b.z; y.w += b.w;
        }
        if (dout.thr != 0u) {
            const unsigned long long e0 = (unsigned long long)tok * (unsigned)DK + (unsigned)o;
            y.x = tdm_keep(dout, e0) ? y.x * dout.scale : 0.f;
            y.y = tdm_keep(dout, e0 + 1) ? y.y * dout.scale : 0.f;
            y.z = tdm_keep(dout, e0 + 2) ? y.z * dout.scale : 0.f;
            y.w = tdm_keep(dout, e0 + 3) ? y.w * dout.scale : 0.f;
        }
        *reinterpret_cast<float4*>(yrow + o) = y;
    }
}

// Y = dropout_out(sum over the hidden-range segments of ypart + bias_b): the epilogue of a segmented launch.  One thread per
// (token, 4 outputs); fixed summation order; the same counter hash as the kernel's own epilogue.
__global__ __launch_bounds__(256) void ffn_combine_kernel(ChainArgs a) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long tok = i >> 6;
    const int o = (int)(i & 63) * 4;
    if (tok >= a.M) return;
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < a.nseg; ++s) {
        const float4 p = *reinterpret_cast<const float4*>(a.ypart + ((long)s * a.M + tok) * DK + o);
        y.x += p.x; y.y += p.y; y.z += p.z; y.w += p.w;
    }
    if (a.bias_b != nullptr) {
        const float4 b = *reinterpret_cast<const float4*>(a.bias_b + o);
        y.x += b.x; y.y += b.y; y.z += b.z; y.w += b.w;
    }
    if (a.drop_out.thr != 0u) {
        const unsigned long long e0 = (unsigned long long)tok * (unsigned)DK + (unsigned)o;
        y.x = tdm_keep(a.drop_out, e0) ? y.x * a.drop_out.scale : 0.f;
        y.y = tdm_keep(a.drop_out, e0 + 1) ? y.y * a.drop_out.scale : 0.f;
        y.z = tdm_keep(a.drop_out, e0 + 2) ? y.z * a.drop_out.scale : 0.f;
        y.w = tdm_keep(a.drop_out, e0 + 3) ? y.w * a.drop_out.scale : 0.f;
    }
    *reinterpret_cast<float4*>(a.Y + tok * DK + o) = y;
}

template <int NPROD, int MODE>
int launch_chain(const ChainArgs& a, hipStream_t st) {
    static bool attr = false;
    const int ldsb = BIAS_OFF + (MODE == 2 ? 0 : a.F * 4);   // ring + store tiles (+ bias copy)
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_chain_kernel<NPROD, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, BIAS_OFF + 2048 * 4);
        if (e != hipSuccess) {
            tdm_set_error("ffn_chain: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return 100 + (int)e;
        }
        attr = true;
    }
    hipLaunchKernelGGL((ffn_chain_kernel<NPROD, MODE>), dim3((unsigned)((a.M + TOK - 1) / TOK) * a.nseg), dim3(512), ldsb, st, a);
    TDM_CHECK_LAUNCH("ffn_chain");
    if (a.nseg > 1) {
        hipLaunchKernelGGL(ffn_combine_kernel, dim3((unsigned)(((long)a.M * 64 + 255) / 256)), dim3(256), 0, st, a);
        TDM_CHECK_LAUNCH("ffn_combine");
    }
    return 0;
}

}  // namespace tdm_chain

static int g_chain_ablate = 0;   // diagnostics only (tdm_ffn_chain_set_ablate)

bool tdm_ffn_chain_ok(long M, int D, int F) {
    return D == tdm_chain::DK && F >= 64 && (F % 32) == 0 && F <= 2048 && M >= 1 && M * (long)(F > D ? F : D) * 4 < 2147483647L;
}

// 32-bit words of the sign-mask buffer: [ceil(M/16) token blocks][ceil(F/128) words][64 lanes], byte b of a lane's word =
// hidden block 4 word + b, bit e of the byte = hidden unit 32 block + 8 (lane >> 4) + e of token 16 tblk + (lane & 15)
int64_t tdm_ffn_chain_mask_elems(int64_t M, int F) { return ((M + 15) / 16) * (int64_t)((F / 32 + 3) / 4) * 64; }

// mode 0 / 1: forward (Y = dropout_out(dropout_mid(relu(X Wa^T + bias_a)) Wb^T + bias_b)); 1 also writes mid16 / mask
// mode 2: data gradient (mid = (X Wa^T) gated by the mask * gate_scale, written to mid16; Y = mid Wb^T)
// hidden-range segments for M tokens (1 = none): as many as fill 256 CUs with the M / 128 token tiles, at most 8, whole
// sign-mask words per segment
int tdm_ffn_chain_segments(long M, int F) {
    const long tiles = (M + tdm_chain::TOK - 1) / tdm_chain::TOK;
    int n = 1;
    while (n < 8 && tiles * (2 * n) <= 256 && ((F >> 5) % (4 * 2 * n)) == 0) n *= 2;
    return n;
}
// floats of the partial-sum buffer a segmented launch needs (0: none)
long tdm_ffn_chain_part_floats(long M, int F) {
    const int n = tdm_ffn_chain_segments(M, F);
    return n > 1 ? (long)n * M * tdm_chain::DK : 0;
}

int tdm_launch_ffn_chain(int mode, int nprod, const float* X16, const float* Wa16, const float* bias_a, const float* Wb16,
                         const float* bias_b, float* Y, float* mid16, unsigned* mask, float gate_scale, DropArgs drop_mid,
                         DropArgs drop_out, long M, int D, int F, hipStream_t st, float* ypart) {
    using namespace tdm_chain;
    TDM_REQUIRE(tdm_ffn_chain_ok(M, D, F), "ffn_chain: unsupported shape M=%ld D=%d F=%d", M, D, F);
    TDM_REQUIRE(X16 && Wa16 && Wb16 && Y, "ffn_chain: NULL pointer");
    TDM_REQUIRE(mode == 0 || (mask != nullptr && mid16 != nullptr), "ffn_chain: modes 1 and 2 need the mask and the mid16 buffers");
    TDM_REQUIRE(mode != 0 || drop_mid.thr == 0u, "ffn_chain: mode 0 has no hidden dropout (use mode 1)");
    ChainArgs a{};
    a.X16 = X16; a.Wa16 = Wa16; a.Wb16 = Wb16; a.bias_a = bias_a; a.bias_b = bias_b; a.Y = Y; a.mid16 = mid16; a.mask = mask;
    a.M = (int)M; a.F = F; a.gate_scale = gate_scale; a.drop_mid = drop_mid; a.drop_out = drop_out;
    a.ablate = g_chain_ablate;
    a.nseg = ypart != nullptr ? tdm_ffn_chain_segments(M, F) : 1;    // (without a partial-sum buffer: one segment)
    a.ypart = ypart;
    if (nprod == 3) {
        if (mode == 0) return launch_chain<3, 0>(a, st);
        if (mode == 1) return launch_chain<3, 1>(a, st);
        return launch_chain<3, 2>(a, st);
    }
    if (mode == 0) return launch_chain<1, 0>(a, st);
    if (mode == 1) return launch_chain<1, 1>(a, st);
    return launch_chain<1, 2>(a, st);
}

extern "C" {

// Unit-test / timing entry of the fused chain (tests/test_gpu_text.py, tools/time_ffn.py): operands are S16 tensors the
// caller prepared (tdm_split_s16_f32); p_drop / seed / sites as in tdm_tt_fwd_f32.
int tdm_ffn_chain_f32(int mode, int nprod, const float* x16, const float* wa16, const float* bias_a, const float* wb16,
                      const float* bias_b, float* y, float* mid16, uint32_t* mask, float gate_scale, float p_drop, uint64_t seed,
                      int site_mid, int site_out, int64_t M, int D, int F, void* stream) {
    TDM_REQUIRE(mode >= 0 && mode <= 2 && (nprod == 1 || nprod == 3), "ffn_chain: mode %d nprod %d", mode, nprod);
    TDM_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "ffn_chain: dropout probability %g outside [0, 1)", (double)p_drop);
    DropArgs dm{}, dout{};
    if (p_drop > 0.f && mode == 1) { dm = tdm_drop_site(p_drop, seed, site_mid); dout = tdm_drop_site(p_drop, seed, site_out); }
    return tdm_launch_ffn_chain(mode, nprod, x16, wa16, bias_a, wb16, bias_b, y, mid16, mask, gate_scale, dm, dout, (long)M, D, F,
                                (hipStream_t)stream, nullptr);
}

int64_t tdm_ffn_chain_mask_count(int64_t M, int F) {
    return (M < 1 || F < 32 || M > ((int64_t)1 << 40)) ? -1 : tdm_ffn_chain_mask_elems(M, F);
}

int tdm_ffn_chain_set_ablate(int bits) { g_chain_ablate = bits; return 0; }

}  // extern "C"
