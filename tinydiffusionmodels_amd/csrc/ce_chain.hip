// Rounding head of the text train step (src/shakespeare.py:87-102, :239-240: logits = Linear(D, V)(x0);
// cross_entropy(logits, token_ids)) with the (tokens x V) logits living in REGISTERS only — the chained-MFMA structure of
// ffn_chain.hip (two 16 x 16 accumulators of a first product are the B operand of a second one) applied twice:
//
//   pass A, token-stationary (one workgroup = 128 tokens, their embeddings held as MFMA fragments; the vocabulary streams
//   through the LDS ring 32 rows at a time): logits^T block = W_blk x^T + b in registers; ONLINE softmax per token (running
//   max m, sum s; the accumulators are rescaled when the max moves — flash-attention's forward with K = V = W);
//   P^T = exp(l - m) split to bf16 hi/lo feeds  dX^T += W_blk^T P^T.  The pass yields lse = m + log s, the loss and
//   dX = scale/M (acc / s - W[id]) — statistics and the data gradient from ONE recomputation of the logits;
//
//   pass B, vocabulary-stationary (one workgroup = 128 vocabulary rows held as fragments; the tokens stream through the ring
//   32 at a time with their lse and target ids): logits^T block again, g = scale/M (exp(l - lse) - [v = id]) in registers,
//   dW^T += x_blk^T g  (and db = sum over tokens of g).  Token segments (NS) balance 393 vocabulary tiles over 256 CUs.
//
// Four GEMM-shaped products (logits twice, dX, dW) like the chunked form it replaces (statistics, chunk logits, dX, dW), but
// no logits tensor, no (max, sum exp) partials, no per-chunk scratch, no split-K slabs of dW: the operands are read from L2 /
// HBM once per workgroup and everything else stays in registers.  Arithmetic: bf16x3 (hi*lo + lo*hi + hi*hi, fp32 accumulate).
#include <math.h>
#include "tdm_common.h"
#include "tdm_transformer.h"
#include "tdm_s16.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace tdm_cechain {

constexpr int DK = 256;          // embedding width = K of the logits product
constexpr int WAVES = 8, WCOL = 16, COLS = WAVES * WCOL;   // stationary columns (tokens / vocabulary rows) per wave / workgroup
constexpr int AUX_OFF = 32768;   // behind a Wa item: one 256-byte piece per wave (pass A: 32 biases; pass B: 32 lse | 32 ids)
constexpr int SLOT = AUX_OFF + WAVES * 256;
constexpr int NSLOT = 4;
constexpr int DPW = 4;           // DMA wave-instructions per wave and item (+ 1 aux piece on Wa items)

#define TDM_LDS(p) ((__attribute__((address_space(3))) void*)(p))
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct CeArgs {
    const float* C16;      // stationary rows [NC][256] S16 (pass A: x; pass B: W)
    const float* Wa16;     // streamed rows   [NR][256] S16 (pass A: W; pass B: x)
    const float* Wb16;     // [256][NRp] S16: the streamed operand transposed (pass A: W^T; pass B: x^T)
    const float* aux;      // pass A: bias [NR]; pass B: lse | id blocks [NRp / 32][64]
    int NC, NR, NRp;
    float gscale;          // grad_scale / M
    // pass A
    const int64_t* ids;    // [M]
    const float* W;        // [V][256] fp32 (the -W[id] term of dX)
    const float* tl;       // [M] target logits (exact fp32 row dots)
    float* dx;             // [M][256] or nullptr
    float* lse_id;         // out: [Mp / 32][64]
    float* rowloss;        // out: [M]
    int Mp;
    // pass A split over the vocabulary (nseg > 1; few token tiles — a small batch — would leave most CUs idle): segment s of
    // the vocabulary blocks yields the token's running maximum, exp-sum and UNNORMALISED dX numerator; ce_combine_kernel merges them
    float* pm;             // [nseg][Mp]
    float* ps;             // [nseg][Mp]
    float* pacc;           // [nseg][Mp][256]
    // pass B
    const float* bias;     // [V]
    float* dW;             // [nseg][V][256]
    float* db;             // [nseg][Vp]
    int nseg, Vp;
};

__device__ __forceinline__ bf16x8 pack8(const float* v) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (__bf16)v[e];
    return r;
}
__device__ __forceinline__ float xor16(float v) { return __shfl_xor(v, 16); }
__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32); }

// PASS 0 = A (token-stationary: online softmax, dX), 1 = B (vocabulary-stationary: dW, db)
template <int PASS>
__global__ __launch_bounds__(512, 2) void ce_chain_kernel(CeArgs a) {
    extern __shared__ float4 ce_smem4[];
    char* const lds = reinterpret_cast<char*>(ce_smem4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int tile = (int)blockIdx.x / a.nseg;
    const int seg = (int)blockIdx.x % a.nseg;
    const int col = (tile * WAVES + wave) * WCOL + c;         // this lane's stationary column (token / vocabulary row)
    const bool col_ok = col < a.NC;
    const int nblk_all = a.NRp >> 5;
    // streamed blocks of this workgroup: segment `seg` of the vocabulary blocks (pass A; one segment unless the batch is small)
    // or of the token blocks (pass B)
    const int fb0 = (int)((long)nblk_all * seg / a.nseg);
    const int NFB = (int)((long)nblk_all * (seg + 1) / a.nseg) - fb0;
    const int total = 2 * NFB;
    if (NFB <= 0) return;

    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.C16), 0, (int)((long)a.NC * DK * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wa16), 0, (int)((long)a.NR * DK * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wb16), 0, (int)((long)DK * a.NRp * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.aux), 0, PASS == 0 ? a.NR * 4 : (a.NRp >> 5) * 256, 0x00020000);

    // ---- stationary fragments: lane (column c, group g) holds C[col][32 ks + 8 g .. + 7], hi and lo
    bf16x8 xh[8], xl[8];
    {
        const int xo = col_ok ? col * (DK * 4) + (g >> 1) * 64 + (g & 1) * 16 : (int)0x80000000;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            xh[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsC, xo, ks * 128, 0));
            xl[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsC, xo, ks * 128 + 32, 0));
        }
    }
    float bias_v = 0.f;
    if constexpr (PASS == 1) bias_v = col_ok ? a.bias[col] : 0.f;

    // ---- DMA plan (ffn_chain.hip's): Wa item = 32 rows x 1 KB, piece q of row R holds logical piece q ^ fA(R); Wb item = 256 rows
    // x 128 B, piece q of row R holds logical piece q ^ ((R >> 1) & 7); + the wave's own 256-byte aux piece behind a Wa item
    int voA[DPW], voB[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int row = wave * DPW + i;
        voA[i] = row * (DK * 4) + ((lane ^ (((row >> 3) << 2) | (row & 3))) << 4);
        const int rb = (wave * DPW + i) * 8 + (lane >> 3);
        voB[i] = rb * (a.NRp * 4) + (((lane & 7) ^ ((rb >> 1) & 7)) << 4);
    }
    // items in consumption order: q = 0: Wa(0); q = 2 fb + 1: Wa(fb + 1); q = 2 fb + 2: Wb(fb); last: Wb(NFB - 1)   (fb relative to fb0)
    auto issue = [&](int q) {
        char* const slot = lds + (q & (NSLOT - 1)) * SLOT;
        char* const dst = slot + wave * (DPW * 1024);
        const int fbq = (q - 1) >> 1;
        const bool isA = q == 0 || (((q - 1) & 1) == 0 && fbq + 1 < NFB);
        if (isA) {
            const int fa = fb0 + (q == 0 ? 0 : fbq + 1);
#pragma unroll
            for (int i = 0; i < DPW; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, TDM_LDS(dst + i * 1024), 16, voA[i], fa * (32 * DK * 4), 0, 0);
            // aux piece (4 bytes per lane): pass A: bias[32 fa + lane] (lanes 32..63 unused); pass B: the 64 words of block fa
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, TDM_LDS(slot + AUX_OFF + wave * 256), 4, lane * 4, fa * (PASS == 0 ? 128 : 256), 0, 0);
        } else {
            const int fbb = fb0 + (((q - 1) & 1) == 0 ? fbq : (q >> 1) - 1);
#pragma unroll
            for (int i = 0; i < DPW; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, TDM_LDS(dst + i * 1024), 16, voB[i], fbb * 128, 0, 0);
        }
    };

    // ---- fragment read addresses (ffn_chain.hip)
    const int RA = ((c >> 2) << 3) | (c & 3);
    const int pA = ((g >> 1) << 2) | (g & 1);
    const int a_hi = (pA ^ (c & 7)) << 4, a_lo = ((pA | 2) ^ (c & 7)) << 4;
    const int f3 = (c >> 3) & 1;
    const int offAe = RA * 1024 + f3 * 128, offAo = RA * 1024 + (1 - f3) * 128;
    const int swB = (c >> 1) & 7;
    const int offBh = c * 128 + ((pA ^ swB) << 4), offBl = c * 128 + (((pA | 2) ^ swB) << 4);

    f32x4 accY[16];
#pragma unroll
    for (int ob = 0; ob < 16; ++ob) accY[ob] = f32x4{0.f, 0.f, 0.f, 0.f};

    // first product of a streamed block out of ring slot `slot`: Z^T (32 streamed rows x 16 columns); lane group g ends up with
    // rows 8 g .. 8 g + 7 (z0: + 0..3, z1: + 4..7).  Pass A starts from the block's biases (rows = vocabulary entries).
    auto gemm1 = [&](int slot, f32x4& z0, f32x4& z1) {
        const char* const sbase = lds + slot * SLOT;
        if constexpr (PASS == 0) {
            const char* bl = sbase + AUX_OFF + wave * 256 + 8 * g * 4;
            z0 = __builtin_bit_cast(f32x4, *reinterpret_cast<const bf16x8*>(bl));
            z1 = __builtin_bit_cast(f32x4, *reinterpret_cast<const bf16x8*>(bl + 16));
        } else {
            z0 = f32x4{0.f, 0.f, 0.f, 0.f};
            z1 = z0;
        }
        const char* const se = sbase + offAe;
        const char* const so = sbase + offAo;
        bf16x8 f0h, f1h, f0l, f1l;
        auto frag = [&](int ks, bf16x8& r0h, bf16x8& r1h, bf16x8& r0l, bf16x8& r1l) {
            const char* const sb = ((ks & 1) ? so : se) + (ks & ~1) * 128;
            r0h = *reinterpret_cast<const bf16x8*>(sb + a_hi);
            r1h = *reinterpret_cast<const bf16x8*>(sb + a_hi + 4096);
            r0l = *reinterpret_cast<const bf16x8*>(sb + a_lo);
            r1l = *reinterpret_cast<const bf16x8*>(sb + a_lo + 4096);
        };
        frag(0, f0h, f1h, f0l, f1l);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            bf16x8 n0h = f0h, n1h = f1h, n0l = f0l, n1l = f1l;
            if (ks + 1 < 8) frag(ks + 1, n0h, n1h, n0l, n1l);
            __builtin_amdgcn_sched_barrier(0);
            z0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0h, xl[ks], z0, 0, 0, 0);
            z1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1h, xl[ks], z1, 0, 0, 0);
            z0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0l, xh[ks], z0, 0, 0, 0);
            z1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1l, xh[ks], z1, 0, 0, 0);
            z0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0h, xh[ks], z0, 0, 0, 0);
            z1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1h, xh[ks], z1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            f0h = n0h; f1h = n1h; f0l = n0l; f1l = n1l;
        }
    };
    // pass B: lse and target ids of the 8 streamed rows (tokens) this lane group holds, out of the slot's aux piece
    auto read_aux = [&](int slot, f32x4& l0, f32x4& l1, i32x4& i0, i32x4& i1) {
        const char* ab = lds + slot * SLOT + AUX_OFF + wave * 256 + 8 * g * 4;
        l0 = __builtin_bit_cast(f32x4, *reinterpret_cast<const bf16x8*>(ab));
        l1 = __builtin_bit_cast(f32x4, *reinterpret_cast<const bf16x8*>(ab + 16));
        i0 = __builtin_bit_cast(i32x4, *reinterpret_cast<const bf16x8*>(ab + 128));
        i1 = __builtin_bit_cast(i32x4, *reinterpret_cast<const bf16x8*>(ab + 144));
    };

    issue(0);
    if (total > 1) issue(1);
    if (total > 2) issue(2);
    wait_vm<0>();
    __syncthreads();
    if (total > 3) issue(3);
    f32x4 zc0, zc1;
    f32x4 lc0{}, lc1{};
    i32x4 ic0{}, ic1{};
    gemm1(0, zc0, zc1);
    if constexpr (PASS == 1) read_aux(0, lc0, lc1, ic0, ic1);
    float m_ref = -INFINITY, ssum = 0.f, dbacc = 0.f;

    for (int fb = 0; fb < NFB; ++fb) {
        // ===== phase A': first product of block fb + 1 beside the mid-op of block fb
        const int qA = 2 * fb + 1;
        // (exact count = the two items behind qA: a Wb item (DPW pieces) and a Wa item (DPW + 1); near the tail the last item is a
        //  Wb item in a Wa position, so the counted wait stops one phase earlier)
        if (fb == 0 || qA + 3 >= total) wait_vm<0>();
        else wait_vm<2 * DPW + 1>();
        __builtin_amdgcn_s_barrier();
        if (qA + 3 < total) issue(qA + 3);
        f32x4 zn0, zn1;
        f32x4 ln0{}, ln1{};
        i32x4 in0{}, in1{};
        gemm1(qA & (NSLOT - 1), zn0, zn1);
        if constexpr (PASS == 1) read_aux(qA & (NSLOT - 1), ln0, ln1, in0, in1);
        float v[8];
        if constexpr (PASS == 0) {
            // online softmax over the vocabulary: this lane's 8 logits are entries 32 (fb0 + fb) + 8 g + 0..7 of its token
            float z[8] = {zc0[0], zc0[1], zc0[2], zc0[3], zc1[0], zc1[1], zc1[2], zc1[3]};
            if (fb0 + fb == nblk_all - 1) {   // the last block may run past V: those rows are zero weights + zero bias, not vocabulary
                const int left = a.NR - (32 * (fb0 + fb) + 8 * g);
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = e < left ? z[e] : -INFINITY;
            }
            float mx = fmaxf(fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], z[3])), fmaxf(fmaxf(z[4], z[5]), fmaxf(z[6], z[7])));
            mx = fmaxf(mx, xor16(mx));
            mx = fmaxf(mx, xor32(mx));       // the token's maximum over the block (the four lane groups share ONE reference)
            if (__builtin_amdgcn_ballot_w64(mx > m_ref) != 0ull) {   // some token of the wave moved its maximum: rescale what it holds
                const float mn = fmaxf(m_ref, mx);
                const float f = __expf(m_ref - mn);                 // (first block: exp(-inf) = 0 on zeros)
#pragma unroll
                for (int ob = 0; ob < 16; ++ob) accY[ob] *= f;
                ssum *= f;
                m_ref = mn;
            }
            float ps = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] = __expf(z[e] - m_ref); ps += v[e]; }
            ssum += ps;
        } else {
            // g = scale / M (exp(l - lse[token]) - [id[token] = v]) for the 8 tokens 32 (fb0 + fb) + 8 g + 0..7 and this lane's v
            const float z[8] = {zc0[0], zc0[1], zc0[2], zc0[3], zc1[0], zc1[1], zc1[2], zc1[3]};
            const float ls[8] = {lc0[0], lc0[1], lc0[2], lc0[3], lc1[0], lc1[1], lc1[2], lc1[3]};
            const int id8[8] = {ic0[0], ic0[1], ic0[2], ic0[3], ic1[0], ic1[1], ic1[2], ic1[3]};
            float ps = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float p = __expf(z[e] + bias_v - ls[e]);
                v[e] = (p - (id8[e] == col ? 1.f : 0.f)) * a.gscale;
                ps += v[e];
            }
            dbacc += ps;
        }
        bf16x8 ph = pack8(v), pl;
        {
            float d[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = v[e] - (float)ph[e];
            pl = pack8(d);
        }
        // ===== phase B: Y^T += Wb(fb) . P^T
        const int qB = fb + 1 < NFB ? 2 * fb + 2 : total - 1;
        if (qB + 2 >= total) wait_vm<0>();
        else wait_vm<2 * DPW + 1>();
        __builtin_amdgcn_s_barrier();
        if (qB + 3 < total && fb + 1 < NFB) issue(qB + 3);
        {
            const char* const sb = lds + (qB & (NSLOT - 1)) * SLOT;
            bf16x8 wh[3], wl[3];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                wh[q] = *reinterpret_cast<const bf16x8*>(sb + q * 2048 + offBh);
                wl[q] = *reinterpret_cast<const bf16x8*>(sb + q * 2048 + offBl);
            }
#pragma unroll
            for (int ob = 0; ob < 16; ++ob) {
                if (ob + 2 < 16) {
                    wh[(ob + 2) % 3] = *reinterpret_cast<const bf16x8*>(sb + (ob + 2) * 2048 + offBh);
                    wl[(ob + 2) % 3] = *reinterpret_cast<const bf16x8*>(sb + (ob + 2) * 2048 + offBl);
                }
                __builtin_amdgcn_sched_barrier(0);
                accY[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ob % 3], pl, accY[ob], 0, 0, 0);
                accY[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ob % 3], ph, accY[ob], 0, 0, 0);
                accY[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ob % 3], ph, accY[ob], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        zc0 = zn0; zc1 = zn1;
        if constexpr (PASS == 1) { lc0 = ln0; lc1 = ln1; ic0 = in0; ic1 = in1; }
    }

    // ---- epilogues: register r of block ob = output 16 ob + 4 g + r of the lane's column
    if constexpr (PASS == 0) {
        float st = ssum + xor16(ssum);
        st += xor32(st);
        const int tok = col;
        if (a.nseg > 1) {   // a vocabulary segment: partial statistics and the unnormalised numerator (merged by ce_combine_kernel)
            if (tok >= a.Mp) return;
            if (g == 0) { a.pm[(long)seg * a.Mp + tok] = m_ref; a.ps[(long)seg * a.Mp + tok] = st; }
            float* const prow = a.pacc + ((long)seg * a.Mp + tok) * DK;
#pragma unroll
            for (int ob = 0; ob < 16; ++ob)
                *reinterpret_cast<float4*>(prow + ob * 16 + 4 * g) = make_float4(accY[ob][0], accY[ob][1], accY[ob][2], accY[ob][3]);
            return;
        }
        const float lse = m_ref + __logf(st);
        if (tok < a.Mp && g == 0) {      // lse | id block of pass B (padding tokens: +inf / -1 -> zero gradient)
            float* blk = a.lse_id + (long)(tok >> 5) * 64 + (tok & 31);
            long id = col_ok ? a.ids[tok] : -1;
            blk[0] = col_ok ? lse : INFINITY;
            blk[32] = __int_as_float((id >= 0 && id < a.NR) ? (int)id : -1);
            if (col_ok) a.rowloss[tok] = lse - a.tl[tok];      // (tl is NaN for an id outside [0, V): the loss shows it)
        }
        if (!col_ok || a.dx == nullptr) return;
        const long id = a.ids[tok];
        const bool idok = id >= 0 && id < a.NR;
        const float inv = 1.f / st;
        float* const drow = a.dx + (long)tok * DK;
#pragma unroll
        for (int ob = 0; ob < 16; ++ob) {
            const int o = ob * 16 + 4 * g;
            float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idok) w4 = *reinterpret_cast<const float4*>(a.W + id * DK + o);
            float4 y;
            y.x = (accY[ob][0] * inv - w4.x) * a.gscale; y.y = (accY[ob][1] * inv - w4.y) * a.gscale;
            y.z = (accY[ob][2] * inv - w4.z) * a.gscale; y.w = (accY[ob][3] * inv - w4.w) * a.gscale;
            *reinterpret_cast<float4*>(drow + o) = y;
        }
    } else {
        float dbt = dbacc + xor16(dbacc);
        dbt += xor32(dbt);
        if (!col_ok) return;
        if (g == 0) a.db[(long)seg * a.Vp + col] = dbt;
        float* const wrow = a.dW + ((long)seg * a.NC + col) * DK;
#pragma unroll
        for (int ob = 0; ob < 16; ++ob)
            *reinterpret_cast<float4*>(wrow + ob * 16 + 4 * g) = make_float4(accY[ob][0], accY[ob][1], accY[ob][2], accY[ob][3]);
    }
}

// merges the vocabulary segments of a split pass A: per token m = max_s m_s, l = sum_s l_s e^(m_s - m), lse = m + log l,
// dX = scale/M (sum_s acc_s e^(m_s - m) / l - W[id]); writes pass B's lse | id blocks and the loss rows like the unsplit epilogue.
// One thread per (token, 4 columns); fixed summation order.
__global__ __launch_bounds__(256) void ce_combine_kernel(CeArgs a) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int tok = (int)(i >> 6), o = (int)(i & 63) * 4;
    if (tok >= a.Mp) return;
    const bool tok_ok = tok < a.NC;
    float m = -INFINITY;
    for (int s = 0; s < a.nseg; ++s) m = fmaxf(m, a.pm[(long)s * a.Mp + tok]);
    float l = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < a.nseg; ++s) {
        const float f = __expf(a.pm[(long)s * a.Mp + tok] - m);
        l = fmaf(a.ps[(long)s * a.Mp + tok], f, l);
        const float4 p = *reinterpret_cast<const float4*>(a.pacc + ((long)s * a.Mp + tok) * DK + o);
        acc.x = fmaf(p.x, f, acc.x); acc.y = fmaf(p.y, f, acc.y); acc.z = fmaf(p.z, f, acc.z); acc.w = fmaf(p.w, f, acc.w);
    }
    const float lse = m + __logf(l);
    long id = tok_ok ? a.ids[tok] : -1;
    const bool idok = id >= 0 && id < a.NR;
    if (o == 0) {
        float* blk = a.lse_id + (long)(tok >> 5) * 64 + (tok & 31);
        blk[0] = tok_ok ? lse : INFINITY;
        blk[32] = __int_as_float(idok ? (int)id : -1);
        if (tok_ok) a.rowloss[tok] = lse - a.tl[tok];
    }
    if (!tok_ok || a.dx == nullptr) return;
    float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idok) w4 = *reinterpret_cast<const float4*>(a.W + id * DK + o);
    const float inv = 1.f / l;
    *reinterpret_cast<float4*>(a.dx + (long)tok * DK + o) =
        make_float4((acc.x * inv - w4.x) * a.gscale, (acc.y * inv - w4.y) * a.gscale, (acc.z * inv - w4.z) * a.gscale, (acc.w * inv - w4.w) * a.gscale);
}

// out S16 [C][Rp] = in [R][C]^T, rows r >= R zero (Rp a multiple of 32, C a multiple of 32): 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_s16_pad_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C,
                                                                int Rp) {
    __shared__ float t[32][33];
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const int r = i >> 5, cc = i & 31;
        t[r][cc] = (r0 + r < R) ? in[(long)(r0 + r) * C + c0 + cc] : 0.f;
    }
    __syncthreads();
    // output row c0 + cc holds elements r0 .. r0 + 31 = two S16 groups of 16: thread -> (row cc, quad q of 8)
    const int cc = threadIdx.x >> 3, q = threadIdx.x & 7;
    const float4 v = make_float4(t[4 * q][cc], t[4 * q + 1][cc], t[4 * q + 2][cc], t[4 * q + 3][cc]);
    tdm_store_s16_4(out, (long)(c0 + cc), Rp, r0 + 4 * q, v);
}

// tl[m] = x[m] . W[ids[m]] + b[ids[m]] in exact fp32 (the target logit of the cross-entropy); NaN for an id outside [0, V)
__global__ __launch_bounds__(256) void target_logit_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                           const float* __restrict__ b, const int64_t* __restrict__ ids,
                                                           float* __restrict__ tl, long M, int V) {
    const int lane = threadIdx.x & 63;
    for (long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (long)gridDim.x * 4) {
        const long id = ids[m];
        if (id < 0 || id >= V) {
            if (lane == 0) tl[m] = __int_as_float(0x7fc00000);
            continue;
        }
        const float4 xa = reinterpret_cast<const float4*>(x + m * DK)[lane];
        const float4 wa = reinterpret_cast<const float4*>(W + id * DK)[lane];
        float s = (xa.x * wa.x + xa.y * wa.y) + (xa.z * wa.z + xa.w * wa.w);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) tl[m] = s + b[id];
    }
}

__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ rowloss, float* __restrict__ out, long M) {
    __shared__ float sh[256];
    float s = 0.f;
    for (long i = threadIdx.x; i < M; i += 256) s += rowloss[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0] / (float)M;
}

// vocabulary segments of pass A: one unless the batch has too few 128-token tiles to fill the chip (B = 32 x 128 tokens: 32
// tiles on 256 CUs ran pass A at an eighth of the machine — 3.5 of the step's 6.5 ms)
inline int pass_a_segments(long M) {
    const long tiles = (((M + 31) & ~31L) + COLS - 1) / COLS;
    const long s = 256 / (tiles < 1 ? 1 : tiles);
    return (int)(s < 1 ? 1 : (s > 8 ? 8 : s));
}
struct FusedWs { float *x16, *w16, *wT16, *xT16, *lse_id, *rowloss, *tl, *dWs, *dbs, *pm, *ps, *pacc; int Vp, Mp, nsegA; long total; };
FusedWs fused_carve(float* base, long M, int V, int nseg) {
    FusedWs w{};
    long off = 0;
    auto take = [&](long n) { float* p = base ? base + off : nullptr; off += (n + 63) & ~63L; return p; };
    w.Vp = (V + 31) & ~31;
    w.Mp = (int)((M + 31) & ~31L);
    w.x16 = take(M * DK); w.w16 = take((long)V * DK); w.wT16 = take((long)DK * w.Vp); w.xT16 = take((long)DK * w.Mp);
    w.lse_id = take((long)(w.Mp / 32) * 64); w.rowloss = take(M); w.tl = take(M);
    w.dWs = nseg > 1 ? take((long)nseg * V * DK) : nullptr;
    w.dbs = take((long)nseg * w.Vp);
    w.nsegA = pass_a_segments(M);
    if (w.nsegA > w.Vp / 32) w.nsegA = w.Vp / 32;      // every vocabulary segment owns at least one 32-row block
    if (w.nsegA > 1) { w.pm = take((long)w.nsegA * w.Mp); w.ps = take((long)w.nsegA * w.Mp); w.pacc = take((long)w.nsegA * w.Mp * DK); }
    w.total = off;
    return w;
}

}  // namespace tdm_cechain

extern "C" {

int tdm_round_fused_ok(int64_t M, int V, int D) {
    return D == tdm_cechain::DK && M >= 1 && M < (1L << 31) && V >= 32 && M * (long)D * 4 < 2147483647L && (long)V * D * 4 < 2147483647L &&
           ((M + 31) & ~31L) * (long)D * 4 < 2147483647L;
}

int64_t tdm_round_workspace_fused_floats(int64_t M, int V, int D, int nseg) {
    if (!tdm_round_fused_ok(M, V, D) || nseg < 1 || nseg > 8 || M > ((int64_t)1 << 31)) return -1;
    return tdm_cechain::fused_carve(nullptr, M, V, nseg).total;
}

// tdm_round_ce_loss_grad_f32's contract (loss = mean CE; dx, dW, db = gradients of grad_scale * loss) with the logits in
// registers only (two chained-MFMA passes, see the file header).  D must be 256.  nseg: token segments of the weight-gradient
// pass (1..8; 3 balances 393 vocabulary tiles of V = 50,257 over 256 CUs).  ws: tdm_round_workspace_fused_floats(M, V, D, nseg).
int tdm_round_ce_loss_grad_fused_f32(const float* x, const float* W, const float* b, const int64_t* ids, float grad_scale,
                                     float* loss_out, float* dx, float* dW, float* db, float* ws, int64_t M, int V, int D, int nseg,
                                     void* stream) {
    using namespace tdm_cechain;
    TDM_REQUIRE(tdm_round_fused_ok(M, V, D), "round_ce_fused: unsupported shape M=%lld V=%d D=%d (D must be 256)", (long long)M, V, D);
    TDM_REQUIRE(x && W && b && ids && loss_out && dW && db && ws, "round_ce_fused: NULL pointer");
    TDM_REQUIRE(nseg >= 1 && nseg <= 8, "round_ce_fused: %d token segments (1..8)", nseg);
    hipStream_t st = (hipStream_t)stream;
    const FusedWs w = fused_carve(ws, M, V, nseg);     // (carved for the caller's nseg; fewer segments may be used)
    if (nseg > w.Mp / 32) nseg = w.Mp / 32;            // every segment of the weight-gradient pass owns at least one token block
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ce_chain_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ce_chain_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT);
        if (e != hipSuccess) {
            tdm_set_error("round_ce_fused: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return 100 + (int)e;
        }
        attr = true;
    }
    // operands: S16 copies of x and W, and their transposes (zero-padded to whole 32-row blocks)
    TDM_TRY(tdm_launch_split_s16(x, w.x16, (long)M * DK, st));
    TDM_TRY(tdm_launch_split_s16(W, w.w16, (long)V * DK, st));
    hipLaunchKernelGGL(transpose_s16_pad_kernel, dim3(w.Vp / 32, DK / 32), dim3(256), 0, st, W, w.wT16, V, DK, w.Vp);
    TDM_CHECK_LAUNCH("transpose_s16_pad(W)");
    hipLaunchKernelGGL(transpose_s16_pad_kernel, dim3(w.Mp / 32, DK / 32), dim3(256), 0, st, x, w.xT16, (int)M, DK, w.Mp);
    TDM_CHECK_LAUNCH("transpose_s16_pad(x)");
    hipLaunchKernelGGL(target_logit_kernel, dim3((unsigned)((M + 3) / 4 < 4096 ? (M + 3) / 4 : 4096)), dim3(256), 0, st, x, W, b, ids,
                       w.tl, (long)M, V);
    TDM_CHECK_LAUNCH("target_logit");
    const float gscale = grad_scale / (float)M;
    {   // pass A: lse, loss rows, dX
        CeArgs a{};
        a.C16 = w.x16; a.Wa16 = w.w16; a.Wb16 = w.wT16; a.aux = b; a.NC = (int)M; a.NR = V; a.NRp = w.Vp; a.gscale = gscale;
        a.ids = ids; a.W = W; a.tl = w.tl; a.dx = dx; a.lse_id = w.lse_id; a.rowloss = w.rowloss; a.Mp = w.Mp;
        a.nseg = w.nsegA; a.pm = w.pm; a.ps = w.ps; a.pacc = w.pacc;
        const unsigned tiles = (unsigned)(w.Mp / COLS + ((w.Mp % COLS) ? 1 : 0));
        hipLaunchKernelGGL((ce_chain_kernel<0>), dim3(tiles * w.nsegA), dim3(512), NSLOT * SLOT, st, a);
        TDM_CHECK_LAUNCH("ce_chain(A)");
        if (w.nsegA > 1) {
            hipLaunchKernelGGL(ce_combine_kernel, dim3((unsigned)(((long)w.Mp * 64 + 255) / 256)), dim3(256), 0, st, a);
            TDM_CHECK_LAUNCH("ce_combine");
        }
    }
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, w.rowloss, loss_out, (long)M);
    TDM_CHECK_LAUNCH("ce_mean");
    {   // pass B: dW, db
        CeArgs a{};
        a.C16 = w.w16; a.Wa16 = w.x16; a.Wb16 = w.xT16; a.aux = w.lse_id; a.NC = V; a.NR = (int)M; a.NRp = w.Mp; a.gscale = gscale;
        a.bias = b; a.dW = nseg > 1 ? w.dWs : dW; a.db = w.dbs; a.nseg = nseg; a.Vp = w.Vp;
        const int ntile = (V + COLS - 1) / COLS;
        hipLaunchKernelGGL((ce_chain_kernel<1>), dim3((unsigned)(ntile * nseg)), dim3(512), NSLOT * SLOT, st, a);
        TDM_CHECK_LAUNCH("ce_chain(B)");
    }
    ReduceArgs ra{};
    ra.nsec = 1;
    if (nseg > 1) {
        ra.sec[0].off = 0; ra.sec[0].len = (int)((long)V * DK); ra.sec[0].nslab = nseg; ra.sec[0].stride_override = (long)V * DK;
        TDM_TRY(tdm_launch_reduce(w.dWs, 0, ra, dW, st));
    }
    ra.sec[0].off = 0; ra.sec[0].len = V; ra.sec[0].nslab = nseg; ra.sec[0].stride_override = w.Vp;
    return tdm_launch_reduce(w.dbs, 0, ra, db, st);
}

}  // extern "C"
