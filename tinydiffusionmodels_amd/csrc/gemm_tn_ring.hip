// Token-major ("TN") bf16 MFMA GEMM over pre-split (S16) operands, 256 x 256 output tiles, LDS-DMA operand ring — the
// transformer denoiser's weight gradients dW[out][in] = sum over tokens dY[token][out] * X[token][in]
// (src/shakespeare.py:105-120 backward: packed in_proj, out_proj, FFN 256 <-> 2048) at config 5's 32,768 tokens.
// Same arithmetic as gemm_tn_bf16_kernel (gemm_bf16.hip): hi*lo + lo*hi + hi*hi per 16-token K step on
// v_mfma_f32_32x32x16_bf16 (NPROD = 3) or hi*hi alone (NPROD = 1), partial sums per token split into slabs, bias gradient
// = per-split column sums of the first operand.
//
// Why another kernel.  The 128 x 128-tile kernel stages every 32-token chunk of BOTH operands once per tile: the 2048 x 256
// and 256 x 2048 gradients re-read the 256-wide operand 16 times (1.07 GB moved L2 -> LDS for 302 MB of tensors) through
// ds_write_b128 (79 B/clk/CU) into plane images.  Here
//   * a workgroup (8 waves, 128 x 64 outputs per wave = 128 accumulator registers) owns a 256 x 256 tile of one token split:
//     536 MB staged for the same product, and one workgroup per CU (256 registers per lane, two waves per SIMD);
//   * the operands need no conversion, so a token's 256-column slice (1 KiB) is ONE `buffer_load_dwordx4 ... lds`
//     wave-instruction, copied as it lies in memory — [16 hi | 16 lo] per 16-column group — with no staging registers and no
//     ds_write; four stages of 16 tokens (33 KB each), three in flight while one is multiplied, one barrier per stage;
//   * fragments come from `ds_read_b64_tr_b16` straight out of those rows: token pitch 2080 B (= 32 mod 256) and the four
//     token rows of a read chosen as {0, 1, 4, 5} (+2 for the second half of the pair, +8 for the upper k half) cover the 64
//     banks exactly once although the hi halves alone occupy only every other 32-byte slot of a row;
//   * several products of the same token count run as ONE launch (TnJobs): {linear1, linear2} and {in_proj, out_proj} are 16
//     and 4 tiles, so 16 / 64 token splits fill the 256 CUs with the slab traffic of the separate launches;
//   * workgroups are dealt to XCDs so that one XCD runs whole splits: the tiles that share an operand's token range run
//     side by side on one L2.
#include "tdm_common.h"
#include <cstdlib>
#include <type_traits>
#include "tdm_transformer.h"
#include "tdm_s16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

namespace tdm_tnr {

constexpr int TT = 256;                 // tile edge (columns of either operand)
constexpr int SK = 16;                  // tokens per stage = one MFMA K step
constexpr int ROWB = TT * 4;            // bytes of one token's 256-column S16 slice
constexpr int PITCH = 2 * ROWB + 32;    // [A slice | B slice | pad]: 2080 = 32 mod 256 (see the bank note above)
constexpr int STAGE = SK * PITCH;       // 33,280 B
constexpr int NST = 4;
constexpr int LDSB = NST * STAGE;       // 133,120 B
constexpr int NDMA = 4;                 // wave-instructions per wave and stage: 2 tokens x 2 operands
static_assert(PITCH % 256 == 32, "bank layout of the transposed reads");

#define TDM_LDS3(p) ((__attribute__((address_space(3))) void*)(p))
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// The transposed fragment reads are inline assembly on purpose.  Through the builtin the compiler knows they read LDS, and
// it orders every LDS read behind every LDS-DMA write it has seen issued: an `s_waitcnt vmcnt(0)` in front of the first read of
// each stage, i.e. a wait for the three stages requested AHEAD (the whole point of the ring).  The stage's own rows are covered
// by the counted wait + barrier at the top of the iteration; the reads' results are ordered in front of their MFMAs by the
// counted lgkmcnt waits below, which name the registers they release.
struct Frag { s16x4 a, b; };   // tokens {0,1,4,5} / {2,3,6,7} (+8 in the upper k half) of one 16-column half: one MFMA operand
template <int OFF> __device__ __forceinline__ void tr_read(Frag& f, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.a) : "v"(addr), "n"(OFF) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.b) : "v"(addr), "n"(OFF + 2 * PITCH) : "memory");
}
template <int N> __device__ __forceinline__ void wait_lgkm(Frag& f0, Frag& f1) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f0.a), "+v"(f0.b), "+v"(f1.a), "+v"(f1.b) : "n"(N));
}
template <int N> __device__ __forceinline__ void wait_lgkm(Frag& f0, Frag& f1, Frag& f2, Frag& f3) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(f0.a), "+v"(f0.b), "+v"(f1.a), "+v"(f1.b), "+v"(f2.a), "+v"(f2.b), "+v"(f3.a), "+v"(f3.b) : "n"(N));
}
__device__ __forceinline__ bf16x8 frag8(const Frag& f) {
    s16x8 r;
    r[0] = f.a[0]; r[1] = f.a[1]; r[2] = f.a[2]; r[3] = f.a[3];
    r[4] = f.b[0]; r[5] = f.b[1]; r[6] = f.b[2]; r[7] = f.b[3];
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ float sum8(bf16x8 v) {
    const uint4 u = __builtin_bit_cast(uint4, v);
    return ((__uint_as_float(u.x << 16) + __uint_as_float(u.x & 0xffff0000u)) + (__uint_as_float(u.y << 16) + __uint_as_float(u.y & 0xffff0000u))) +
           ((__uint_as_float(u.z << 16) + __uint_as_float(u.z & 0xffff0000u)) + (__uint_as_float(u.w << 16) + __uint_as_float(u.w & 0xffff0000u)));
}

template <int NPROD>
__global__ __launch_bounds__(512) void gemm_tn_ring_kernel(TnJobs js) {
    extern __shared__ float4 tnr_smem4[];
    char* const lds = reinterpret_cast<char*>(tnr_smem4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;      // first-operand columns wm * 128 .. +127, second-operand columns wn * 64 .. +63
    // workgroup -> (token split, tile).  Workgroups go to the 8 XCDs round-robin by linear id; with a split count that is a
    // multiple of 8, XCD x runs splits x * sk/8 .. (x + 1) * sk/8 - 1 whole, tile after tile.
    int split, tile;
    {
        const int lin = blockIdx.x;
        if (js.use_map) {
            const unsigned e = js.map[lin];
            tile = (int)(e & 255u); split = (int)(e >> 8);
        } else if ((js.splitk & 7) == 0) {
            const int xcd = lin & 7, k = lin >> 3;
            split = xcd * (js.splitk >> 3) + k / js.ntiles;
            tile = k % js.ntiles;
        } else {
            split = lin / js.ntiles;
            tile = lin - split * js.ntiles;
        }
    }
    int jj = 0;
#pragma unroll
    for (int q = 1; q < TDM_TN_JOBS; ++q)
        if (q < js.njobs && tile >= js.j[q].tile0) jj = q;
    const TnJob& J = js.j[jj];
    const float* const Ap = J.A; const float* const Bp = J.B;
    const int a_cs4 = (int)J.a_cs * 4, b_rs4 = (int)J.b_rs * 4;   // bytes per token row
    const int M = J.M, N = J.N;
    const int lt = tile - J.tile0;
    const int by = lt / J.tn, bx = lt - by * J.tn;
    const int i0 = by * TT, j0 = bx * TT;
    const int chunk = ((js.K + js.splitk - 1) / js.splitk + SK - 1) / SK * SK;
    const int kbeg = min(split * chunk, js.K), kend = min(js.K, kbeg + chunk);
    const int nst = (kend - kbeg + SK - 1) / SK;

    // rows at and past `kend` are beyond num_records: they land as zeros (tools/micro/dma_oob.hip)
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ap), 0, kend * a_cs4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bp), 0, kend * b_rs4, 0x00020000);
    // lane -> 16-byte piece `lane` of the slice (16-column group lane >> 2); groups past the matrix edge are not fetched
    const int voffA = (i0 + (lane >> 2) * 16 < M) ? i0 * 4 + lane * 16 : (int)0x80000000;
    const int voffB = (j0 + (lane >> 2) * 16 < N) ? j0 * 4 + lane * 16 : (int)0x80000000;
    const bool no_dma = (TDM_ABLATE(js.ablate) & 1) != 0;
    auto issue = [&](int s) {   // stage s & 3 <- tokens kbeg + 16 s .. +15; this wave: tokens 2 wave, 2 wave + 1
        if (no_dma && s > 3) return;
        char* const st = lds + (s & (NST - 1)) * STAGE + wave * 2 * PITCH;
        const int tk = kbeg + ((TDM_ABLATE(js.ablate) & 16) ? 0 : s * SK) + wave * 2;   // (diagnostics, 16: every stage re-reads the split's first 16 tokens — L2 hits)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, TDM_LDS3(st + u * PITCH), 16, voffA, (tk + u) * a_cs4, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, TDM_LDS3(st + u * PITCH + ROWB), 16, voffB, (tk + u) * b_rs4, 0, 0);
        }
    };

    // fragment addresses: lane -> (16-column half cb, k half hh, token row q of the read, 4-column piece pcq)
    const int g4 = lane >> 4, cb = g4 & 1, hh = g4 >> 1, q = (lane >> 2) & 3, pcq = lane & 3;
    const int tok0 = hh * 8 + (q & 1) + 4 * (q >> 1);            // the pair's second read: + 2 tokens
    // byte offsets inside a stage: first operand + t * 128 (32-column block t), + 32 (lo); second operand likewise behind ROWB
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned fa = lds0 + tok0 * PITCH + (wm * 8 + cb) * 64 + pcq * 8;
    const unsigned fb = lds0 + tok0 * PITCH + ROWB + (wn * 4 + cb) * 64 + pcq * 8;

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const bool do_cs = J.colsum != nullptr && bx == 0;
    float cs = 0.f;   // column (wm * 4 + wn) * 32 + (lane & 31) of the tile's first operand, this lane's k half

    // Fragments are double-buffered in REGISTERS: while the 24 MFMAs of stage s run, the 24 (NPROD = 1: 12) transposed reads of
    // stage s + 1 are in flight into the other set.  (Read, wait, multiply per stage — with one barrier per stage all eight waves
    // read together and multiply together — kept the matrix pipe idle for every stage's LDS round trip: 76 us of loop for 41 us
    // of MFMAs on the 2048 x 256 gradient.)
    struct FragSet { Frag ah[4], al[4], bh[2], bl[2]; };
    auto read_frags = [&](FragSet& f, unsigned so) {
        const unsigned pa = fa + so, pb = fb + so;
        tr_read<0>(f.bh[0], pb); tr_read<128>(f.bh[1], pb);
        if (NPROD == 3) { tr_read<32>(f.bl[0], pb); tr_read<128 + 32>(f.bl[1], pb); }
        tr_read<0>(f.ah[0], pa);   if (NPROD == 3 || do_cs) tr_read<32>(f.al[0], pa);
        tr_read<128>(f.ah[1], pa); if (NPROD == 3 || do_cs) tr_read<128 + 32>(f.al[1], pa);
        tr_read<256>(f.ah[2], pa); if (NPROD == 3 || do_cs) tr_read<256 + 32>(f.al[2], pa);
        tr_read<384>(f.ah[3], pa); if (NPROD == 3 || do_cs) tr_read<384 + 32>(f.al[3], pa);
    };
    auto land_frags = [&](FragSet& f) {   // every read issued so far is back; names the registers it releases
        wait_lgkm<0>(f.bh[0], f.bh[1], f.ah[0], f.ah[1]);
        wait_lgkm<0>(f.ah[2], f.ah[3]);
        if (NPROD == 3 || do_cs) wait_lgkm<0>(f.al[0], f.al[1], f.al[2], f.al[3]);
        if (NPROD == 3) wait_lgkm<0>(f.bl[0], f.bl[1]);
    };
    // multiply stage s out of `f` while the reads of stage s + 1 (into `n`, LDS byte offset `so`) go out ONE PAIR PER MFMA GAP:
    // issued as a block in front of the MFMAs they kept both waves of a SIMD off the matrix pipe for the 24 issue slots (the
    // barrier puts the waves in step, so both read at the same time and both multiply at the same time)
    auto multiply = [&](const FragSet& f, FragSet& n, unsigned so, bool rd) {
        const unsigned pa = fa + so, pb = fb + so;
        const bf16x8 bhv[2] = {frag8(f.bh[0]), frag8(f.bh[1])};
        bf16x8 blv[2];
        if (NPROD == 3) { blv[0] = frag8(f.bl[0]); blv[1] = frag8(f.bl[1]); }
        auto rd_slot = [&](auto slot) {   // 12 slots, one per MFMA triple (NPROD = 3) of the 4 x 2 accumulators + 4 spare
            constexpr int S = decltype(slot)::value;
            if (!rd) return;
            if constexpr (S == 0) tr_read<0>(n.bh[0], pb);
            if constexpr (S == 1) tr_read<128>(n.bh[1], pb);
            if constexpr (S == 2) { if (NPROD == 3) tr_read<32>(n.bl[0], pb); }
            if constexpr (S == 3) { if (NPROD == 3) tr_read<128 + 32>(n.bl[1], pb); }
            if constexpr (S == 4) tr_read<0>(n.ah[0], pa);
            if constexpr (S == 5) tr_read<128>(n.ah[1], pa);
            if constexpr (S == 6) tr_read<256>(n.ah[2], pa);
            if constexpr (S == 7) tr_read<384>(n.ah[3], pa);
            if constexpr (S == 8) { if (NPROD == 3 || do_cs) tr_read<32>(n.al[0], pa); }
            if constexpr (S == 9) { if (NPROD == 3 || do_cs) tr_read<128 + 32>(n.al[1], pa); }
            if constexpr (S == 10) { if (NPROD == 3 || do_cs) tr_read<256 + 32>(n.al[2], pa); }
            if constexpr (S == 11) { if (NPROD == 3 || do_cs) tr_read<384 + 32>(n.al[3], pa); }
            __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise gathers the MFMAs and leaves the reads in clumps)
        };
        auto mm = [&](auto mtc, auto ntc) {
            constexpr int mt = decltype(mtc)::value, nt = decltype(ntc)::value;
            const bf16x8 ahv = frag8(f.ah[mt]);
            if (NPROD == 3) {
                const bf16x8 alv = frag8(f.al[mt]);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alv, bhv[nt], acc[mt][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                rd_slot(std::integral_constant<int, (mt * 2 + nt) + 0>{});
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahv, blv[nt], acc[mt][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (mt * 2 + nt < 4) rd_slot(std::integral_constant<int, 8 + mt * 2 + nt>{});
            } else {
                rd_slot(std::integral_constant<int, (mt * 2 + nt) + 0>{});
                if constexpr (mt * 2 + nt < 4) rd_slot(std::integral_constant<int, 8 + mt * 2 + nt>{});
            }
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahv, bhv[nt], acc[mt][nt], 0, 0, 0);
        };
        mm(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); mm(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        mm(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); mm(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        mm(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}); mm(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
        mm(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{}); mm(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
        if (do_cs) {   // each wave sums ONE first-operand block: wm * 4 + wn
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                if (mt == wn) cs += sum8(frag8(f.ah[mt])) + sum8(frag8(f.al[mt]));
        }
    };

    // ---- the stream.  The memory counter retires in order, so "stage s has landed" = at most the NDMA operations of each stage
    // requested after it are outstanding.  Every iteration issues (past the split's end the rows are out of range and cost no
    // traffic), so the counts are the same in the tail.  Iteration s: the fragments of stage s are back (every wave is done with
    // stage s's LDS), stage s + 1 has landed for every wave (barrier), stage s + 4 is requested into the stage just vacated
    // — three stages in flight —, stage s + 1 is read into the other fragment set, stage s is multiplied.
    FragSet F0, F1;
    issue(0); issue(1); issue(2); issue(3);
    if (no_dma) wait_vm<0>(); else wait_vm<3 * NDMA>();
    __builtin_amdgcn_s_barrier();
    read_frags(F0, 0u);
    auto iter = [&](FragSet& cur, FragSet& nxt, int s) {
        land_frags(cur);
        if (no_dma) wait_vm<0>(); else wait_vm<2 * NDMA>();
        __builtin_amdgcn_s_barrier();
        issue(s + 4);
        const unsigned so = (unsigned)(((s + 1) & (NST - 1)) * STAGE);
        if (!(TDM_ABLATE(js.ablate) & 2)) multiply(cur, nxt, so, true);
        else read_frags(nxt, so);
    };
    for (int s = 0; s < nst; s += 2) {
        iter(F0, F1, s);
        if (s + 1 < nst) iter(F1, F0, s + 1);
    }
    land_frags(F0); land_frags(F1);   // (the last iteration's look-ahead reads: nothing may be in flight into dead registers)
    wait_vm<0>();

    if (do_cs) {
        cs += __shfl_xor(cs, 32);
        const int col = i0 + (wm * 4 + wn) * 32 + (lane & 31);
        if (lane < 32 && col < M) J.colsum[(long)split * J.colsum_stride + col] = cs;
    }
    if (TDM_ABLATE(js.ablate) & 4) {
        float keep = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) keep += acc[a][b][0] + acc[a][b][15];
        asm volatile("" :: "v"(keep));
        return;
    }
    // slab stores through a descriptor: one lane offset for the whole tile (columns past the matrix -> out of range), the row
    // as the scalar offset, rows past the matrix beyond num_records — no exec mask, no 64-bit address arithmetic per store
    // (128 stores per wave: the per-store mask + address sequence of plain stores was ~1,500 instructions per wave)
    float* const C = J.C + ((TDM_ABLATE(js.ablate) & 8) ? 0L : (long)split * J.c_split_stride);   // (diagnostics, 8: every split stores into slab 0)
    const int h = lane >> 5, jl = lane & 31;
    const int c_rs4 = (int)J.c_rs * 4;
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(C, 0, M * c_rs4, 0x00020000);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = j0 + wn * 64 + nt * 32 + jl;
        const int voff = col < N ? (4 * h) * c_rs4 + col * 4 : (int)0x80000000;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int row0 = i0 + wm * 128 + mt * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[mt][nt][r];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsC, voff, (row0 + (r & 3) + 8 * (r >> 2)) * c_rs4, 0);
            }
        }
    }
}

}  // namespace tdm_tnr
using namespace tdm_tnr;

bool tdm_gemm_tn_ring_ok(const GemmArgs& g) {
    static const bool off = getenv("TDM_TN_RING") && atoi(getenv("TDM_TN_RING")) == 0;   // A/B timing: the 128 x 128 kernel
    return !off && g.s16_in && g.ce_lse == nullptr && g.a_rs == 1 && g.b_cs == 1 && (g.M % 16) == 0 && (g.N % 16) == 0 &&
           (g.a_cs % 16) == 0 && (g.b_rs % 16) == 0 && g.M >= TT && g.N >= TT && g.K >= 1 &&
           ((long)g.K + 64) * g.a_cs * 4 < 2147483647L && ((long)g.K + 64) * g.b_rs * 4 < 2147483647L &&
           (long)g.M * g.c_rs * 4 < 2147483647L && (((uintptr_t)g.A | (uintptr_t)g.B) & 15) == 0;
}

int tdm_tn_ring_add_job(TnJobs& js, const GemmArgs& g) {
    TDM_REQUIRE(js.njobs < TDM_TN_JOBS, "gemm_tn_ring: more than %d products in one launch", TDM_TN_JOBS);
    TDM_REQUIRE(tdm_gemm_tn_ring_ok(g), "gemm_tn_ring: unsupported product (%d x %d over %d tokens)", g.M, g.N, g.K);
    const int sk = g.splitk > 1 ? g.splitk : 1;
    if (js.njobs == 0) { js.K = g.K; js.splitk = sk; js.ablate = g.ablate; }
    TDM_REQUIRE(js.K == g.K && js.splitk == sk, "gemm_tn_ring: the products of one launch share the token count and the split count");
    TnJob& j = js.j[js.njobs++];
    j.A = g.A; j.B = g.B; j.C = g.C; j.colsum = g.colsum;
    j.a_cs = g.a_cs; j.b_rs = g.b_rs; j.c_rs = g.c_rs; j.c_split_stride = g.c_split_stride; j.colsum_stride = g.colsum_stride;
    j.M = g.M; j.N = g.N;
    j.tn = (g.N + TT - 1) / TT;
    j.tile0 = js.ntiles;
    js.ntiles += j.tn * ((g.M + TT - 1) / TT);
    return 0;
}

int tdm_launch_gemm_tn_ring(TnJobs& js, int nprod, hipStream_t st) {
    TDM_REQUIRE(js.njobs >= 1 && js.ntiles >= 1 && js.splitk >= 1 && (nprod == 1 || nprod == 3), "gemm_tn_ring: empty launch");
    js.use_map = 0;
    if ((long)js.ntiles * js.splitk <= TDM_TN_MAP && js.ntiles <= 256 && js.splitk <= 256) {
        // XCD x runs workgroups x, x + 8, ...: walk the (split, product) groups in order and fill XCD after XCD, so a group's
        // tiles sit on one XCD (two where a share boundary cuts it)
        const int total = js.ntiles * js.splitk;
        int xcd = 0, used = 0;
        auto cap = [&](int x) { return (total - x + 7) / 8; };
        for (int z = 0; z < js.splitk; ++z)
            for (int t = 0; t < js.ntiles; ++t) {
                while (used >= cap(xcd)) { ++xcd; used = 0; }
                js.map[xcd + 8 * used] = (unsigned short)(t | (z << 8));
                ++used;
            }
        js.use_map = 1;
    }
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_ring_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_ring_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
        if (e != hipSuccess) {
            tdm_set_error("gemm_tn_ring: hipFuncSetAttribute(%d B LDS) failed: %s", LDSB, hipGetErrorString(e));
            return 100 + (int)e;
        }
        attr = true;
    }
    const long grid = (long)js.ntiles * js.splitk;
    TDM_REQUIRE(grid < 65536L * 16, "gemm_tn_ring: %ld workgroups", grid);
    if (nprod == 3) hipLaunchKernelGGL(gemm_tn_ring_kernel<3>, dim3((unsigned)grid), dim3(512), LDSB, st, js);
    else hipLaunchKernelGGL(gemm_tn_ring_kernel<1>, dim3((unsigned)grid), dim3(512), LDSB, st, js);
    TDM_CHECK_LAUNCH("gemm_tn_ring");
    return 0;
}
