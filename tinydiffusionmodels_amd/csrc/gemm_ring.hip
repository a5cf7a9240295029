// K-contiguous ("NT") bf16 MFMA GEMM over pre-split (S16) operands with an LDS-DMA operand ring — the transformer
// denoiser's linear layers and their data gradients (src/shakespeare.py:105-120: packed in_proj, out_proj, FFN 256 <-> 2048)
// at config 5's size (32,768 tokens).  Same arithmetic, same MFMA order per accumulator and therefore the same bits as
// gemm_nt_bf16_kernel (gemm_bf16.hip): C[M][N] = A[M][K] . B[N][K]^T, hi*lo + lo*hi + hi*hi per 16-deep K step on
// v_mfma_f32_32x32x16_bf16 (NPROD = 3), or hi*hi alone (NPROD = 1, plain bf16 operands).
//
// What is different is how the operands reach the matrix cores.  The register-staged kernel ran
// {barrier, 4-6 ds_write_b128 per thread, barrier, issue the next chunk's loads, 12 MFMAs} per 32-deep chunk with four waves
// per SIMD: the matrix pipe was ~40 % busy inside the loop and idle during every tile's epilogue (268 MB of stores on the
// N = 2048 layer).  Here persistent workgroups walk XCD-contiguous tile ranges, in one of two shapes (RingCfg):
//   BIG   8 waves, tile 256 tokens x 128 columns, three 48 KB stages, one workgroup per CU — the K-heavy layers;
//   TWIN  4 waves, tile 128 x 128, two 32 KB stages, TWO workgroups per CU (A/B only: no faster, see the launcher);
//   * 64 x 64 per wave: 8 ds_read_b128 feed 12 MFMAs per K step (the 32 x 64 form read 6 for 6);
//   * the S16 operands need no conversion, so a chunk (256 + 128 rows x 128 B) is copied global -> LDS by
//     `buffer_load_dwordx4 ... lds` (1 KiB per wave-instruction, 6 per wave and chunk): no staging registers, no ds_write,
//     no vector instruction per piece (per-tile 32-bit row offsets + the chunk's scalar K offset; rows past the end are
//     beyond num_records and land as zeros — tools/micro/dma_oob.hip);
//   * chunk s + D (D = stages - 1) is requested while chunk s is multiplied, ONE barrier per chunk, and the stream of
//     (tile, chunk) pairs runs across tile seams, so the first chunks of the next tile are in flight during the epilogue;
//   * LDS rows are dense (128 B) — an LDS-DMA cannot pad — so the 16-byte pieces of a row are XOR-swizzled with
//     (row >> 1) & 7 on the SOURCE address and on the fragment reads (the same involution): conflict-free ds_read_b128;
//   * the epilogue transposes one 32 x 32 accumulator at a time through the stage the tile's last chunk has just vacated and
//     walks it 8 columns per lane: 16-byte stores that cover whole 64-byte runs of a row.  (A register transpose by
//     v_permlane32_swap — tools/micro/swp2.hip — needs no LDS and halves the epilogue's arithmetic, but its stores touch 32
//     rows x 32 bytes per instruction: measured 185 vs 155 us on the N = 2048 layer.  Store shape matters more.)
#include "tdm_common.h"
#include <cstdlib>
#include "tdm_transformer.h"
#include "tdm_s16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace tdm_ring {

constexpr int RK = 32;
constexpr int ROWB = 128;                       // bytes of one staged row: 32 k = two S16 groups [hi 16 | lo 16]
constexpr int RN = 128;                         // tile columns (weight rows): two waves of 64
constexpr int TP = 36;                          // floats per row of a wave's 32 x 32 transpose block (9 x 16 B)

// WMS = waves along the token dimension (64 rows each); two waves along the columns
template <int WMS_, int NSTAGE_> struct RingCfg {
    static constexpr int WMS = WMS_, NSTAGE = NSTAGE_;
    static constexpr int WAVES = WMS * 2, THREADS = WAVES * 64;
    static constexpr int RM = WMS * 64;
    static constexpr int STAGE_A = RM * ROWB, STAGE_B = RN * ROWB, STAGE = STAGE_A + STAGE_B;
    static constexpr int LDS = NSTAGE * STAGE;
    static constexpr int DMA_A = (RM / 8) / WAVES, DMA_B = (RN / 8) / WAVES, NDMA = DMA_A + DMA_B;   // wave-instructions per wave and chunk
    static constexpr int D = NSTAGE - 1;                                                         // prefetch distance in chunks
    static_assert((RM / 8) % WAVES == 0 && (RN / 8) % WAVES == 0, "whole wave-instructions per wave");
    static_assert(WAVES * 32 * TP * 4 <= STAGE, "the transpose blocks live in one vacated stage");
};
using RingBig = RingCfg<4, 3>;    // 512 threads, 256 x 128, 144 KB
using RingTwin = RingCfg<2, 2>;   // 256 threads, 128 x 128, 64 KB

#define TDM_LDS(p) ((__attribute__((address_space(3))) void*)(p))

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// (int template parameters, not the config type: a kernel instantiated over a type of the anonymous namespace gets a host
//  stub the linker cannot resolve)
template <int NPROD, int WMS, int NST>
__global__ __launch_bounds__(WMS * 128, 2) void gemm_nt_ring_kernel(GemmArgs g, int ntx, int ntiles) {
    using Cf = RingCfg<WMS, NST>;
    constexpr int RM = Cf::RM, NSTAGE = Cf::NSTAGE, STAGE = Cf::STAGE, STAGE_A = Cf::STAGE_A;
    constexpr int DMA_A = Cf::DMA_A, DMA_B = Cf::DMA_B, NDMA = Cf::NDMA, D = Cf::D;
    extern __shared__ float4 ring_smem4[];
    char* const lds = reinterpret_cast<char*>(ring_smem4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;     // rows wm * 64 .. +63 of the tile, columns wn * 64 .. +63
    // Tile -> workgroup map.  Tiles are numbered token-block-major (the ntx column tiles of a 256-token block are consecutive).
    // Workgroups are dealt to the 8 XCDs round-robin by blockIdx, each XCD with its own L2: XCD x owns the contiguous tile
    // range [ntiles x / 8, ntiles (x + 1) / 8) and its q workgroups walk it INTERLEAVED (workgroup k: tiles k, k + q, ...),
    // so the q tiles in flight on an XCD at any time are q / ntx token blocks with all their column tiles — a token block's
    // rows are fetched from HBM once and the other column tiles hit in that L2.  (Contiguous per-workgroup ranges had 16
    // different token blocks per XCD in flight, each re-read 8 times: 552 MB of HBM traffic for the 304 MB of the N = 2048
    // layer's tensors, profiles/r03_conv_traffic.json text_ffn1_gemm of the first round-3 run.)
    const int ngrp = gridDim.x < 8 ? (int)gridDim.x : 8;
    const int grp = blockIdx.x % ngrp, kx = blockIdx.x / ngrp;
    const int qx = ((int)gridDim.x + ngrp - 1 - grp) / ngrp;                       // workgroups of this group
    const int t_lo = (int)((long)ntiles * grp / ngrp), t_hi = (int)((long)ntiles * (grp + 1) / ngrp);
    const int t_beg = t_lo + kx;
    const int ntw = t_hi - t_lo > kx ? (t_hi - t_lo - kx + qx - 1) / qx : 0;      // tiles t_beg, t_beg + qx, ...
    const int nchunk = g.K / RK;
    const int total = ntw * nchunk;               // the workgroup's stream of (tile, chunk) pairs
    if (total == 0) return;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, (int)((long)g.M * g.a_rs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, (int)((long)g.N * g.b_cs * 4), 0x00020000);

    // ---- DMA plan.  Wave-instruction i of this wave fills 8 consecutive rows of the stage: lane -> row (lane >> 3),
    // physical piece (lane & 7); it FETCHES logical piece (lane & 7) ^ ((row >> 1) & 7).  With rows (wave * DMA + i) * 8 +
    // (lane >> 3) and DMA even, that swizzle is ((i & 1) * 4 + (lane >> 4)) for both operands.
    static_assert(DMA_A % 2 == 0 && DMA_B % 2 == 0, "the swizzle formula assumes an even number of wave-instructions per wave");
    // (fixed-size arrays: an array whose bound depends on the kernel's template parameters, captured by a lambda that calls
    //  the LDS-DMA builtin, makes hipcc 7.2 drop the kernel's HOST stub without a diagnostic — undefined symbol at dlopen)
    static_assert(DMA_A <= 4 && DMA_B <= 4, "offset arrays");
    int voffA[4], voffB[4];
    const int lrow = lane >> 3;
    int it_tile = t_beg, it_chunk = 0;            // position of the issue stream
    auto plan = [&](int tile) {
        const int i0 = (tile / ntx) * RM, j0 = (tile % ntx) * RN;
#pragma unroll
        for (int i = 0; i < DMA_A; ++i) {
            const int row = i0 + (wave * DMA_A + i) * 8 + lrow;
            const int lq = (lane & 7) ^ ((i & 1) * 4 + (lane >> 4));
            voffA[i] = row < g.M ? row * (int)g.a_rs * 4 + lq * 16 : (int)0x80000000;
        }
#pragma unroll
        for (int i = 0; i < DMA_B; ++i) {
            const int row = j0 + (wave * DMA_B + i) * 8 + lrow;
            const int lq = (lane & 7) ^ ((i & 1) * 4 + (lane >> 4));
            voffB[i] = row < g.N ? row * (int)g.b_cs * 4 + lq * 16 : (int)0x80000000;
        }
    };
    // request pair s of the stream into stage s % NSTAGE, in two halves so that the main loop can put them between its MFMA groups
    char* ist = lds;
    int ikoff = 0;
    auto issue_a = [&](int s) {
        if (it_chunk == 0) plan(it_tile);
        ist = lds + (s % NSTAGE) * STAGE;
        ikoff = it_chunk * (RK * 4);              // bytes into a row
#pragma unroll
        for (int i = 0; i < DMA_A; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, TDM_LDS(ist + (wave * DMA_A + i) * 1024), 16, voffA[i], ikoff, 0, 0);
    };
    auto issue_b = [&]() {
#pragma unroll
        for (int i = 0; i < DMA_B; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, TDM_LDS(ist + STAGE_A + (wave * DMA_B + i) * 1024), 16, voffB[i], ikoff, 0, 0);
        if (++it_chunk == nchunk) { it_chunk = 0; it_tile += qx; }
    };
    auto issue = [&](int s) { issue_a(s); issue_b(); };

    // ---- fragment addresses (tile- and chunk-invariant): row of the lane inside its wave block + the four swizzled pieces
    const int fj = (j >> 1) & 7;
    int poff[2][2];                               // [ks][hi / lo] byte offset inside a row
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int lo = 0; lo < 2; ++lo) poff[ks][lo] = ((ks * 4 + lo * 2 + h) ^ fj) * 16;
    const int arow = (wm * 64 + j) * ROWB;        // + mt * 32 rows
    const int brow = STAGE_A + (wn * 64 + j) * ROWB;

    f32x16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    zero_acc();

    auto kstep = [&](const char* st, int ks) {
        bf16x8 xh[2], xl[2], wh[2], wl[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            xh[mt] = *reinterpret_cast<const bf16x8*>(st + arow + mt * 32 * ROWB + poff[ks][0]);
            if (NPROD == 3) xl[mt] = *reinterpret_cast<const bf16x8*>(st + arow + mt * 32 * ROWB + poff[ks][1]);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            wh[nt] = *reinterpret_cast<const bf16x8*>(st + brow + nt * 32 * ROWB + poff[ks][0]);
            if (NPROD == 3) wl[nt] = *reinterpret_cast<const bf16x8*>(st + brow + nt * 32 * ROWB + poff[ks][1]);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {   // D[n][m]: weight rows are the MFMA A operand (gemm_nt_bf16_kernel's order)
                if (NPROD == 3) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[nt], xl[mt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[nt], xh[mt], acc[mt][nt], 0, 0, 0);
                }
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[nt], xh[mt], acc[mt][nt], 0, 0, 0);
            }
    };

    // ---- epilogue of one tile: bias, residual, ReLU, ReLU / dropout backward gate, dropout, fp32 and / or S16 output —
    // gemm_nt_bf16_kernel's epilogue (N % 8 == 0 here), one 32 x 32 accumulator at a time through a wave-private block
    auto epilogue = [&](int tile, char* freest) {
        const int i0 = (tile / ntx) * RM, j0 = (tile % ntx) * RN;
        float* const T = reinterpret_cast<float*>(freest) + wave * (32 * TP);
        // a lane walks EIGHT consecutive columns of a row (4 lanes per 32-column row, 16 rows per pass): its S16 output is
        // hi[8] and lo[8] = two 16-byte stores
        const int c8 = lane & 3, r16 = lane >> 2;
        // the bias of both column blocks is fetched before the first store: a load behind stores makes the compiler wait for
        // vmcnt(0) at its first use, i.e. for the acknowledgement of every store issued so far
        float bzz[2][8];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = j0 + wn * 64 + nt * 32 + c8 * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) bzz[nt][e] = 0.f;
            if (g.bias != nullptr && n < g.N) {
                const float4 b0 = *reinterpret_cast<const float4*>(g.bias + n), b1 = *reinterpret_cast<const float4*>(g.bias + n + 4);
                bzz[nt][0] = b0.x; bzz[nt][1] = b0.y; bzz[nt][2] = b0.z; bzz[nt][3] = b0.w;
                bzz[nt][4] = b1.x; bzz[nt][5] = b1.y; bzz[nt][6] = b1.z; bzz[nt][7] = b1.w;
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = j0 + wn * 64 + nt * 32 + c8 * 8;
            const bool ncol = n < g.N;            // (N % 8 == 0: an octet is inside or outside)
            const float* bz = bzz[nt];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(T + j * TP + 8 * q + 4 * h) =
                        make_float4(acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]);
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int rr = it * 16 + r16;
                    const int m = i0 + wm * 64 + mt * 32 + rr;
                    const float4 t0 = *reinterpret_cast<const float4*>(T + rr * TP + c8 * 8);
                    const float4 t1 = *reinterpret_cast<const float4*>(T + rr * TP + c8 * 8 + 4);
                    if (m < g.M && ncol) {
                        float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                        const long o = (long)m * g.c_rs + n;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += bz[e];
                        if (g.res != nullptr) {
                            const float4 r0 = *reinterpret_cast<const float4*>(g.res + o), r1 = *reinterpret_cast<const float4*>(g.res + o + 4);
                            v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                        }
                        if (g.relu) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
                        }
                        if (g.gate != nullptr) {
                            if (g.gate_s16) {   // S16 gate (values >= 0): element nonzero <=> hi or lo nonzero
                                const char* gb = reinterpret_cast<const char*>(g.gate + (o - (n & 15))) + (n & 15) * 2;
                                const uint4 gh = *reinterpret_cast<const uint4*>(gb), gl = *reinterpret_cast<const uint4*>(gb + 32);
                                const unsigned bb[4] = {gh.x | gl.x, gh.y | gl.y, gh.z | gl.z, gh.w | gl.w};
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    v[2 * e] = (bb[e] & 0xffffu) ? v[2 * e] * g.gate_scale : 0.f;
                                    v[2 * e + 1] = (bb[e] >> 16) ? v[2 * e + 1] * g.gate_scale : 0.f;
                                }
                            } else {
                                const float4 g0 = *reinterpret_cast<const float4*>(g.gate + o), g1 = *reinterpret_cast<const float4*>(g.gate + o + 4);
                                const float gz[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = gz[e] > 0.f ? v[e] * g.gate_scale : 0.f;
                            }
                        }
                        if (g.drop.thr != 0u) {
                            const unsigned long long e0 = (unsigned long long)m * (unsigned)g.N + (unsigned)n;
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = tdm_keep(g.drop, e0 + e) ? v[e] * g.drop.scale : 0.f;
                        }
                        if (TDM_ABLATE(g.ablate) & 4) {   // timing diagnostics: the epilogue's arithmetic without its stores
                            float keep = 0.f;
#pragma unroll
                            for (int e = 0; e < 8; ++e) keep += v[e];
                            asm volatile("" :: "v"(keep));
                            continue;
                        }
                        if (g.C != nullptr) {
                            *reinterpret_cast<float4*>(g.C + o) = make_float4(v[0], v[1], v[2], v[3]);
                            *reinterpret_cast<float4*>(g.C + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
                        }
                        if (g.C16 != nullptr) {   // hi[8] | lo[8] of the octet inside its 16-element group: 16 bytes each
                            tdm_bf16x4 h0, l0, h1, l1;
                            tdm_split4(make_float4(v[0], v[1], v[2], v[3]), h0, l0);
                            tdm_split4(make_float4(v[4], v[5], v[6], v[7]), h1, l1);
                            char* base = reinterpret_cast<char*>(g.C16 + (o - (n & 15))) + (n & 15) * 2;
                            bf16x8 hv, lv;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { hv[e] = h0[e]; hv[4 + e] = h1[e]; lv[e] = l0[e]; lv[4 + e] = l1[e]; }
                            *reinterpret_cast<bf16x8*>(base) = hv;
                            *reinterpret_cast<bf16x8*>(base + 32) = lv;
                        }
                    }
                }
            }
        }
    };

    // ---- the stream
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < total) issue(d);
    int cur_tile = t_beg, cur_chunk = 0;
    // Counted waits.  The memory counter retires in order, so "pair s has landed" = at most as many operations outstanding as
    // were issued AFTER pair s's: the D - 1 pairs requested since.  An epilogue's stores are issued between two pairs; with an
    // explicit drain behind the epilogue the wave sits out the acknowledgement of its 16 KB of stores while the matrix pipe
    // idles.  When the epilogue is loads-free after its first store (no residual / gate) and the tile is interior (every
    // store instruction really issues), its `estores` stores are counted for the D iterations whose pairs they sit behind,
    // and drain under the next tile's MFMAs.
    const int estores = 16 * ((g.C != nullptr ? 1 : 0) + (g.C16 != nullptr ? 1 : 0));
    const bool loads_free = g.res == nullptr && g.gate == nullptr;
    int behind = 0;                               // iterations left whose pair has the last epilogue's stores behind it
    for (int s = 0; s < total; ++s) {
        // pair s has landed once each wave's own pieces have and all waves agree
        if (s + D > total) wait_vm<0>();                                           // (the stream's tail: fewer pairs behind)
        else if (behind > 0 && estores == 16) wait_vm<(D - 1) * NDMA + 16>();
        else if (behind > 0 && estores == 32) wait_vm<(D - 1) * NDMA + 32>();
        else wait_vm<(D - 1) * NDMA>();
        if (behind > 0) --behind;
        __builtin_amdgcn_s_barrier();             // ... and every wave has finished reading stage (s - 1) % NSTAGE
        char* const st = lds + (s % NSTAGE) * STAGE;
        const bool more = s + D < total && !(TDM_ABLATE(g.ablate) & 1);   // pair s + D goes into the stage pair s - 1 occupied (free since the barrier)
        // (requesting both K steps' fragments before the first MFMA was measured: K loop unchanged (81 -> 80 us), whole kernel
        //  153 -> 173 us on the N = 2048 layer — the longer live ranges cost the epilogue more than the loop gains)
        if (!(TDM_ABLATE(g.ablate) & 2)) kstep(st, 0);
        if (more) issue_a(s + D);
        if (!(TDM_ABLATE(g.ablate) & 2)) kstep(st, 1);
        if (more) issue_b();
        if (++cur_chunk == nchunk) {
            __builtin_amdgcn_s_barrier();         // all waves are done with the tile's last stage: it becomes the transpose space
            const int i0 = (cur_tile / ntx) * RM, j0 = (cur_tile % ntx) * RN;
            const bool interior = i0 + RM <= g.M && j0 + RN <= g.N;
            if (!(TDM_ABLATE(g.ablate) & 8)) epilogue(cur_tile, st);
            zero_acc();
            cur_chunk = 0; cur_tile += qx;
            if (loads_free && interior && estores > 0 && !(TDM_ABLATE(g.ablate) & 12)) behind = D;
            else wait_vm<0>();                    // (ragged tile or loads in the epilogue: drain, keep the counts exact)
        }
    }
}

template <int NPROD, typename Cf>
int launch_ring(const GemmArgs& g, hipStream_t st) {
    static int resident = 0;   // workgroups the device holds at once: the persistent grid
    if (resident == 0) {
        int dev = 0, n = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_ring_kernel<NPROD, Cf::WMS, Cf::NSTAGE>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS);
        if (e != hipSuccess || n <= 0) {
            tdm_set_error("gemm_nt_ring: device query / LDS attribute (%d B) failed: %s", Cf::LDS, hipGetErrorString(e));
            return 100 + (int)e;
        }
        resident = n * (163840 / Cf::LDS);
    }
    const int ntx = (g.N + RN - 1) / RN, ntiles = ntx * ((g.M + Cf::RM - 1) / Cf::RM);
    const int grid = ntiles < resident ? ntiles : resident;
    hipLaunchKernelGGL((gemm_nt_ring_kernel<NPROD, Cf::WMS, Cf::NSTAGE>), dim3(grid), dim3(Cf::THREADS), Cf::LDS, st, g, ntx, ntiles);
    TDM_CHECK_LAUNCH("gemm_nt_ring");
    return 0;
}


// (A token-major / weight-gradient twin of this kernel — 256 x 128 output tile per (tile, split) workgroup, token rows copied
//  by LDS-DMA as they lie in memory with a token XOR swizzle for ds_read_b64_tr_b16, XCD-contiguous splits — was built in
//  round 3 and was bit-identical to gemm_tn_bf16_kernel but NOT faster: 132 vs 124 us on the 2048 x 256 gradient, 128 vs 118
//  on 256 x 2048.  Those products move 0.8-1.1 GB of operand re-reads per launch through L2 and are bound there, not by
//  staging instructions; DESIGN.md section 5.  It is not compiled in.)

}  // namespace tdm_ring
using namespace tdm_ring;

// true when the ring kernel serves this problem (the caller has already checked the NT form's general requirements)
bool tdm_gemm_nt_ring_ok(const GemmArgs& g) {
    return g.s16_in && (g.K % RK) == 0 && (g.N % 8) == 0 && (g.c_rs % 8) == 0 && g.ce_lse == nullptr && g.ce_part == nullptr &&
           g.splitk <= 1 && (long)g.M * g.a_rs * 4 < 2147483647L && (long)g.N * g.b_cs * 4 < 2147483647L &&
           (long)g.M * g.N >= 256L * 128L * 32L;      // small problems: the 128 x 128 tiles of gemm_nt_bf16_kernel fill the chip better
}

int tdm_launch_gemm_nt_ring(const GemmArgs& g, int nprod, hipStream_t st) {
    TDM_REQUIRE(tdm_gemm_nt_ring_ok(g), "gemm_nt_ring: unsupported problem");
    // BIG is the product's shape.  TWIN (two 4-wave workgroups per CU, TDM_RING_CFG=2) was built to let one workgroup's
    // epilogue run under the other's MFMAs on the K = 256 / N = 2048 layers; measured (tools/time_ring.py, M = 32,768):
    // 166 vs 168 us there, 105 vs 98 us at K = 2048, and a half-tile start offset between the pair changed nothing — the
    // epilogue's ~70 us (268 MB of stores + ~12 vector instructions per element) do not hide under a co-resident workgroup.
    static const int forced = [] { const char* e = getenv("TDM_RING_CFG"); return e ? atoi(e) : 0; }();   // 1 = BIG, 2 = TWIN (A/B timing)
    const bool twin = forced == 2;
    if (twin) return nprod == 3 ? launch_ring<3, RingTwin>(g, st) : launch_ring<1, RingTwin>(g, st);
    return nprod == 3 ? launch_ring<3, RingBig>(g, st) : launch_ring<1, RingBig>(g, st);
}

