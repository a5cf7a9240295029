// Probe for the round-4 LayerNorm-backward miscompute (DESIGN §5a "packed-fp32 code that is only right on a quiet GPU").
// The SLP-vectorised ln_bwd_kernel<1> differs from the scalar build in ONE place that matches the recorded failure signature
// (rows 1 and 3 of a wave's four, x / z components, lanes 48-63): for those rows the x / z products are formed by
//     I0  v_pk_add_f32 v[T:T+1], A, B        packed write of the pair T
//     I1  v_pk_add_f32 U, U, C               (independent)
//     I2  v_add_f32    S, vT, vT+1           reads both halves
//     I3  v_mov_b32    vT,   X               32-bit overwrite of the low half
//     I4  v_mov_b32    vT+1, Z               32-bit overwrite of the high half
//     I5  v_pk_mul_f32 R, v[T:T+1], A        64-bit read of the pair, back to back
// while rows 0 / 2 have an independent instruction between I4 and I5 and the y / w products have no packed write of T before
// the moves.  This program issues exactly that sequence (inline asm, fixed registers) in variants, alone and next to foreign
// kernel streams (an MFMA spinner, a memory streamer), and counts results that differ from the scalar product per 16-lane group.
//     hipcc --offload-arch=gfx950 -O3 -o pk_waw.bin pk_waw.hip && ./pk_waw.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <dlfcn.h>
#include <stdint.h>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

// counters[variant][value 0..4 = R.lo R.hi R2.lo R2.hi S][lane group 0..3]
template <int V>
__global__ __launch_bounds__(256) void probe(const f4* __restrict__ src, long n4, unsigned* __restrict__ counters, int iters) {
    extern __shared__ float occupancy_pad[];
    const int lane = threadIdx.x & 63;
    const long gw = ((long)blockIdx.x * 256 + threadIdx.x);
    f2 U = {0.f, 0.f};
    unsigned bad[5] = {0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const long base = ((gw * 3 + (long)it * 7919 * 3) % (n4 - 3));
        const f4 p = src[base], q = src[base + 1], r = src[base + 2];
        const f2 A = {p.x, p.y}, B = {p.z, p.w}, C = {q.x, q.y};
        const float X = q.z, Z = q.w, Y = r.x, W = r.y;
        f2 R, R2;
        float S, Rl = 0.f, Rh = 0.f;
        if constexpr (V == 0) {          // the sequence as compiled
            asm volatile("v_pk_add_f32 v[100:101], %[A], %[B]\n v_pk_add_f32 %[U], %[U], %[C]\n v_add_f32 %[S], v100, v101\n"
                         "v_mov_b32 v100, %[X]\n v_mov_b32 v101, %[Z]\n v_pk_mul_f32 %[R], v[100:101], %[A]\n"
                         "v_mov_b32 v100, %[Y]\n v_mov_b32 v101, %[W]\n v_pk_mul_f32 %[R2], v[100:101], %[B]\n"
                         : [U] "+v"(U), [S] "=&v"(S), [R] "=&v"(R), [R2] "=&v"(R2)
                         : [A] "v"(A), [B] "v"(B), [C] "v"(C), [X] "v"(X), [Y] "v"(Y), [Z] "v"(Z), [W] "v"(W) : "v100", "v101");
        } else if constexpr (V == 1) {   // idle slots between the moves and the packed read
            asm volatile("v_pk_add_f32 v[100:101], %[A], %[B]\n v_pk_add_f32 %[U], %[U], %[C]\n v_add_f32 %[S], v100, v101\n"
                         "v_mov_b32 v100, %[X]\n v_mov_b32 v101, %[Z]\n s_nop 3\n v_pk_mul_f32 %[R], v[100:101], %[A]\n"
                         "v_mov_b32 v100, %[Y]\n v_mov_b32 v101, %[W]\n s_nop 3\n v_pk_mul_f32 %[R2], v[100:101], %[B]\n"
                         : [U] "+v"(U), [S] "=&v"(S), [R] "=&v"(R), [R2] "=&v"(R2)
                         : [A] "v"(A), [B] "v"(B), [C] "v"(C), [X] "v"(X), [Y] "v"(Y), [Z] "v"(Z), [W] "v"(W) : "v100", "v101");
        } else if constexpr (V == 2) {   // no packed WRITE of the pair before the moves (two scalar adds instead of I0)
            asm volatile("v_add_f32 v100, %[Ax], %[Bx]\n v_add_f32 v101, %[Ay], %[By]\n v_pk_add_f32 %[U], %[U], %[C]\n v_add_f32 %[S], v100, v101\n"
                         "v_mov_b32 v100, %[X]\n v_mov_b32 v101, %[Z]\n v_pk_mul_f32 %[R], v[100:101], %[A]\n"
                         "v_mov_b32 v100, %[Y]\n v_mov_b32 v101, %[W]\n v_pk_mul_f32 %[R2], v[100:101], %[B]\n"
                         : [U] "+v"(U), [S] "=&v"(S), [R] "=&v"(R), [R2] "=&v"(R2)
                         : [A] "v"(A), [B] "v"(B), [C] "v"(C), [X] "v"(X), [Y] "v"(Y), [Z] "v"(Z), [W] "v"(W),
                           [Ax] "v"(p.x), [Ay] "v"(p.y), [Bx] "v"(p.z), [By] "v"(p.w) : "v100", "v101");
        } else if constexpr (V == 3) {   // packed write + moves, but the products are scalar multiplies
            asm volatile("v_pk_add_f32 v[100:101], %[A], %[B]\n v_pk_add_f32 %[U], %[U], %[C]\n v_add_f32 %[S], v100, v101\n"
                         "v_mov_b32 v100, %[X]\n v_mov_b32 v101, %[Z]\n v_mul_f32 %[Rl], v100, %[Ax]\n v_mul_f32 %[Rh], v101, %[Ay]\n"
                         "v_mov_b32 v100, %[Y]\n v_mov_b32 v101, %[W]\n v_pk_mul_f32 %[R2], v[100:101], %[B]\n"
                         : [U] "+v"(U), [S] "=&v"(S), [Rl] "=&v"(Rl), [Rh] "=&v"(Rh), [R2] "=&v"(R2)
                         : [A] "v"(A), [B] "v"(B), [C] "v"(C), [X] "v"(X), [Y] "v"(Y), [Z] "v"(Z), [W] "v"(W), [Ax] "v"(p.x), [Ay] "v"(p.y)
                         : "v100", "v101");
            R.x = Rl; R.y = Rh;
        } else if constexpr (V == 4) {   // the moves build the pair in OTHER registers than the packed add wrote
            asm volatile("v_pk_add_f32 v[100:101], %[A], %[B]\n v_pk_add_f32 %[U], %[U], %[C]\n v_add_f32 %[S], v100, v101\n"
                         "v_mov_b32 v102, %[X]\n v_mov_b32 v103, %[Z]\n v_pk_mul_f32 %[R], v[102:103], %[A]\n"
                         "v_mov_b32 v102, %[Y]\n v_mov_b32 v103, %[W]\n v_pk_mul_f32 %[R2], v[102:103], %[B]\n"
                         : [U] "+v"(U), [S] "=&v"(S), [R] "=&v"(R), [R2] "=&v"(R2)
                         : [A] "v"(A), [B] "v"(B), [C] "v"(C), [X] "v"(X), [Y] "v"(Y), [Z] "v"(Z), [W] "v"(W) : "v100", "v101", "v102", "v103");
        } else {                         // V == 5: no reader (I2) between the packed write and the moves
            asm volatile("v_pk_add_f32 v[100:101], %[A], %[B]\n v_pk_add_f32 %[U], %[U], %[C]\n"
                         "v_mov_b32 v100, %[X]\n v_mov_b32 v101, %[Z]\n v_pk_mul_f32 %[R], v[100:101], %[A]\n"
                         "v_mov_b32 v100, %[Y]\n v_mov_b32 v101, %[W]\n v_pk_mul_f32 %[R2], v[100:101], %[B]\n v_mov_b32 %[S], 0\n"
                         : [U] "+v"(U), [S] "=&v"(S), [R] "=&v"(R), [R2] "=&v"(R2)
                         : [A] "v"(A), [B] "v"(B), [C] "v"(C), [X] "v"(X), [Y] "v"(Y), [Z] "v"(Z), [W] "v"(W) : "v100", "v101");
        }
        const float e0 = __fmul_rn(X, A.x), e1 = __fmul_rn(Z, A.y), e2 = __fmul_rn(Y, B.x), e3 = __fmul_rn(W, B.y);
        const float e4 = V == 5 ? 0.f : __fadd_rn(__fadd_rn(A.x, B.x), __fadd_rn(A.y, B.y));
        bad[0] += __float_as_uint(R.x) != __float_as_uint(e0);
        bad[1] += __float_as_uint(R.y) != __float_as_uint(e1);
        bad[2] += __float_as_uint(R2.x) != __float_as_uint(e2);
        bad[3] += __float_as_uint(R2.y) != __float_as_uint(e3);
        bad[4] += __float_as_uint(S) != __float_as_uint(e4);
    }
    if (U.x == 12345.678f) counters[1023] = 1;   // keep U alive
#pragma unroll
    for (int k = 0; k < 5; ++k)
        if (bad[k]) atomicAdd(&counters[(V * 5 + k) * 4 + (lane >> 4)], bad[k]);
}

// ---- probe 2: the packed add with a CROSS half-select.  tools/ln_slp_forensics.py (the SLP build of ln_bwd_kernel<1> next to the
// foreign GEMM stream, wrong rows decomposed on the host) shows every wrong value to be the LO result, lanes 48-63, of
//     v_pk_add_f32 D, A, B op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]        D.lo = A.lo - B.hi,  D.hi = A.hi - B.hi
// computed as if B.hi were 0 (three sites: the mean subtraction of row 1, "- c1" of rows 1 and 3 at the output).  The variants
// issue that instruction alone, right behind a real s_waitcnt vmcnt(0) (loads and a store in flight), with B wave-uniform
// (loaded by every lane from one address, like mean[row]) or VALU-produced.
// counters2[variant][lo / hi][lane group]
template <int V>
__global__ __launch_bounds__(256) void probe2(const f4* __restrict__ src, long n4, const float* __restrict__ uni, float* __restrict__ sink,
                                              unsigned* __restrict__ counters, int iters) {
    extern __shared__ float occupancy_pad[];
    const int lane = threadIdx.x & 63;
    const long gw = ((long)blockIdx.x * 256 + threadIdx.x);
    const long wv = gw >> 6;
    unsigned bad[2] = {0, 0};
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
        const long base = ((gw * 2 + (long)it * 7919 * 2) % (n4 - 2));
        const f4 p = src[base], q = src[base + 1];
        // wave-uniform pair (every lane loads the same two dwords, like mean[row0], mean[row0 + 1])
        const long ui = ((wv * 2 + (long)it * 13) % 4096);
        const float u0 = uni[ui], u1 = uni[ui + 1];
        sink[gw] = keep;                                     // a store in flight (gfx9: vmcnt counts stores too)
        f2 A = {p.x, p.y}, A2 = {p.z, p.w}, B, R, R2;
        if constexpr (V == 4) { B[0] = q.x * q.y; B[1] = q.z * q.w; }   // VALU-produced pair (per lane)
        else { B[0] = u0; B[1] = u1; }
        if constexpr (V == 0 || V == 4)      // the failing form
            asm volatile("s_waitcnt vmcnt(0)\n v_pk_add_f32 %[R], %[A], %[B] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %[R2], %[A2], %[B] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                         : [R] "=&v"(R), [R2] "=&v"(R2) : [A] "v"(A), [A2] "v"(A2), [B] "v"(B) : "memory");
        else if constexpr (V == 1)           // without the negation
            asm volatile("s_waitcnt vmcnt(0)\n v_pk_add_f32 %[R], %[A], %[B] op_sel:[0,1]\n v_pk_add_f32 %[R2], %[A2], %[B] op_sel:[0,1]\n"
                         : [R] "=&v"(R), [R2] "=&v"(R2) : [A] "v"(A), [A2] "v"(A2), [B] "v"(B) : "memory");
        else if constexpr (V == 2)           // the other broadcast (hi result from the low dword): rows 0 / 2, never seen wrong
            asm volatile("s_waitcnt vmcnt(0)\n v_pk_add_f32 %[R], %[A], %[B] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %[R2], %[A2], %[B] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                         : [R] "=&v"(R), [R2] "=&v"(R2) : [A] "v"(A), [A2] "v"(A2), [B] "v"(B) : "memory");
        else if constexpr (V == 5) {         // IN PLACE (destination = first source), the form at every failing site
            R = A; R2 = A2;
            asm volatile("s_waitcnt vmcnt(0)\n v_pk_add_f32 %[R], %[R], %[B] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %[R2], %[R2], %[B] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                         : [R] "+v"(R), [R2] "+v"(R2) : [B] "v"(B) : "memory");
        } else                               // V == 3: idle slots between the wait and the packed add
            asm volatile("s_waitcnt vmcnt(0)\n s_nop 7\n v_pk_add_f32 %[R], %[A], %[B] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %[R2], %[A2], %[B] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                         : [R] "=&v"(R), [R2] "=&v"(R2) : [A] "v"(A), [A2] "v"(A2), [B] "v"(B) : "memory");
        float e0, e1, e2, e3;
        if constexpr (V == 1) { e0 = __fadd_rn(A[0], B[1]); e1 = __fadd_rn(A[1], B[1]); e2 = __fadd_rn(A2[0], B[1]); e3 = __fadd_rn(A2[1], B[1]); }
        else if constexpr (V == 2) { e0 = __fsub_rn(A[0], B[0]); e1 = __fsub_rn(A[1], B[0]); e2 = __fsub_rn(A2[0], B[0]); e3 = __fsub_rn(A2[1], B[0]); }
        else { e0 = __fsub_rn(A[0], B[1]); e1 = __fsub_rn(A[1], B[1]); e2 = __fsub_rn(A2[0], B[1]); e3 = __fsub_rn(A2[1], B[1]); }
        bad[0] += (__float_as_uint(R[0]) != __float_as_uint(e0)) + (__float_as_uint(R2[0]) != __float_as_uint(e2));
        bad[1] += (__float_as_uint(R[1]) != __float_as_uint(e1)) + (__float_as_uint(R2[1]) != __float_as_uint(e3));
        keep += R[0] + R2[1];
    }
    if (keep == 12345.678f) counters[1023] = 1;
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (bad[k]) atomicAdd(&counters[512 + (V * 2 + k) * 4 + (lane >> 4)], bad[k]);
}

// foreign work 1: MFMA spinner (no memory traffic)
__global__ __launch_bounds__(256) void mfma_spin(float* out, int iters) {
    f16v acc = {0};
    h4 a = {(_Float16)1.f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)0.125f}, b = a;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, acc, 0, 0, 0);
    }
    if (acc[0] == 1.2345f) out[0] = acc[3];
}
// foreign work 2: memory streamer
__global__ __launch_bounds__(256) void mem_stream(const f4* __restrict__ a, f4* __restrict__ b, long n4, int reps) {
    for (int r = 0; r < reps; ++r)
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
            f4 v = a[i]; v.x += 1.f; b[i] = v;
        }
}

template <int V>
static void launch(const f4* src, long n4, unsigned* counters, int iters, int lds, hipStream_t st) {
    CK(hipFuncSetAttribute((const void*)probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipLaunchKernelGGL(probe<V>, dim3(2048), dim3(256), lds, st, src, n4, counters, iters);
}

template <int V>
static void launch2(const f4* src, long n4, const float* uni, float* sink, unsigned* counters, int iters, int lds, hipStream_t st) {
    CK(hipFuncSetAttribute((const void*)probe2<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipLaunchKernelGGL(probe2<V>, dim3(2048), dim3(256), lds, st, src, n4, uni, sink, counters, iters);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    const long n4 = 1L << 24;                       // 256 MB of inputs: the probe's loads miss
    f4 *src, *dst; float *junk, *uni, *sink; unsigned* counters;
    CK(hipMalloc(&src, n4 * 16)); CK(hipMalloc(&dst, n4 * 16)); CK(hipMalloc(&junk, 4096)); CK(hipMalloc(&counters, 4096));
    CK(hipMalloc(&uni, 8192 * 4)); CK(hipMalloc(&sink, 2048L * 256 * 4));
    std::vector<float> h(n4 * 4);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.f / (1 << 22)); }
    CK(hipMemcpy(src, h.data(), n4 * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(uni, h.data() + 1000, 8192 * 4, hipMemcpyHostToDevice));
    hipStream_t main_st, side_a, side_b;
    CK(hipStreamCreate(&main_st)); CK(hipStreamCreateWithPriority(&side_a, hipStreamNonBlocking, 0)); CK(hipStreamCreateWithPriority(&side_b, hipStreamNonBlocking, 0));
    // foreign work 3 (mode 4): the library's own token-major bf16x3 GEMM, 2048 x 256 x 32768 in 8 splits - the stream next to which the
    // SLP build of ln_bwd_kernel<1> went wrong (tests/test_gpu_text.py, tools/ln_slp_forensics.py)
    typedef int (*split_fn)(const float*, float*, int64_t, void*);
    typedef int (*gemm_fn)(const float*, int64_t, int64_t, const float*, int64_t, int64_t, float*, int64_t, const float*, const float*, int, int, int, int, int,
                           int64_t, void*);
    typedef int (*mode_fn)(int);
    split_fn lib_split = nullptr; gemm_fn lib_gemm = nullptr;
    float *ga = nullptr, *gb = nullptr, *ga16 = nullptr, *gb16 = nullptr, *gslab = nullptr;
    if (void* h = dlopen(argc > 2 ? argv[2] : "tinydiffusionmodels_amd/csrc/libtdm_hip.so", RTLD_NOW)) {
        lib_split = (split_fn)dlsym(h, "tdm_split_s16_f32"); lib_gemm = (gemm_fn)dlsym(h, "tdm_gemm_f32");
        if (mode_fn sm = (mode_fn)dlsym(h, "tdm_set_gemm_mode")) sm(1);
        const long Ms = 32768;
        CK(hipMalloc(&ga, Ms * 2048 * 4)); CK(hipMalloc(&gb, Ms * 256 * 4)); CK(hipMalloc(&ga16, Ms * 2048 * 4)); CK(hipMalloc(&gb16, Ms * 256 * 4));
        CK(hipMalloc(&gslab, 8L * 2048 * 256 * 4));
        CK(hipMemcpy(ga, src, Ms * 2048 * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(gb, src, Ms * 256 * 4, hipMemcpyDeviceToDevice));
        if (lib_split(ga, ga16, Ms * 2048, nullptr) || lib_split(gb, gb16, Ms * 256, nullptr)) { printf("split failed\n"); lib_gemm = nullptr; }
        CK(hipDeviceSynchronize());
    } else printf("libtdm_hip.so not found (%s): no library-GEMM mode\n", dlerror());
    const char* modes[] = {"quiet", "next to MFMA spinner", "next to memory streamer", "next to both", "next to the library's token-major GEMMs"};
    const int ldss[] = {0, 40960};                  // 40 KB per workgroup: 4 workgroups (4 waves / SIMD) per CU at most, like the 162-VGPR kernel's 3
    const char* vn[] = {"V0 as compiled", "V1 s_nop 3 before the packed read", "V2 scalar adds instead of the packed write",
                        "V3 scalar multiplies instead of the packed read", "V4 moves build the pair in other registers", "V5 no reader between write and moves"};
    for (int lds : ldss)
        for (int m = 0; m < (lib_gemm ? 5 : 4); ++m) {
            CK(hipMemset(counters, 0, 4096));
            CK(hipDeviceSynchronize());
            if (m == 4)
                for (int r = 0; r < 1500; ++r)
                    if (lib_gemm(ga16, 1, 2048, gb16, 256, 1, gslab, 256, nullptr, nullptr, 2048, 256, 32768, 2, 8, 2048L * 256, side_a)) { printf("gemm failed\n"); break; }
            if (m < 4 && (m & 1)) hipLaunchKernelGGL(mfma_spin, dim3(1024), dim3(256), 0, side_a, junk, 60000);
            if (m < 4 && (m & 2)) hipLaunchKernelGGL(mem_stream, dim3(1024), dim3(256), 0, side_b, src, dst, n4, 40);
            for (int rep = 0; rep < 3; ++rep) {
                launch<0>(src, n4, counters, iters, lds, main_st); launch<1>(src, n4, counters, iters, lds, main_st);
                launch<2>(src, n4, counters, iters, lds, main_st); launch<3>(src, n4, counters, iters, lds, main_st);
                launch<4>(src, n4, counters, iters, lds, main_st); launch<5>(src, n4, counters, iters, lds, main_st);
                launch2<0>(src, n4, uni, sink, counters, iters, lds, main_st); launch2<1>(src, n4, uni, sink, counters, iters, lds, main_st);
                launch2<2>(src, n4, uni, sink, counters, iters, lds, main_st); launch2<3>(src, n4, uni, sink, counters, iters, lds, main_st);
                launch2<4>(src, n4, uni, sink, counters, iters, lds, main_st); launch2<5>(src, n4, uni, sink, counters, iters, lds, main_st);
            }
            CK(hipStreamSynchronize(main_st));
            const hipError_t side_busy = (m ? hipStreamQuery((m & 1) || m == 4 ? side_a : side_b) : hipSuccess);
            CK(hipDeviceSynchronize());
            unsigned c[6 * 5 * 4];
            CK(hipMemcpy(c, counters, sizeof(c), hipMemcpyDeviceToHost));
            printf("== %s, probe LDS %d B (foreign work still running when the probes finished: %s); %ld products per value\n", modes[m], lds,
                   m ? (side_busy == hipErrorNotReady ? "yes" : "NO - contention window too short") : "-", 3L * 2048 * 256 * iters);
            for (int v = 0; v < 6; ++v) {
                unsigned tot = 0;
                for (int k = 0; k < 20; ++k) tot += c[v * 20 + k];
                printf("  %-52s mismatches %u", vn[v], tot);
                if (tot) {
                    const char* kn[] = {"x", "z", "y", "w", "S"};
                    for (int k = 0; k < 5; ++k)
                        printf("  %s[lanes 0-15,16-31,32-47,48-63]=%u,%u,%u,%u", kn[k], c[v * 20 + k * 4], c[v * 20 + k * 4 + 1], c[v * 20 + k * 4 + 2], c[v * 20 + k * 4 + 3]);
                }
                printf("\n");
            }
            unsigned c2[6 * 2 * 4];
            CK(hipMemcpy(c2, counters + 512, sizeof(c2), hipMemcpyDeviceToHost));
            const char* v2n[] = {"P0 v_pk_add_f32 op_sel:[0,1] neg, uniform B, behind vmcnt(0)", "P1 the same without neg", "P2 op_sel_hi:[1,0] neg (control)",
                                 "P3 P0 with s_nop 7 behind the wait", "P4 P0 with a VALU-produced per-lane B", "P5 P0 in place (destination = first source)"};
            for (int v = 0; v < 6; ++v) {
                unsigned tot = 0;
                for (int k = 0; k < 8; ++k) tot += c2[v * 8 + k];
                printf("  %-62s mismatches %u   lo[lanes 0-15,16-31,32-47,48-63]=%u,%u,%u,%u  hi=%u,%u,%u,%u\n", v2n[v], tot, c2[v * 8], c2[v * 8 + 1], c2[v * 8 + 2],
                       c2[v * 8 + 3], c2[v * 8 + 4], c2[v * 8 + 5], c2[v * 8 + 6], c2[v * 8 + 7]);
            }
            fflush(stdout);
        }
    return 0;
}
