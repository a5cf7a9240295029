// Microbenchmark: cost of ds_read_b128 with all 64 lanes vs 2 active lanes, and of wave_shr DPP moves (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[80 * 512];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 80 * 512 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const char* p = lds + (lane & 31) * 80 + (lane >> 5) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {            // full-wave b128 read
                const f4 v = *reinterpret_cast<const f4*>(p + u * 160);
                acc += v;
            } else if (MODE == 1) {     // 2 active lanes
                if ((lane & 31) == 0) { const f4 v = *reinterpret_cast<const f4*>(p + u * 160); acc += v; }
            } else {                    // 4 DPP wave_shr:1 moves (one b128 fragment)
                f4 v;
                v[0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc[0]), 0x138, 0xf, 0xf, false));
                v[1] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc[1]), 0x138, 0xf, 0xf, false));
                v[2] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc[2]), 0x138, 0xf, 0xf, false));
                v[3] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc[3]), 0x138, 0xf, 0xf, false));
                acc += v + (float)u;
            }
        }
        asm volatile("" ::: "memory");
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <int MODE> float run(float* d, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, d, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float* d; hipMalloc(&d, 512 * 256 * 4);
    const int iters = 2000;
    const float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters);
    // 512 WGs x 4 waves over 256 CUs = 8 waves per CU; per wave iters*16 ops
    const double ops = (double)iters * 16;
    printf("full b128: %.3f ms  (%.1f ns/op/wave)\n2-lane b128: %.3f ms (%.1f ns/op/wave)\ndpp x4: %.3f ms (%.1f ns/op/wave)\n",
           t0, t0 * 1e6 / ops, t1, t1 * 1e6 / ops, t2, t2 * 1e6 / ops);
    // verify wave_shr semantics once
    return 0;
}
