#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void swap32(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__global__ void k(const float* in, float* out, float* out2) {
    f32x16 a;
    for (int i = 0; i < 16; ++i) a[i] = in[threadIdx.x * 16 + i];
    float v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float x = a[r], y = a[4 + r];
        swap32(x, y);
        v[r] = x; v[4 + r] = y;
    }
    for (int i = 0; i < 8; ++i) out[threadIdx.x * 8 + i] = v[i];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float x = a[r], y = a[4 + r];
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
        v[r] = __uint_as_float(sw[0]); v[4 + r] = __uint_as_float(sw[1]);
    }
    for (int i = 0; i < 8; ++i) out2[threadIdx.x * 8 + i] = v[i];
}
int main() {
    float h[64 * 16], o[64 * 8], o2[64 * 8];
    for (int t = 0; t < 64; ++t) for (int i = 0; i < 16; ++i) h[t * 16 + i] = t * 100 + i;
    float *d, *e, *f; hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o)); hipMalloc(&f, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e, f);
    hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost); hipMemcpy(o2, f, sizeof(o2), hipMemcpyDeviceToHost);
    for (int t : {0, 1, 32, 33}) { printf("asm     lane %2d:", t); for (int i = 0; i < 8; ++i) printf(" %g", o[t * 8 + i]); printf("\n");
                                   printf("builtin lane %2d:", t); for (int i = 0; i < 8; ++i) printf(" %g", o2[t * 8 + i]); printf("\n"); }
}
