// Workgroup dispatch-rate probe: how long does a grid of N do-nothing 256-thread workgroups take, as a function of the
// LDS allocation (which bounds workgroups per CU) and register footprint?   hipcc --offload-arch=gfx950 -O3 dispatch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
extern "C" __global__ __launch_bounds__(256) void empty_kernel(float* out, int flag) {
    extern __shared__ float sm[];
    if (flag) { sm[threadIdx.x] = 1.f; __syncthreads(); out[blockIdx.x] = sm[(threadIdx.x + 1) & 255]; }
}
extern "C" __global__ __launch_bounds__(256, 2) void fat_kernel(float* out, int flag) {   // 200+ VGPRs allocated
    extern __shared__ float sm[];
    float v[200];
#pragma unroll
    for (int i = 0; i < 200; ++i) v[i] = out[i + flag];
    if (flag) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 200; ++i) s += v[i] * v[(i * 7) % 200];
        out[blockIdx.x] = s + sm[0];
    }
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grids[] = {1568, 12544};
    const int ldss[] = {0, 54432, 80000};
    hipFuncSetAttribute((const void*)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    hipFuncSetAttribute((const void*)fat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    for (int which = 0; which < 2; ++which)
        for (int g : grids)
            for (int l : ldss) {
                for (int r = 0; r < 3; ++r) {
                    if (which == 0) hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(256), l, 0, d, 0);
                    else hipLaunchKernelGGL(fat_kernel, dim3(g), dim3(256), l, 0, d, 0);
                }
                hipEventRecord(e0);
                for (int r = 0; r < 20; ++r) {
                    if (which == 0) hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(256), l, 0, d, 0);
                    else hipLaunchKernelGGL(fat_kernel, dim3(g), dim3(256), l, 0, d, 0);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("%s grid=%d lds=%d: %.1f us per launch\n", which ? "fat(200 vgpr loads)" : "empty", g, l, ms * 1000 / 20);
            }
    return 0;
}
