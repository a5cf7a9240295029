// Probe: what does an out-of-range lane of `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer descriptor) leave in
// LDS — zeros, or the previous contents?  (conv / GEMM loaders rely on descriptor range checks for padding.)
//   hipcc -O3 --offload-arch=gfx950 tools/micro/dma_oob.hip -o /tmp/dma_oob && /tmp/dma_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* src, float* dst, int n) {
    extern __shared__ float4 smem[];
    char* lds = reinterpret_cast<char*>(smem);
    smem[threadIdx.x] = make_float4(-7.f, -7.f, -7.f, -7.f);      // sentinel
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n * 4, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int voff = threadIdx.x * 16;
    if (lane == 5) voff = (int)0x80000000;   // far out of range
    if (lane == 9) voff = n * 4;             // first byte past the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + wave * 1024), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    reinterpret_cast<float4*>(dst)[threadIdx.x] = smem[threadIdx.x];
}
int main() {
    const int n = 256 * 4;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *s, *d;
    hipMalloc(&s, n * 4); hipMalloc(&d, n * 4);
    hipMemcpy(s, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, s, d, n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    printf("lane 4: %g %g | lane 5 (oob far): %g %g %g %g | lane 9 (oob +0): %g %g | lane 69 (wave 1, oob far): %g | lane 70: %g\n", h[16], h[17], h[20], h[21],
           h[22], h[23], h[36], h[37], h[(64 + 5) * 4], h[(64 + 6) * 4]);
    return 0;
}
