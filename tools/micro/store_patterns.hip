// What a store PATTERN costs on MI355X: the same 51.4 MB written by 16-byte stores in four shapes.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/store_patterns.hip -o /tmp/store_patterns && /tmp/store_patterns
//   0 linear:   a wave-instruction writes 1 KiB contiguous
//   1 runs64:   16 runs of 64 B at a 128-B pitch per instruction (one S16 group per pixel), two instructions per 128 B
//   2 runs32:   32 runs of 32 B at a 64-B pitch (hi halves, then lo halves: the split-piece S16 store of 8 channels / lane)
//   3 pieces8:  8-byte stores, 64 runs of 16 B ... (tdm_store_s16_4: 4 channels per lane)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int PAT>
__global__ __launch_bounds__(256) void wr(char* out, long npix) {   // 128 B per "pixel"
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nwave = ((long)gridDim.x * 256) >> 6;
    const uint4 v = make_uint4(1u, 2u, 3u, lane);
    for (long p0 = wave * 16; p0 + 16 <= npix; p0 += nwave * 16) {   // 16 pixels = 2 KiB per wave and trip
        char* base = out + p0 * 128;
        if (PAT == 0) {
            *reinterpret_cast<uint4*>(base + lane * 16) = v;
            *reinterpret_cast<uint4*>(base + 1024 + lane * 16) = v;
        } else if (PAT == 1) {
            char* q = base + (lane >> 2) * 128 + (lane & 3) * 16;
            *reinterpret_cast<uint4*>(q) = v;
            *reinterpret_cast<uint4*>(q + 64) = v;
        } else if (PAT == 2) {
            char* q = base + (lane >> 2) * 128 + ((lane & 3) >> 1) * 64 + (lane & 1) * 16;
            *reinterpret_cast<uint4*>(q) = v;
            *reinterpret_cast<uint4*>(q + 32) = v;
        } else {
            // 8 lanes per pixel, each 4 channels: hi 8 B at group*64 + (c&3)*8, lo at +32; two pixels' worth per 16 lanes: 4 stores
            for (int half = 0; half < 2; ++half) {
                char* q = base + (half * 8 + (lane >> 3)) * 128 + ((lane & 7) >> 2) * 64 + (lane & 3) * 8;
                *reinterpret_cast<uint2*>(q) = make_uint2(v.x, v.y);
                *reinterpret_cast<uint2*>(q + 32) = make_uint2(v.z, v.w);
            }
        }
    }
}
template <int PAT> void run(char* d, long npix, int grid, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wr<PAT>, dim3(grid), dim3(256), 0, 0, d, npix);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(wr<PAT>, dim3(grid), dim3(256), 0, 0, d, npix);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-9s grid %5d: %7.1f us  %6.2f TB/s\n", name, grid, ms / 20 * 1e3, npix * 128.0 / (ms / 20 * 1e-3) / 1e12);
}
int main() {
    const long npix = 401408;   // B = 512 images of 28 x 28, 32 channels x 4 B
    char* d; hipMalloc(&d, npix * 128 * 2);
    for (int grid : {1024, 2048, 8192}) {
        run<0>(d, npix, grid, "linear"); run<1>(d, npix, grid, "runs64"); run<2>(d, npix, grid, "runs32"); run<3>(d, npix, grid, "pieces8");
    }
    return 0;
}
