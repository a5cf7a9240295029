/* A host written in plain C against include/tdm_hip.h — no torch, no Python: hipMalloc, a few entry points of the
 * MNIST path, results checked on the host.  Built and run by tests/test_gpu_c_abi.py:
 *   gcc smoke.c -I include -I /opt/rocm/include -L tinydiffusionmodels_amd/csrc -ltdm_hip -L /opt/rocm/lib -lamdhip64 -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include "tdm_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define TD(x) do { int r_ = (x); if (r_ != 0) { printf("tdm error %d: %s (%s:%d)\n", r_, tdm_last_error(), __FILE__, __LINE__); return 3; } } while (0)

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (float)(*s >> 8) / 16777216.0f; }

int main(void) {
    const int64_t B = 4, inner = 784;
    printf("tdm_version %d, %d UNet parameters\n", tdm_version(), (int)TDM_UNET_NPARAM);
    /* schedule tables (src/mnist.py:23-33) in double on the host, rounded once: good enough for a smoke check */
    float acp_sqrt[1000], om_sqrt[1000];
    double cp = 1.0;
    for (int i = 0; i < 1000; ++i) {
        const double beta = 1e-4 + (2e-2 - 1e-4) * i / 999.0;
        cp *= 1.0 - beta;
        acp_sqrt[i] = (float)sqrt(cp); om_sqrt[i] = (float)sqrt(1.0 - cp);
    }
    unsigned seed = 1;
    float *x0 = malloc(B * inner * 4), *nz = malloc(B * inner * 4), *out = malloc(B * inner * 4);
    int64_t t[4] = {0, 17, 500, 999};
    for (int64_t i = 0; i < B * inner; ++i) { x0[i] = 2.f * frand(&seed) - 1.f; nz[i] = 2.f * frand(&seed) - 1.f; }
    float *dx0, *dnz, *dout, *da, *db; int64_t* dt;
    CK(hipMalloc((void**)&dx0, B * inner * 4)); CK(hipMalloc((void**)&dnz, B * inner * 4)); CK(hipMalloc((void**)&dout, B * inner * 4));
    CK(hipMalloc((void**)&da, 4000)); CK(hipMalloc((void**)&db, 4000)); CK(hipMalloc((void**)&dt, B * 8));
    CK(hipMemcpy(dx0, x0, B * inner * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dnz, nz, B * inner * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(da, acp_sqrt, 4000, hipMemcpyHostToDevice)); CK(hipMemcpy(db, om_sqrt, 4000, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, t, B * 8, hipMemcpyHostToDevice));
    /* q_sample (src/mnist.py:36-42): bit-exact against the same two-rounding fp32 expression on the host */
    TD(tdm_q_sample_f32(dx0, dnz, dt, da, db, dout, B, inner, NULL));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, dout, B * inner * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int64_t b = 0; b < B; ++b)
        for (int64_t i = 0; i < inner; ++i) {
            const volatile float p1 = acp_sqrt[t[b]] * x0[b * inner + i], p2 = om_sqrt[t[b]] * nz[b * inner + i];
            const float ref = p1 + p2;
            if (out[b * inner + i] != ref) ++bad;
        }
    printf("q_sample: %d mismatching elements of %d\n", bad, (int)(B * inner));
    /* UNet forward on zero parameters except out.bias = 0.25: eps must be exactly 0.25 everywhere */
    int32_t offs[33];
    TD(tdm_unet_param_offsets(offs));
    float* params; CK(hipMalloc((void**)&params, TDM_UNET_NPARAM * 4)); CK(hipMemset(params, 0, TDM_UNET_NPARAM * 4));
    const float quarter = 0.25f;
    CK(hipMemcpy(params + offs[31], &quarter, 4, hipMemcpyHostToDevice));   /* tensor 31 = out.bias (state_dict order) */
    const int64_t nws = tdm_unet_workspace_floats(B, 0);
    float *ws, *eps; CK(hipMalloc((void**)&ws, nws * 4)); CK(hipMalloc((void**)&eps, B * inner * 4));
    TD(tdm_unet_fwd_f32(params, dout, dt, eps, ws, B, 0, NULL));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, eps, B * inner * 4, hipMemcpyDeviceToHost));
    int bad2 = 0;
    for (int64_t i = 0; i < B * inner; ++i) if (out[i] != 0.25f) ++bad2;
    printf("unet_fwd (zero weights, out.bias 0.25): %d mismatching elements\n", bad2);
    /* error path: a NULL pointer is refused with a message, nothing is launched */
    const int rc = tdm_unet_fwd_f32(NULL, dout, dt, eps, ws, B, 0, NULL);
    printf("NULL params -> rc %d, \"%s\"\n", rc, tdm_last_error());
    if (bad || bad2 || rc == 0) { printf("FAIL\n"); return 1; }
    printf("C ABI smoke OK\n");
    return 0;
}
