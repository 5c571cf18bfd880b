// Does the real GELU+split VALU stream of one wave overlap with another wave's MFMAs on the same SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../phyloformer_amd/csrc/pf_device.hip.h"
using namespace pfk;
// MODE: 1 = waves 0-3 MFMA only (others idle), 2 = waves 4-7 GELU only, 3 = both
template <int MODE, int VK>
__global__ void __launch_bounds__(512, 2) k(float* out, int iters) {
    const bool mf = __builtin_amdgcn_readfirstlane(threadIdx.x) < 256;
    f32x16 acc0, acc1, hv;
    bf16x8 fa, fb, gh, gl;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; hv[i] = 0.01f * (threadIdx.x % 13) + 0.1f * i - 0.7f; }
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(float)(threadIdx.x & 7); fb[i] = (__bf16)0.5f; gh[i] = fa[i]; gl[i] = fb[i]; }
    float sacc = 0.f;
    if (mf && (MODE & 1)) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 12; ++r) { acc0 = PF_MFMA(fa, fb, acc0); acc1 = PF_MFMA(fa, fb, acc1); }
        }
    } else if (!mf && (MODE & 2)) {
        for (int it = 0; it < iters; ++it) {
            if (VK == 0) {
                gelu_split8(hv, 0, gh, gl);
                bf16x8 g2, l2;
                gelu_split8(hv, 8, g2, l2);
#pragma unroll
                for (int i = 0; i < 8; ++i) hv[i] += 1e-3f * ((float)gh[i] + (float)l2[i]);   // keep a dependency
#pragma unroll
                for (int i = 0; i < 8; ++i) hv[8 + i] -= 1e-3f * ((float)g2[i] + (float)gl[i]);
            } else if (VK == 1) {   // GELU only, no split
#pragma unroll
                for (int i = 0; i < 16; ++i) hv[i] = gelu_scaled(hv[i]) + 0.3f;
            } else {                // split only
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = hv[i];
                bf16x8 g2, l2;
                split8(&v[0], gh, gl); split8(&v[8], g2, l2);
#pragma unroll
                for (int i = 0; i < 8; ++i) { hv[i] += 1e-3f * (float)gl[i]; hv[8 + i] += 1e-3f * (float)l2[i]; }
            }
        }
    }
    for (int i = 0; i < 16; ++i) sacc += acc0[i] + acc1[i] + hv[i];
    out[blockIdx.x * 512 + threadIdx.x] = sacc + (float)gh[0];
}
template <int MODE, int VK>
float run(float* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int r = 0; r < 2; ++r) { hipEventRecord(a); hipLaunchKernelGGL((k<MODE, VK>), dim3(256), dim3(512), 0, 0, out, 2000); hipEventRecord(b); hipEventSynchronize(b); }
    hipEventElapsedTime(&ms, a, b);
    return ms;
}
int main() {
    float* out; hipMalloc((void**)&out, 256 * 512 * 4);
    printf("24 MFMA per iter (waves 0-3)            : %.3f ms\n", run<1, 0>(out));
    printf("gelu_split8 x2 per iter (waves 4-7)      : %.3f ms   both: %.3f ms\n", run<2, 0>(out), run<3, 0>(out));
    printf("16 gelu_scaled per iter (no split)       : %.3f ms   both: %.3f ms\n", run<2, 1>(out), run<3, 1>(out));
    printf("split8 x2 per iter (no gelu)             : %.3f ms   both: %.3f ms\n", run<2, 2>(out), run<3, 2>(out));
    return 0;
}
