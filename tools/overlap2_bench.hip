// Which wave mixes overlap the matrix pipe with the GELU+split VALU stream on one SIMD?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/overlap2_bench.hip -o tools/overlap2_bench
// Workgroup = NM "matrix" waves per SIMD (waves 0 .. 4*NM-1) + NV "vector" waves per SIMD.
// Matrix wave: iterations of 24 MFMA (CHAINS independent accumulators, A fragments from LDS if LDSA).
// Vector wave: iterations of 2 x gelu_split8 (16 hidden values per lane), the real k_main stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../phyloformer_amd/csrc/pf_device.hip.h"
using namespace pfk;

template <int NM, int NV, int CHAINS, bool LDSA, bool RUN_M, bool RUN_V>
__global__ void __launch_bounds__(256 * (NM + NV)) k(float* out, int iters) {
    __shared__ bf16x8 frag[24 * 64];
    for (int i = threadIdx.x; i < 24 * 64; i += blockDim.x) {
        bf16x8 f;
        for (int j = 0; j < 8; ++j) f[j] = (__bf16)(0.001f * ((i + j) % 17));
        frag[i] = f;
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const bool is_m = wave < 4 * NM;
    f32x16 acc[2], hv;
    bf16x8 fb, gh, gl;
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; hv[i] = 0.01f * (threadIdx.x % 13) + 0.1f * i - 0.7f; }
    for (int i = 0; i < 8; ++i) { fb[i] = (__bf16)0.5f; gh[i] = fb[i]; gl[i] = fb[i]; }
    if (is_m) {
        if (RUN_M)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 24; ++r) {
                    const bf16x8 a = LDSA ? frag[r * 64 + lane] : fb;
                    acc[r % CHAINS] = PF_MFMA(a, fb, acc[r % CHAINS]);
                }
            }
    } else if (RUN_V) {
        for (int it = 0; it < iters; ++it) {
            bf16x8 g2, l2;
            gelu_split8(hv, 0, gh, gl);
            gelu_split8(hv, 8, g2, l2);
#pragma unroll
            for (int i = 0; i < 8; ++i) hv[i] += 1e-3f * ((float)gh[i] + (float)l2[i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) hv[8 + i] -= 1e-3f * ((float)g2[i] + (float)gl[i]);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i] + hv[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)gh[0];
}

template <int NM, int NV, int CHAINS, bool LDSA, bool RUN_M, bool RUN_V>
float run(float* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int r = 0; r < 2; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<NM, NV, CHAINS, LDSA, RUN_M, RUN_V>), dim3(256), dim3(256 * (NM + NV)), 0, 0, out, 2000);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    hipEventElapsedTime(&ms, a, b);
    return ms;
}
template <int NM, int NV, int CHAINS, bool LDSA>
void combo(const char* name, float* out) {
    const float m = run<NM, NV, CHAINS, LDSA, true, false>(out), v = run<NM, NV, CHAINS, LDSA, false, true>(out),
                both = run<NM, NV, CHAINS, LDSA, true, true>(out);
    // per SIMD and iteration: NM * 24 MFMA and NV * 16 GELU values per lane
    const double cyc = 2.4e6 / 2000.0;
    printf("%-34s matrix %7.3f ms (%5.1f cyc/MFMA)  vector %7.3f ms (%5.1f cyc/value)  both %7.3f ms  overlap %4.0f %%\n", name, m,
           m * cyc / (24.0 * NM), v, v * cyc / (16.0 * NV), both, 100.0 * (m + v - both) / (m < v ? m : v));
}
int main() {
    float* out; hipMalloc((void**)&out, 256 * 1024 * 4);
    combo<1, 1, 1, false>("1M(1 chain, reg A) + 1V", out);
    combo<1, 1, 2, false>("1M(2 chains, reg A) + 1V", out);
    combo<1, 1, 1, true>("1M(1 chain, LDS A) + 1V", out);
    combo<1, 1, 2, true>("1M(2 chains, LDS A) + 1V", out);
    combo<1, 2, 1, true>("1M(1 chain, LDS A) + 2V", out);
    combo<1, 2, 2, true>("1M(2 chains, LDS A) + 2V", out);
    combo<1, 2, 2, false>("1M(2 chains, reg A) + 2V", out);
    combo<1, 3, 2, true>("1M(2 chains, LDS A) + 3V", out);
    combo<2, 2, 2, true>("2M(2 chains, LDS A) + 2V", out);
    return 0;
}
