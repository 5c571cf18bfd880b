// Does alternating roles (matrix phase / vector phase) between the two waves of a SIMD keep the overlap
// that fixed roles show?  One workgroup = 8 waves (2 per SIMD), groups g0 = waves 0-3, g1 = waves 4-7.
//   MODE 0: fixed roles, no barrier      MODE 1: fixed roles, s_barrier every iteration
//   MODE 2: roles swap every iteration (barrier between phases) — the ping-pong schedule
//   MODE 3: both groups do matrix then vector in lockstep (what k_main does today)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DPF_SPLIT_NODOT tools/overlap4_bench.hip -o tools/overlap4_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../phyloformer_amd/csrc/pf_device.hip.h"
using namespace pfk;

template <int MODE>
__global__ void __launch_bounds__(512, 2) k(float* out, int iters) {
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    f32x16 acc[2], hv;
    bf16x8 fb, gh, gl;
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; hv[i] = 0.01f * (threadIdx.x % 13) + 0.1f * i - 0.7f; }
    for (int i = 0; i < 8; ++i) { fb[i] = (__bf16)0.5f; gh[i] = fb[i]; gl[i] = fb[i]; }
    auto matrix = [&]() {
#pragma unroll
        for (int r = 0; r < 24; ++r) acc[r & 1] = PF_MFMA(fb, gh, acc[r & 1]);
    };
    auto vector = [&]() {
        bf16x8 g2, l2;
        gelu_split8(hv, 0, gh, gl);
        gelu_split8(hv, 8, g2, l2);
#pragma unroll
        for (int i = 0; i < 8; ++i) hv[i] += 1e-3f * ((float)gh[i] + (float)l2[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i) hv[8 + i] -= 1e-3f * ((float)g2[i] + (float)gl[i]);
    };
    if (MODE == 0 || MODE == 1) {
        for (int it = 0; it < iters; ++it) {
            if (grp == 0) matrix(); else vector();
            if (MODE == 1) __builtin_amdgcn_s_barrier();
        }
    } else if (MODE == 2) {
        // every wave does iters/2 matrix phases and iters/2 vector phases, groups in anti-phase
        for (int it = 0; it < iters; it += 2) {
            if (grp == 0) matrix(); else vector();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if (grp == 0) vector(); else matrix();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
    } else {
        for (int it = 0; it < iters; it += 2) { matrix(); vector(); }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i] + hv[i];
    out[blockIdx.x * 512 + threadIdx.x] = s + (float)gh[0] + (float)gl[1];
}
template <int MODE>
void run(const char* name, float* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int r = 0; r < 2; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, 2000);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    hipEventElapsedTime(&ms, a, b);
    // every mode executes, per SIMD, 2000 x (24 MFMA + 16 GELU values per lane) in total
    printf("%-44s %7.3f ms = %6.0f cycles per (24 MFMA + 16 values) per SIMD\n", name, ms, ms * 2.4e6 / 2000.0);
}
int main() {
    float* out; hipMalloc((void**)&out, 256 * 512 * 4);
    run<0>("fixed roles", out);
    run<1>("fixed roles + barrier per iteration", out);
    run<2>("roles swap every iteration (ping-pong)", out);
    run<3>("lockstep: both waves matrix then vector", out);
    return 0;
}
