// v_dot2c_f32_bf16 as "x - bf16_hi(x)" : correctness and rate probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstring>
__global__ void k_check(const float* in, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    float g0 = in[2 * i], g1 = in[2 * i + 1];
    unsigned h, m0 = 0x0000bf80u, m1 = 0xbf800000u;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(g0), "v"(g1));
    float r0 = g0, r1 = g1;
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(r0) : "s"(0x0000bf80u), "v"(h));
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(r1) : "s"(0xbf800000u), "v"(h));
    out[4 * i] = r0; out[4 * i + 1] = r1;
    out[4 * i + 2] = g0 - __uint_as_float(h << 16); out[4 * i + 3] = g1 - __uint_as_float(h & 0xffff0000u);
}
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void k_rate(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f + 1.f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned h = 0x3f803f80u, m = 0x0000bf80u;
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    f32x16 acc0 = {0}, acc1 = {0}; bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)1.f; fb[i] = (__bf16)0.5f; }
    const bool mf = __builtin_amdgcn_readfirstlane(threadIdx.x) < 256;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || (MODE == 2 && !mf)) {
            REP8(asm volatile("v_dot2c_f32_bf16 %0, %8, %9\n v_dot2c_f32_bf16 %1, %8, %9\n v_dot2c_f32_bf16 %2, %8, %9\n v_dot2c_f32_bf16 %3, %8, %9\n"
                              "v_dot2c_f32_bf16 %4, %8, %9\n v_dot2c_f32_bf16 %5, %8, %9\n v_dot2c_f32_bf16 %6, %8, %9\n v_dot2c_f32_bf16 %7, %8, %9\n"
                              "v_dot2c_f32_bf16 %0, %8, %9\n v_dot2c_f32_bf16 %1, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(h), "v"(m));)
        }
        if (MODE == 1 || (MODE == 2 && mf)) {
            REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);)
        }
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> float rate(float* out, int threads) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float ms;
    for (int r = 0; r < 2; ++r) { hipEventRecord(a); hipLaunchKernelGGL(k_rate<MODE>, dim3(256), dim3(threads), 0, 0, out, 2000); hipEventRecord(b); hipEventSynchronize(b); }
    hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    const int n = 1 << 16;
    std::vector<float> in(n), out(2 * n);
    for (int i = 0; i < n; ++i) in[i] = (float)((i * 2654435761u) % 100003) * 1e-3f * ((i & 1) ? 1.f : -0.37f) * std::pow(2.f, (i % 40) - 20);
    float *di, *dou; hipMalloc((void**)&di, n * 4); hipMalloc((void**)&dou, 2 * n * 4 + 256 * 1024 * 4);
    hipMemcpy(di, in.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(n / 2 / 256), dim3(256), 0, 0, di, dou, n);
    hipMemcpy(out.data(), dou, 2 * n * 4, hipMemcpyDeviceToHost);
    int bad = 0; 
    for (int i = 0; i < n / 2; ++i) for (int j = 0; j < 2; ++j) if (std::memcmp(&out[4 * i + j], &out[4 * i + 2 + j], 4) != 0) { if (bad < 5) printf("mismatch %g: dot2c %g vs sub %g\n", in[2 * i + j], out[4 * i + j], out[4 * i + 2 + j]); ++bad; }
    printf("dot2c residual == fp32 subtraction for %d of %d values\n", n - bad, n);
    printf("dot2c 80/iter: 1 wave/SIMD %.3f ms, 2 waves/SIMD %.3f ms; mfma 16/iter alone (512 thr, waves 0-3) %.3f; both %.3f\n",
           rate<0>(dou, 256), rate<0>(dou, 512), rate<1>(dou, 512), rate<2>(dou, 512));
    return 0;
}
