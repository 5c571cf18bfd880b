// Micro-benchmark: FFN hidden loop with v_mfma_f32_16x16x32_bf16 (16-token tiles, 4 lanes per token).
// Layout-agnostic (synthetic operands): measures cycles per 32 tokens per SIMD at 2/3/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../phyloformer_amd/csrc/pf_device.hip.h"
using namespace pfk;
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ void mfma3_16(f32x4v& acc, const bf16x8& a_hi, const bf16x8& a_lo, const bf16x8& b_hi, const bf16x8& b_lo) {
    acc = MFMA16(a_lo, b_hi, acc); acc = MFMA16(a_hi, b_lo, acc); acc = MFMA16(a_hi, b_hi, acc);
}
__device__ __forceinline__ void gelu_split8v(const f32x4v& a, const f32x4v& b, bf16x8& hi, bf16x8& lo) {
    u32x4 h, l; unsigned p, q;
    gelu_split_pair(a[0], a[1], p, q); h[0] = p; l[0] = q;
    gelu_split_pair(a[2], a[3], p, q); h[1] = p; l[1] = q;
    gelu_split_pair(b[0], b[1], p, q); h[2] = p; l[2] = q;
    gelu_split_pair(b[2], b[3], p, q); h[3] = p; l[3] = q;
    hi = __builtin_bit_cast(bf16x8, h); lo = __builtin_bit_cast(bf16x8, l);
}
template <int THREADS, int STAGGER, int PIPE>
__global__ void __launch_bounds__(THREADS, THREADS / 256) k_ffn16(const bf16x8* wimg, float* out, int tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const bf16x8* lw = reinterpret_cast<const bf16x8*>(smem);
    { uint4* dst = reinterpret_cast<uint4*>(smem); const uint4* src = reinterpret_cast<const uint4*>(wimg);
      for (int i = threadIdx.x; i < FRAG_END; i += THREADS) dst[i] = src[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    if (STAGGER) { const int slot = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
        for (int i = 0; i < slot * STAGGER; ++i) __builtin_amdgcn_s_sleep(8); }
    float x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = 0.01f * (float)((lane * 7 + j * 3) % 97) - 0.4f;
    const bf16x8* w1 = lw + FRAG_W1 + lane;   // 16 hidden tiles x 2 ksteps x 2 (hi,lo) x 64 lanes = 4096 frags
    const bf16x8* w2 = lw + FRAG_W2 + lane;   // 4 out tiles x 8 ksteps x 2 x 64 = 4096 frags
    for (int tile = 0; tile < tiles; ++tile) {
        bf16x8 xb_hi[2], xb_lo[2];
        { float xn[16]; float s = 0.f;
#pragma unroll
          for (int j = 0; j < 16; ++j) s += x[j];
          s = row16_sum(s) * (1.f / 64.f); float v = 0.f;
#pragma unroll
          for (int j = 0; j < 16; ++j) { xn[j] = x[j] - s; v = fmaf(xn[j], xn[j], v); }
          const float r = 1.0f / sqrtf(row16_sum(v) * (1.f / 64.f) + 1e-5f);
#pragma unroll
          for (int j = 0; j < 16; ++j) xn[j] *= r;
          split8(&xn[0], xb_hi[0], xb_lo[0]); split8(&xn[8], xb_hi[1], xb_lo[1]); }
        f32x4v oa[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) oa[o] = f32x4v{0.01f, 0.02f, 0.03f, 0.04f};
        if (PIPE) {
            f32x4v ha0 = {0.1f, 0.1f, 0.1f, 0.1f}, ha1 = ha0;
#pragma unroll
            for (int s = 0; s < 2; ++s) { mfma3_16(ha0, w1[(0 * 2 + s) * 128], w1[(0 * 2 + s) * 128 + 64], xb_hi[s], xb_lo[s]);
                                          mfma3_16(ha1, w1[(1 * 2 + s) * 128], w1[(1 * 2 + s) * 128 + 64], xb_hi[s], xb_lo[s]); }
#pragma unroll 1
            for (int T = 0; T < 8; ++T) {   // T = pair of 16-row hidden tiles = one GEMM2 k-step
                const int Tn = (T + 1) & 7;
                f32x4v hn0 = {0.1f, 0.1f, 0.1f, 0.1f}, hn1 = hn0;
                bf16x8 g_hi, g_lo;
                gelu_split8v(ha0, ha1, g_hi, g_lo);
                const bf16x8* f1 = w1 + Tn * 512;
#pragma unroll
                for (int s = 0; s < 2; ++s) { mfma3_16(hn0, f1[(0 * 2 + s) * 128], f1[(0 * 2 + s) * 128 + 64], xb_hi[s], xb_lo[s]);
                                              mfma3_16(hn1, f1[(1 * 2 + s) * 128], f1[(1 * 2 + s) * 128 + 64], xb_hi[s], xb_lo[s]); }
                const bf16x8* f2 = w2 + T * 128;
#pragma unroll
                for (int o = 0; o < 4; ++o) mfma3_16(oa[o], f2[o * 1024], f2[o * 1024 + 64], g_hi, g_lo);
                ha0 = hn0; ha1 = hn1;
            }
        } else {
#pragma unroll 1
            for (int T = 0; T < 8; ++T) {
                f32x4v ha0 = {0.1f, 0.1f, 0.1f, 0.1f}, ha1 = ha0;
                const bf16x8* f1 = w1 + T * 512;
#pragma unroll
                for (int s = 0; s < 2; ++s) { mfma3_16(ha0, f1[(0 * 2 + s) * 128], f1[(0 * 2 + s) * 128 + 64], xb_hi[s], xb_lo[s]);
                                              mfma3_16(ha1, f1[(1 * 2 + s) * 128], f1[(1 * 2 + s) * 128 + 64], xb_hi[s], xb_lo[s]); }
                bf16x8 g_hi, g_lo;
                gelu_split8v(ha0, ha1, g_hi, g_lo);
                const bf16x8* f2 = w2 + T * 128;
#pragma unroll
                for (int o = 0; o < 4; ++o) mfma3_16(oa[o], f2[o * 1024], f2[o * 1024 + 64], g_hi, g_lo);
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = 0.5f * x[j] + 1e-3f * oa[j >> 2][j & 3];
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += x[j];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}
template <int THREADS, int STAGGER, int PIPE>
void run(const bf16x8* wimg, float* out) {
    const int tiles = 128;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ffn16<THREADS, STAGGER, PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, MAIN_LDS_BYTES);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int r = 0; r < 2; ++r) { hipEventRecord(a);
        hipLaunchKernelGGL((k_ffn16<THREADS, STAGGER, PIPE>), dim3(256), dim3(THREADS), MAIN_LDS_BYTES, 0, wimg, out, tiles);
        hipEventRecord(b); hipEventSynchronize(b); }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double tiles_total = 256.0 * THREADS / 64 * tiles;   // 16-token tiles
    printf("16x16 threads %4d stagger %d pipe %d: %7.3f ms  %7.0f cycles per 32 tokens per SIMD (MFMA floor 6144) %s\n", THREADS, STAGGER, PIPE, ms,
           ms * 1e-3 * 2.4e9 * 1024.0 / tiles_total * 2.0, hipGetErrorString(hipGetLastError()));
}
int main() {
    std::vector<uint16_t> img((size_t)FRAG_END * 8);
    for (size_t i = 0; i < img.size(); ++i) img[i] = (uint16_t)(0x3c00 + (i * 2654435761u >> 20) % 512);
    bf16x8* d_img; float* d_out;
    hipMalloc((void**)&d_img, img.size() * 2); hipMalloc((void**)&d_out, 256 * 1024 * 4);
    hipMemcpy(d_img, img.data(), img.size() * 2, hipMemcpyHostToDevice);
    run<512, 0, 0>(d_img, d_out); run<768, 0, 0>(d_img, d_out); run<1024, 0, 0>(d_img, d_out);
    run<512, 2, 0>(d_img, d_out); run<768, 2, 0>(d_img, d_out); run<1024, 2, 0>(d_img, d_out);
    run<512, 0, 1>(d_img, d_out); run<768, 0, 1>(d_img, d_out); run<1024, 0, 1>(d_img, d_out);
    run<768, 2, 1>(d_img, d_out); run<1024, 2, 1>(d_img, d_out); run<1024, 1, 1>(d_img, d_out);
    return 0;
}
