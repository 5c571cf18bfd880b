// Micro-benchmark of the FFN hidden-tile loop of k_main (perf experiments only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ffn_bench.hip -o tools/ffn_bench
//   tools/ffn_bench [waves_per_wg=8]
// Variants isolate MFMA, LDS-fragment and GELU cost of one 32-token tile (192 MFMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../phyloformer_amd/csrc/pf_device.hip.h"

using namespace pfk;

enum { V_FULL = 0, V_NOGELU = 1, V_NOLDS = 2, V_GELUONLY = 3, V_SPLIT_HALF = 4, V_PIPE = 5 };

template <int VAR, int THREADS, int STAGGER = 0>
__global__ void __launch_bounds__(THREADS, THREADS / 256) k_ffn(const bf16x8* wimg, const float* consts,
                                                                float* out, int tiles_per_wave) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const bf16x8* lw = reinterpret_cast<const bf16x8*>(smem);
    const float* lc = reinterpret_cast<const float*>(smem + FRAG_END * 16);
    {
        uint4* dst = reinterpret_cast<uint4*>(smem);
        const uint4* src = reinterpret_cast<const uint4*>(wimg);
        for (int i = threadIdx.x; i < FRAG_END; i += THREADS) dst[i] = src[i];
        float* dc = reinterpret_cast<float*>(smem + FRAG_END * 16);
        for (int i = threadIdx.x; i < CONST_LEN; i += THREADS) dc[i] = consts[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    if (STAGGER) {
        // de-phase the waves that share a SIMD: wave w sits on SIMD (w & 3), slot (w >> 2)
        const int slot = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
        for (int i = 0; i < slot * STAGGER; ++i) __builtin_amdgcn_s_sleep(8);   // 8 * 64 cycles each
    }
    float x[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) x[j] = 0.01f * (float)((lane * 7 + j * 3) % 97) - 0.4f;

    for (int tile = 0; tile < tiles_per_wave; ++tile) {
        bf16x8 xb_hi[4], xb_lo[4];
        {
            float xn[32];
            ln_pair(x, xn);
#pragma unroll
            for (int s = 0; s < 4; ++s) split8(&xn[8 * s], xb_hi[s], xb_lo[s]);
        }
        f32x16 oa[2];
        load_acc_bias(oa[0], lc + CONST_B2, h);
        load_acc_bias(oa[1], lc + CONST_B2 + 32, h);
        if (VAR == V_PIPE) {
            // software pipeline: GEMM1(T+1) is issued before GELU(T) so MFMA and VALU overlap in-wave
            f32x16 ha;
            load_acc_bias(ha, lc + CONST_B1, h);
            {
                const bf16x8* f1 = lw + FRAG_W1 + lane;
#pragma unroll
                for (int s = 0; s < 4; ++s) mfma3(ha, f1[s * 128], f1[s * 128 + 64], xb_hi[s], xb_lo[s]);
            }
#pragma unroll 1
            for (int T = 0; T < 8; ++T) {
                f32x16 hn;
                const int Tn = (T + 1) & 7;
                load_acc_bias(hn, lc + CONST_B1 + 32 * Tn, h);
                const bf16x8* f1 = lw + FRAG_W1 + (Tn * 4 * 2) * 64 + lane;
                const bf16x8* f2 = lw + FRAG_W2 + (2 * T * 2) * 64 + lane;
                float gv[16];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    mfma3(hn, f1[s * 128], f1[s * 128 + 64], xb_hi[s], xb_lo[s]);
#pragma unroll
                    for (int r = 4 * s; r < 4 * s + 4; ++r) gv[r] = gelu_as(ha[r]);
                }
                bf16x8 g_hi[2], g_lo[2];
                split8(&gv[0], g_hi[0], g_lo[0]);
                split8(&gv[8], g_hi[1], g_lo[1]);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int To = 0; To < 2; ++To) {
                        const bf16x8* f = f2 + (To * 32 + u * 2) * 64;
                        mfma3(oa[To], f[0], f[64], g_hi[u], g_lo[u]);
                    }
                ha = hn;
            }
        } else {
#pragma unroll 1
            for (int T = 0; T < 8; ++T) {
                f32x16 ha;
                load_acc_bias(ha, lc + CONST_B1 + 32 * T, h);
                const bf16x8* f1 = lw + FRAG_W1 + (T * 4 * 2) * 64 + lane;
                const bf16x8* f2 = lw + FRAG_W2 + (2 * T * 2) * 64 + lane;
                if (VAR != V_GELUONLY) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        if (VAR == V_NOLDS) mfma3(ha, xb_lo[s], xb_hi[(s + 1) & 3], xb_hi[s], xb_lo[s]);
                        else mfma3(ha, f1[s * 128], f1[s * 128 + 64], xb_hi[s], xb_lo[s]);
                    }
                }
                float gv[16];
                if (VAR == V_FULL || VAR == V_GELUONLY) {
                    bf16x8 g_hi[2], g_lo[2];
                    gelu_split8(ha, 0, g_hi[0], g_lo[0]);
                    gelu_split8(ha, 8, g_hi[1], g_lo[1]);
                    if (VAR == V_FULL) {
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int To = 0; To < 2; ++To) {
                                const bf16x8* f = f2 + (To * 32 + u * 2) * 64;
                                mfma3(oa[To], f[0], f[64], g_hi[u], g_lo[u]);
                            }
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) { oa[0][i] += (float)g_hi[0][i] + (float)g_lo[1][i]; oa[1][i] += (float)g_hi[1][i] + (float)g_lo[0][i]; }
                    }
                    continue;
                }
                if (VAR == V_SPLIT_HALF) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
#pragma unroll
                        for (int r = 8 * u; r < 8 * u + 8; ++r) gv[r] = gelu_as(ha[r]);
                        bf16x8 g_hi, g_lo;
                        split8(&gv[8 * u], g_hi, g_lo);
#pragma unroll
                        for (int To = 0; To < 2; ++To) {
                            const bf16x8* f = f2 + (To * 32 + u * 2) * 64;
                            mfma3(oa[To], f[0], f[64], g_hi, g_lo);
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) gv[r] = (VAR == V_NOGELU) ? ha[r] : gelu_as(ha[r]);
                bf16x8 g_hi[2], g_lo[2];
                split8(&gv[0], g_hi[0], g_lo[0]);
                split8(&gv[8], g_hi[1], g_lo[1]);
                if (VAR != V_GELUONLY) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int To = 0; To < 2; ++To) {
                            if (VAR == V_NOLDS) mfma3(oa[To], xb_lo[u], xb_hi[To], g_hi[u], g_lo[u]);
                            else {
                                const bf16x8* f = f2 + (To * 32 + u * 2) * 64;
                                mfma3(oa[To], f[0], f[64], g_hi[u], g_lo[u]);
                            }
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { oa[0][i] += (float)g_hi[0][i] + (float)g_lo[1][i]; oa[1][i] += (float)g_hi[1][i] + (float)g_lo[0][i]; }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) x[j] = 0.5f * x[j] + 1e-3f * oa[j >> 4][j & 15];
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) s += x[j];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int VAR, int THREADS, int STAGGER = 0>
void run(const char* name, const bf16x8* wimg, const float* consts, float* out, int tiles) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ffn<VAR, THREADS, STAGGER>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, MAIN_LDS_BYTES);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_ffn<VAR, THREADS, STAGGER>), dim3(256), dim3(THREADS), MAIN_LDS_BYTES, 0, wimg, consts, out, tiles);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipError_t e = hipGetLastError();
    const double waves = 256.0 * THREADS / 64;
    const double tile_total = waves * tiles;
    const double cyc_per_tile_simd = ms * 1e-3 * 2.4e9 * 1024.0 / tile_total;
    printf("%-14s threads %4d: %8.3f ms  %8.0f cycles/tile/SIMD (MFMA floor 6144)  %s\n", name, THREADS, ms,
           cyc_per_tile_simd, e == hipSuccess ? "" : hipGetErrorString(e));
}

int main(int argc, char** argv) {
    const int tiles = 64;
    std::vector<uint16_t> img((size_t)FRAG_END * 8);
    for (size_t i = 0; i < img.size(); ++i) img[i] = (uint16_t)(0x3c00 + (i * 2654435761u >> 20) % 512);  // ~[0.008, 0.03]
    std::vector<float> cst(CONST_LEN, 0.01f);
    bf16x8* d_img; float *d_c, *d_out;
    hipMalloc((void**)&d_img, img.size() * 2);
    hipMalloc((void**)&d_c, cst.size() * 4);
    hipMalloc((void**)&d_out, 256 * 1024 * 4);
    hipMemcpy(d_img, img.data(), img.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(d_c, cst.data(), cst.size() * 4, hipMemcpyHostToDevice);
    run<V_FULL, 512>("full", d_img, d_c, d_out, tiles);
    run<V_NOGELU, 512>("no-gelu", d_img, d_c, d_out, tiles);
    run<V_NOLDS, 512>("no-lds", d_img, d_c, d_out, tiles);
    run<V_GELUONLY, 512>("gelu-only", d_img, d_c, d_out, tiles);
    run<V_SPLIT_HALF, 512>("split-half", d_img, d_c, d_out, tiles);
    run<V_PIPE, 512>("pipelined", d_img, d_c, d_out, tiles);
    run<V_FULL, 256>("full", d_img, d_c, d_out, tiles);
    run<V_SPLIT_HALF, 256>("split-half", d_img, d_c, d_out, tiles);
    run<V_PIPE, 256>("pipelined", d_img, d_c, d_out, tiles);
    run<V_FULL, 512, 1>("full stagger1", d_img, d_c, d_out, tiles);
    run<V_FULL, 512, 2>("full stagger2", d_img, d_c, d_out, tiles);
    run<V_FULL, 512, 4>("full stagger4", d_img, d_c, d_out, tiles);
    run<V_PIPE, 512, 1>("pipe stagger1", d_img, d_c, d_out, tiles);
    run<V_FULL, 768, 1>("full stagger1", d_img, d_c, d_out, tiles);
    run<V_FULL, 768, 2>("full stagger2", d_img, d_c, d_out, tiles);
    run<V_FULL, 768>("full", d_img, d_c, d_out, tiles);
    run<V_SPLIT_HALF, 768>("split-half", d_img, d_c, d_out, tiles);
    run<V_PIPE, 768>("pipelined", d_img, d_c, d_out, tiles);
    return 0;
}
