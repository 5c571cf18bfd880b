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

enum { V_FULL = 0, V_NOGELU = 1, V_NOLDS = 2, V_GELUONLY = 3, V_SPLIT_HALF = 4, V_PIPE = 5, V_DUAL = 6, V_DUAL_NOGELU = 7, V_DUAL_PRIO = 8, V_PIPE2 = 9, V_PIPE2_NOSGB = 10, V_PINGPONG = 11 };

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
        if (VAR == V_PINGPONG) {
            // barrier-phased wave pairs: waves 0-3 and 4-7 (one of each per SIMD) alternate
            // matrix-only and VALU-only phases in anti-phase
            const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
            if (grp == 1) __builtin_amdgcn_s_barrier();
            f32x16 ha;
            load_acc_bias(ha, lc + CONST_B1, h);
            {
                const bf16x8* f1 = lw + FRAG_W1 + lane;
#pragma unroll
                for (int s = 0; s < 4; ++s) mfma3(ha, f1[s * 128], f1[s * 128 + 64], xb_hi[s], xb_lo[s]);
            }
#pragma unroll 1
            for (int T = 0; T < 8; ++T) {
                __builtin_amdgcn_s_barrier();
                bf16x8 g_hi[2], g_lo[2];
                gelu_split8(ha, 0, g_hi[0], g_lo[0]);
                gelu_split8(ha, 8, g_hi[1], g_lo[1]);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                const int Tn = (T + 1) & 7;
                const bf16x8* f1 = lw + FRAG_W1 + (Tn * 4 * 2) * 64 + lane;
                const bf16x8* f2 = lw + FRAG_W2 + (2 * T * 2) * 64 + lane;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int To = 0; To < 2; ++To) {
                        const bf16x8* f = f2 + (To * 32 + u * 2) * 64;
                        mfma3(oa[To], f[0], f[64], g_hi[u], g_lo[u]);
                    }
                load_acc_bias(ha, lc + CONST_B1 + 32 * Tn, h);
#pragma unroll
                for (int s = 0; s < 4; ++s) mfma3(ha, f1[s * 128], f1[s * 128 + 64], xb_hi[s], xb_lo[s]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (grp == 0) __builtin_amdgcn_s_barrier();
        } else if (VAR == V_PIPE2 || VAR == V_PIPE2_NOSGB) {
            // full software pipeline: iteration T issues GEMM2(T-1) and GEMM1(T+1) on the matrix pipe
            // while the VALU evaluates GELU+split of tile T; sched_group_barrier pins the interleave
            f32x16 ha;
            load_acc_bias(ha, lc + CONST_B1, h);
            {
                const bf16x8* f1 = lw + FRAG_W1 + lane;
#pragma unroll
                for (int s = 0; s < 4; ++s) mfma3(ha, f1[s * 128], f1[s * 128 + 64], xb_hi[s], xb_lo[s]);
            }
            bf16x8 gp_hi[2], gp_lo[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { gp_hi[u] = zero_frag(); gp_lo[u] = zero_frag(); }
#pragma unroll 1
            for (int T = 0; T < 9; ++T) {
                const int Tn = (T + 1) & 7, Tp = (T + 7) & 7;
                f32x16 hn;
                load_acc_bias(hn, lc + CONST_B1 + 32 * Tn, h);
                const bf16x8* f1 = lw + FRAG_W1 + (Tn * 4 * 2) * 64 + lane;
                const bf16x8* f2 = lw + FRAG_W2 + (2 * Tp * 2) * 64 + lane;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int To = 0; To < 2; ++To) {
                        const bf16x8* f = f2 + (To * 32 + u * 2) * 64;
                        mfma3(oa[To], f[0], f[64], gp_hi[u], gp_lo[u]);
                    }
#pragma unroll
                for (int s = 0; s < 4; ++s) mfma3(hn, f1[s * 128], f1[s * 128 + 64], xb_hi[s], xb_lo[s]);
                bf16x8 g_hi[2], g_lo[2];
                gelu_split8(ha, 0, g_hi[0], g_lo[0]);
                gelu_split8(ha, 8, g_hi[1], g_lo[1]);
                if (VAR == V_PIPE2) {
#pragma unroll
                    for (int i = 0; i < 24; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS read (A fragments)
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);  // VALU
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) { gp_hi[u] = g_hi[u]; gp_lo[u] = g_lo[u]; }
                ha = hn;
            }
        } else if (VAR == V_PIPE) {
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
                if (VAR != V_GELUONLY && VAR < V_DUAL) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        if (VAR == V_NOLDS) mfma3(ha, xb_lo[s], xb_hi[(s + 1) & 3], xb_hi[s], xb_lo[s]);
                        else mfma3(ha, f1[s * 128], f1[s * 128 + 64], xb_hi[s], xb_lo[s]);
                    }
                }
                if (VAR == V_DUAL || VAR == V_DUAL_NOGELU || VAR == V_DUAL_PRIO) {
                    // two independent accumulator chains per GEMM, MFMAs interleaved at instruction level
                    f32x16 hb;
#pragma unroll
                    for (int r = 0; r < 16; ++r) hb[r] = 0.f;
                    bf16x8 fh[4], fl[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) { fh[s] = f1[s * 128]; fl[s] = f1[s * 128 + 64]; }
                    if (VAR == V_DUAL_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        ha = PF_MFMA(fl[s], xb_hi[s], ha);         hb = PF_MFMA(fl[s + 2], xb_hi[s + 2], hb);
                        ha = PF_MFMA(fh[s], xb_lo[s], ha);         hb = PF_MFMA(fh[s + 2], xb_lo[s + 2], hb);
                        ha = PF_MFMA(fh[s], xb_hi[s], ha);         hb = PF_MFMA(fh[s + 2], xb_hi[s + 2], hb);
                    }
                    if (VAR == V_DUAL_PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) ha[r] += hb[r];
                    bf16x8 g_hi[2], g_lo[2];
                    if (VAR == V_DUAL_NOGELU) {
                        float gv2[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) gv2[r] = ha[r];
                        split8(&gv2[0], g_hi[0], g_lo[0]); split8(&gv2[8], g_hi[1], g_lo[1]);
                    } else {
                        gelu_split8(ha, 0, g_hi[0], g_lo[0]);
                        gelu_split8(ha, 8, g_hi[1], g_lo[1]);
                    }
                    bf16x8 wh[4], wl2[4];
#pragma unroll
                    for (int st = 0; st < 4; ++st) { wh[st] = f2[((st & 1) * 32 + (st >> 1) * 2) * 64]; wl2[st] = f2[((st & 1) * 32 + (st >> 1) * 2) * 64 + 64]; }
                    if (VAR == V_DUAL_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        oa[0] = PF_MFMA(wl2[2 * u], g_hi[u], oa[0]);   oa[1] = PF_MFMA(wl2[2 * u + 1], g_hi[u], oa[1]);
                        oa[0] = PF_MFMA(wh[2 * u], g_lo[u], oa[0]);    oa[1] = PF_MFMA(wh[2 * u + 1], g_lo[u], oa[1]);
                        oa[0] = PF_MFMA(wh[2 * u], g_hi[u], oa[0]);    oa[1] = PF_MFMA(wh[2 * u + 1], g_hi[u], oa[1]);
                    }
                    if (VAR == V_DUAL_PRIO) __builtin_amdgcn_s_setprio(0);
                    continue;
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
    run<V_PINGPONG, 512>("pingpong", d_img, d_c, d_out, tiles);
    run<V_PIPE2, 512>("pipe2 sgb", d_img, d_c, d_out, tiles);
    run<V_PIPE2_NOSGB, 512>("pipe2 nosgb", d_img, d_c, d_out, tiles);
    run<V_PIPE2, 512, 2>("pipe2 sgb stg", d_img, d_c, d_out, tiles);
    run<V_PIPE2, 768>("pipe2 sgb", d_img, d_c, d_out, tiles);
    run<V_DUAL, 512>("dual", d_img, d_c, d_out, tiles);
    run<V_DUAL_NOGELU, 512>("dual no-gelu", d_img, d_c, d_out, tiles);
    run<V_DUAL_PRIO, 512>("dual prio", d_img, d_c, d_out, tiles);
    run<V_DUAL, 512, 2>("dual stagger2", d_img, d_c, d_out, tiles);
    run<V_DUAL, 768>("dual", d_img, d_c, d_out, tiles);
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
