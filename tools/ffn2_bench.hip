// Micro-benchmark of k_main's FFN hidden loop in the ONE-WAVE-PER-SIMD regime (round 2 experiments).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/ffn2_bench.hip -o tools/ffn2_bench
// A 256-thread workgroup (one wave per SIMD, up to 512 VGPRs) walks TWO 32-token tiles, skewed by half a
// hidden-tile step: while the matrix pipe runs GEMM2(X, T) + GEMM1(X, T+1) of one tile, the VALU
// evaluates GELU + bf16 split of the other tile's hidden tile.  Consecutive MFMAs alternate accumulators
// (GEMM1's chain / GEMM2's two chains) so no MFMA waits for its predecessor.
// Reports cycles per 32-token tile per SIMD (MFMA floor 192 x 32 = 6144).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../phyloformer_amd/csrc/pf_device.hip.h"

using namespace pfk;

enum { V_SKEW_SGB = 0, V_SKEW_NOSGB = 1, V_SKEW_NOGELU = 2, V_SKEW_MFMAONLY = 3, V_PLAIN = 4, V_SKEW_SGB6 = 5,
       V_SKEW_SGB8 = 6, V_SKEW_VALUONLY = 7 };

struct TileSt {
    f32x16 oa[2];
    f32x16 ha;
    bf16x8 xb_hi[4], xb_lo[4];
    bf16x8 g_hi[2], g_lo[2];
};

// matrix stream of tile X: GEMM2 of hidden tile T (operands X.g_*) and GEMM1 of hidden tile T+1 (into hn),
// MFMAs alternating between the two; vector stream of tile Y: GELU + split of Y.ha -> Y.g_*.
template <int VAR, bool G2, bool G1, bool GELU>
__device__ __forceinline__ void half_step(TileSt& X, TileSt& Y, lds_frag_t f1, lds_frag_t f2, lds_f32_t b1) {
    f32x16 hn;
    if (G1) load_acc_bias(hn, b1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = i >> 1, To = i & 1;
        bf16x8 ah, al, wh, wl;
        if (G1) { ah = f1[i * 128]; al = f1[i * 128 + 64]; }
        if (G2) { wh = f2[(To * 32 + u * 2) * 64]; wl = f2[(To * 32 + u * 2) * 64 + 64]; }
        if (G1) hn = PF_MFMA(al, X.xb_hi[i], hn);
        if (G2) X.oa[To] = PF_MFMA(wl, X.g_hi[u], X.oa[To]);
        if (G1) hn = PF_MFMA(ah, X.xb_lo[i], hn);
        if (G2) X.oa[To] = PF_MFMA(wh, X.g_lo[u], X.oa[To]);
        if (G1) hn = PF_MFMA(ah, X.xb_hi[i], hn);
        if (G2) X.oa[To] = PF_MFMA(wh, X.g_hi[u], X.oa[To]);
    }
    if (GELU && VAR != V_SKEW_MFMAONLY) {
        if (VAR == V_SKEW_NOGELU) {
            float gv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) gv[r] = Y.ha[r];
            split8(&gv[0], Y.g_hi[0], Y.g_lo[0]);
            split8(&gv[8], Y.g_hi[1], Y.g_lo[1]);
        } else {
            gelu_split8(Y.ha, 0, Y.g_hi[0], Y.g_lo[0]);
            gelu_split8(Y.ha, 8, Y.g_hi[1], Y.g_lo[1]);
        }
    }
    if (VAR == V_SKEW_SGB || VAR == V_SKEW_SGB6 || VAR == V_SKEW_SGB8) {
        constexpr int NV = VAR == V_SKEW_SGB6 ? 6 : (VAR == V_SKEW_SGB8 ? 8 : 7);
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);  // VALU
        }
    }
    if (G1) X.ha = hn;
}

template <int VAR>
__global__ void __launch_bounds__(256, 1) k_ffn2(const bf16x8* wimg, const float* consts, float* out, int pairs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_frag_t lw = (lds_frag_t)smem;
    lds_f32_t lc = (lds_f32_t)(smem + FRAG_END * 16);
    {
        uint4* dst = reinterpret_cast<uint4*>(smem);
        const uint4* src = reinterpret_cast<const uint4*>(wimg);
        for (int i = threadIdx.x; i < FRAG_END; i += 256) dst[i] = src[i];
        float* dc = reinterpret_cast<float*>(smem + FRAG_END * 16);
        for (int i = threadIdx.x; i < CONST_LEN; i += 256) dc[i] = consts[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    lds_frag_t w1p = lw + FRAG_W1 + lane;
    lds_frag_t w2p = lw + FRAG_W2 + lane;
    lds_f32_t lch = lc + 4 * h;
    float xa[32], xb[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) { xa[j] = 0.01f * (float)((lane * 7 + j * 3) % 97) - 0.4f; xb[j] = 0.013f * (float)((lane * 5 + j * 11) % 89) - 0.5f; }

    for (int it = 0; it < pairs; ++it) {
        TileSt A, B;
        {
            float xn[32];
            ln_pair(xa, xn);
#pragma unroll
            for (int s = 0; s < 4; ++s) split8(&xn[8 * s], A.xb_hi[s], A.xb_lo[s]);
            ln_pair(xb, xn);
#pragma unroll
            for (int s = 0; s < 4; ++s) split8(&xn[8 * s], B.xb_hi[s], B.xb_lo[s]);
        }
        load_acc_bias(A.oa[0], lch + CONST_B2); load_acc_bias(A.oa[1], lch + CONST_B2 + 32);
        load_acc_bias(B.oa[0], lch + CONST_B2); load_acc_bias(B.oa[1], lch + CONST_B2 + 32);
        if (VAR == V_PLAIN) {
            // the production loop body, one tile after the other (no interleave), 1 wave per SIMD
#pragma unroll 1
            for (int tl = 0; tl < 2; ++tl) {
                TileSt& X = tl ? B : A;
#pragma unroll 1
                for (int T = 0; T < 8; ++T) {
                    lds_frag_t f1 = w1p + T * 512;
                    lds_frag_t f2 = w2p + T * 256;
                    lds_f32_t bp = lch + CONST_B1 + 32 * T;
                    PF_OPAQUE(f1); PF_OPAQUE(f2); PF_OPAQUE(bp);
                    f32x16 ha;
                    load_acc_bias(ha, bp);
#pragma unroll
                    for (int s = 0; s < 4; ++s) { const bf16x8 fh = f1[s * 128], fl = f1[s * 128 + 64]; mfma3(ha, fh, fl, X.xb_hi[s], X.xb_lo[s]); }
                    bf16x8 g_hi[2], g_lo[2];
                    gelu_split8(ha, 0, g_hi[0], g_lo[0]);
                    gelu_split8(ha, 8, g_hi[1], g_lo[1]);
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const int u = st >> 1, To = st & 1;
                        { const bf16x8 fh = f2[(To * 32 + u * 2) * 64], fl = f2[(To * 32 + u * 2) * 64 + 64]; mfma3(X.oa[To], fh, fl, g_hi[u], g_lo[u]); }
                    }
                }
            }
        } else {
            // prologue: GEMM1(A, 0);  then  [B: G1(0) || gelu A(0)]
            {
                lds_frag_t f1 = w1p; lds_f32_t bp = lch + CONST_B1;
                PF_OPAQUE(f1); PF_OPAQUE(bp);
                half_step<VAR, false, true, false>(A, B, f1, f1, bp);
                half_step<VAR, false, true, true>(B, A, f1, f1, bp);
            }
#pragma unroll 1
            for (int T = 0; T < 7; ++T) {
                lds_frag_t f1 = w1p + (T + 1) * 512;
                lds_frag_t f2 = w2p + T * 256;
                lds_f32_t bp = lch + CONST_B1 + 32 * (T + 1);
                PF_OPAQUE(f1); PF_OPAQUE(f2); PF_OPAQUE(bp);
                half_step<VAR, true, true, true>(A, B, f1, f2, bp);   // A: G2(T) + G1(T+1) || gelu B(T)
                __builtin_amdgcn_sched_barrier(0);
                half_step<VAR, true, true, true>(B, A, f1, f2, bp);   // B: G2(T) + G1(T+1) || gelu A(T+1)
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                lds_frag_t f2 = w2p + 7 * 256; lds_f32_t bp = lch;
                PF_OPAQUE(f2); PF_OPAQUE(bp);
                half_step<VAR, true, false, true>(A, B, f2, f2, bp);  // A: G2(7) || gelu B(7)
                half_step<VAR, true, false, false>(B, A, f2, f2, bp); // B: G2(7)
            }
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            xa[j] = 0.5f * xa[j] + 1e-3f * A.oa[j >> 4][j & 15];
            xb[j] = 0.5f * xb[j] + 1e-3f * B.oa[j >> 4][j & 15];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) s += xa[j] + xb[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VAR>
void run(const char* name, const bf16x8* wimg, const float* consts, float* out, int pairs) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ffn2<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        MAIN_LDS_BYTES);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_ffn2<VAR>), dim3(256), dim3(256), MAIN_LDS_BYTES, 0, wimg, consts, out, pairs);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    hipError_t e = hipGetLastError();
    float chk = 0;
    hipMemcpy(&chk, out, 4, hipMemcpyDeviceToHost);
    const double tiles = 1024.0 * pairs * 2;                       // one wave per SIMD, two tiles per iteration
    const double cyc = best * 1e-3 * 2.4e9 * 1024.0 / tiles;      // at the nominal 2.4 GHz
    printf("%-16s %8.3f ms  %8.0f cycles/tile/SIMD @2.4GHz (MFMA floor 6144)  chk %.5f %s\n", name, best, cyc, chk,
           e == hipSuccess ? "" : hipGetErrorString(e));
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int pairs = 32;
    std::vector<uint16_t> img((size_t)FRAG_END * 8);
    for (size_t i = 0; i < img.size(); ++i) img[i] = (uint16_t)(0x3c00 + (i * 2654435761u >> 20) % 512);
    std::vector<float> cst(CONST_LEN, 0.01f);
    bf16x8* d_img; float *d_c, *d_out;
    hipMalloc((void**)&d_img, img.size() * 2);
    hipMalloc((void**)&d_c, cst.size() * 4);
    hipMalloc((void**)&d_out, 256 * 256 * 4);
    hipMemcpy(d_img, img.data(), img.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(d_c, cst.data(), cst.size() * 4, hipMemcpyHostToDevice);
    run<V_PLAIN>("plain 1w", d_img, d_c, d_out, pairs);
    run<V_SKEW_NOSGB>("skew nosgb", d_img, d_c, d_out, pairs);
    run<V_SKEW_SGB6>("skew sgb6", d_img, d_c, d_out, pairs);
    run<V_SKEW_SGB>("skew sgb7", d_img, d_c, d_out, pairs);
    run<V_SKEW_SGB8>("skew sgb8", d_img, d_c, d_out, pairs);
    run<V_SKEW_NOGELU>("skew split-only", d_img, d_c, d_out, pairs);
    run<V_SKEW_MFMAONLY>("skew mfma-only", d_img, d_c, d_out, pairs);
    return 0;
}
