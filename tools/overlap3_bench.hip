// Per-instruction-class overlap with the matrix pipe: one MFMA wave + two VALU waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/overlap3_bench.hip -o tools/overlap3_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define PF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
// 8 independent registers, each instruction class applied to all of them, 8 rounds per iteration = 64 instrs
#define OP_FMA(i)   asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c0), "v"(c1));
#define OP_MUL(i)   asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c0));
#define OP_ADD(i)   asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c1));
#define OP_EXP(i)   asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
#define OP_CVT(i)   asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c0));
#define OP_DOT(i)   asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(r[i]) : "v"(c0), "v"(c1));
#define OP_MAX(i)   asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c0));
#define OP_AND(i)   asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(c0));
#define OP_LSH(i)   asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(r[i]));
#define OP_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
#define OP_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pc));
#define OP_PERM(i)  asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c0), "v"(c1));
#define OP_MOV(i)   asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(c0));
#define OP_FMAMK(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f000000" : "+v"(r[i]) : "v"(c0));
#define OP_RCP(i)   asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
#define OP_CVTI(i)  asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r[i]));
#define OP_MULABS(i) asm volatile("v_mul_f32 %0, |%0|, %1" : "+v"(r[i]) : "v"(c0));
#define OP_LDEXP(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c2));

enum { K_FMA, K_MUL, K_ADD, K_EXP, K_CVT, K_DOT, K_MAX, K_AND, K_LSH, K_PKMUL, K_PKFMA, K_PERM, K_MOV, K_FMAMK, K_RCP, K_CVTI, K_MULABS, K_LDEXP, K_N };
const char* NAMES[] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_exp_f32", "v_cvt_pk_bf16_f32", "v_dot2c_f32_bf16", "v_max_f32", "v_and_b32",
                       "v_lshlrev_b32", "v_pk_mul_f32", "v_pk_fma_f32", "v_perm_b32", "v_mov_b32", "v_fmaak_f32", "v_rcp_f32", "v_cvt_f32_i32", "v_mul |x|", "v_ldexp_f32"};

template <int KIND, int NV, bool RUN_M, bool RUN_V>
__global__ void __launch_bounds__(256 * (1 + NV)) k(float* out, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc0, acc1;
    bf16x8 fb;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    for (int i = 0; i < 8; ++i) fb[i] = (__bf16)0.5f;
    float r[8];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 p[8], pc = {0.999f, 1.001f};
    for (int i = 0; i < 8; ++i) { r[i] = 0.5f + 0.01f * (threadIdx.x % 7 + i); p[i] = f32x2{r[i], r[i]}; }
    float c0 = 0.999f, c1 = 1e-6f; int c2 = 0;
    asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2));
    if (wave < 4) {
        if (RUN_M)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 12; ++q) { acc0 = PF_MFMA(fb, fb, acc0); acc1 = PF_MFMA(fb, fb, acc1); }
            }
    } else if (RUN_V) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int round = 0; round < 8; ++round) {
                if (KIND == K_FMA) { REP8(OP_FMA) } else if (KIND == K_MUL) { REP8(OP_MUL) } else if (KIND == K_ADD) { REP8(OP_ADD) }
                else if (KIND == K_EXP) { REP8(OP_EXP) } else if (KIND == K_CVT) { REP8(OP_CVT) } else if (KIND == K_DOT) { REP8(OP_DOT) }
                else if (KIND == K_MAX) { REP8(OP_MAX) } else if (KIND == K_AND) { REP8(OP_AND) } else if (KIND == K_LSH) { REP8(OP_LSH) }
                else if (KIND == K_PKMUL) { REP8(OP_PKMUL) } else if (KIND == K_PKFMA) { REP8(OP_PKFMA) } else if (KIND == K_PERM) { REP8(OP_PERM) }
                else if (KIND == K_MOV) { REP8(OP_MOV) } else if (KIND == K_FMAMK) { REP8(OP_FMAMK) } else if (KIND == K_RCP) { REP8(OP_RCP) }
                else if (KIND == K_CVTI) { REP8(OP_CVTI) } else if (KIND == K_MULABS) { REP8(OP_MULABS) } else if (KIND == K_LDEXP) { REP8(OP_LDEXP) }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 8; ++i) s += r[i] + p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NV, bool RUN_M, bool RUN_V>
float run(float* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<KIND, NV, RUN_M, RUN_V>), dim3(256), dim3(256 * (1 + NV)), 0, 0, out, 2000);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    hipEventElapsedTime(&ms, a, b);
    return ms;
}
template <int KIND, int NV>
void combo(float* out) {
    const float m = run<KIND, NV, true, false>(out), v = run<KIND, NV, false, true>(out), both = run<KIND, NV, true, true>(out);
    const double cyc = 2.4e6 / 2000.0;
    printf("%-20s x%d waves: matrix %6.3f ms  vector %6.3f ms (%4.1f cyc/instr/SIMD)  both %6.3f ms  overlap %4.0f %%\n", NAMES[KIND], NV, m, v,
           v * cyc / (64.0 * NV), both, 100.0 * (m + v - both) / (m < v ? m : v));
}
template <int KIND> void both_nv(float* out) { combo<KIND, 1>(out); combo<KIND, 2>(out); }
int main() {
    float* out; hipMalloc((void**)&out, 256 * 1024 * 4);
    both_nv<K_FMA>(out); both_nv<K_MUL>(out); both_nv<K_ADD>(out); both_nv<K_EXP>(out); both_nv<K_CVT>(out); both_nv<K_DOT>(out);
    both_nv<K_MAX>(out); both_nv<K_AND>(out); both_nv<K_LSH>(out); both_nv<K_PKMUL>(out); both_nv<K_PKFMA>(out); both_nv<K_PERM>(out);
    both_nv<K_MOV>(out); both_nv<K_FMAMK>(out); both_nv<K_RCP>(out); both_nv<K_CVTI>(out); both_nv<K_MULABS>(out); both_nv<K_LDEXP>(out);
    return 0;
}
