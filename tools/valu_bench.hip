// VALU / MFMA issue-rate probe for gfx950 (perf experiments only).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip -o tools/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
template <int OP, int VK = 0>
__global__ void k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f + 1.f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 1.0001f, b1 = 0.9999f;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b0, b1};
    f32x16 acc0 = {0}, acc1 = {0};
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(float)threadIdx.x; fb[i] = (__bf16)1.0f; }
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {  // v_fma_f32, 8 independent chains
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));)
        } else if (OP == 1) {  // v_pk_fma_f32, 4 independent chains (8 values)
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                              "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));)
        } else if (OP == 2) {  // v_exp_f32
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                              "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (OP == 3) {  // v_rcp_f32
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                              "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (OP == 4) {  // v_cvt_pk_bf16_f32
            REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n"
                              "v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (OP == 5) {  // v_and_b32 / v_max_f32 mix (plain 32-bit ops)
            REP8(asm volatile("v_and_b32 %0, %0, %8\n v_max_f32 %1, %1, %9\n v_and_b32 %2, %2, %8\n v_max_f32 %3, %3, %9\n"
                              "v_and_b32 %4, %4, %8\n v_max_f32 %5, %5, %9\n v_and_b32 %6, %6, %8\n v_max_f32 %7, %7, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));)
        } else if (OP == 6) {  // v_mul_f32
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                              "v_mul_f32 %4, %4, %9\n v_mul_f32 %5, %5, %9\n v_mul_f32 %6, %6, %9\n v_mul_f32 %7, %7, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));)
        } else if (OP == 7) {  // MFMA 32x32x16 bf16, 2 independent accumulators, 8 per REP -> 64 per iteration
            REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
                 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
                 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
                 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);)
        } else if (OP == 8) {  // MFMA interleaved with 8 v_fma per MFMA (same wave)
            REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                 asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));)
        } else if (OP == 10 || OP == 11 || OP == 12) {
            // wave-specialised: waves 0-3 (one per SIMD) issue only MFMA, waves 4-7 only VALU
            const bool mf = __builtin_amdgcn_readfirstlane(threadIdx.x) < 256;
            if (OP == 12 || (OP == 10 && mf) ) {
                if (OP == 12 && !mf) {} else {
                REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                     acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);) }
            }
            if ((OP == 11 || (OP == 10 && !mf)) && VK == 1) {
                if (OP == 11 && mf) {} else {
                REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                                  "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n"
                                  : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));) }
            } else if ((OP == 11 || (OP == 10 && !mf)) && VK == 2) {
                if (OP == 11 && mf) {} else {
                REP8(asm volatile("v_exp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_exp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                                  "v_exp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_exp_f32 %6, %6\n v_rcp_f32 %7, %7\n v_exp_f32 %0, %0\n v_rcp_f32 %1, %1\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
            } else if ((OP == 11 || (OP == 10 && !mf)) && VK == 3) {
                if (OP == 11 && mf) {} else {
                REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_and_b32 %1, %1, %8\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_lshlrev_b32 %3, 16, %3\n"
                                  "v_cvt_pk_bf16_f32 %4, %4, %5\n v_and_b32 %5, %5, %8\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_lshlrev_b32 %7, 16, %7\n v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %9\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));) }
            } else if (OP == 11 || (OP == 10 && !mf)) {
                if (OP == 11 && mf) {} else {
                REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                  "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                                  "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));) }
            }
        } else if (OP == 9) {  // MFMA interleaved with 4 v_pk_fma per MFMA (same wave)
            REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                 asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));)
        }
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP, int VK = 0>
void run(const char* name, int threads, int per_iter, float* out) {
    const int iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int r = 0; r < 2; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<OP, VK>), dim3(256), dim3(threads), 0, 0, out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double waves_per_simd = threads / 256.0;
    const double cyc = ms * 1e-3 * 2.4e9;   // assumed clock
    printf("%-26s %d waves/SIMD: %7.3f ms  %6.2f cycles per wave-instr per SIMD\n", name, (int)waves_per_simd, ms,
           cyc / (iters * (double)per_iter * waves_per_simd));
}

int main() {
    float* out; hipMalloc((void**)&out, 256 * 1024 * 4);
    for (int th : {256, 512, 1024}) {
        run<0>("v_fma_f32", th, 64, out);
        run<1>("v_pk_fma_f32", th, 64, out);
        run<2>("v_exp_f32", th, 64, out);
        run<3>("v_rcp_f32", th, 64, out);
        run<4>("v_cvt_pk_bf16_f32", th, 64, out);
        run<5>("v_and/v_max", th, 64, out);
        run<6>("v_mul_f32", th, 64, out);
        run<7>("mfma_32x32x16_bf16", th, 64, out);
        run<8>("mfma + 8 v_fma (per mfma)", th, 8, out);
        run<9>("mfma + 4 v_pk_fma (per mfma)", th, 8, out);
    }
    // wave-specialised overlap test at 2 waves/SIMD: 16 MFMA per iteration on waves 0-3 and 80 v_fma on waves 4-7
    run<12>("only waves0-3: 16 mfma/iter", 512, 1, out);
    run<11>("only waves4-7: 80 v_fma/iter", 512, 1, out);
    run<10>("both concurrently", 512, 1, out);
    run<11, 1>("only waves4-7: 80 v_pk_fma/iter", 512, 1, out);
    run<10, 1>("mfma + pk_fma concurrently", 512, 1, out);
    run<11, 2>("only waves4-7: 80 exp/rcp per iter", 512, 1, out);
    run<10, 2>("mfma + exp/rcp concurrently", 512, 1, out);
    run<11, 3>("only waves4-7: 80 cvt/and/lshl per iter", 512, 1, out);
    run<10, 3>("mfma + cvt/logic concurrently", 512, 1, out);
    return 0;
}
