// gfx950 probe behind the round-6 switch of the split MFMA operands from bf16 to fp16:
//   A  does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs (A side, B side)?  The lo limb of every operand below
//      2^-3 is subnormal, so a flushing matrix core would cut such operands to 11 bits.
//   B  v_cvt_pk_f16_f32: round-to-nearest-even, subnormal results, overflow to inf - bit for bit the host's f2h()
//      (pf_host_prep.h), which packs the weights.
//   C  v_dot2c_f32_f16 as "g - fp16(g)": bit-identical to unpack + fp32 subtract (subnormal hi included)?  And the
//      read-after-write hazard that v_dot2c_f32_bf16 has (two wait states, not interlocked): present here too?
//   D  hi + lo reconstructs g to max(2^-22 |g|, 2^-25).
//   E  energy and rate of the matrix pipe alone: the three passes of mfma3 on realistic hi / lo operands, fp16
//      against bf16, socket joules per MFMA under sustained load (rsmi energy counter).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/f16_probe.hip -o tools/f16_probe -ldl ; tools/f16_probe [seconds]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../phyloformer_amd/csrc/pf_host_prep.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- A: subnormal inputs of the matrix core ------------------------------------------------------
__global__ void k_mfma_subnormal(float* out) {
    // D = A B with A[m][k] = a, B[k][n] = b for all entries: D = 16 a b
    const _Float16 sub = (_Float16)9.5367431640625e-07f;    // 2^-20: subnormal in fp16 (smallest normal 2^-14)
    const _Float16 big = (_Float16)1024.f;
    h8 A, B;
    f32x16 z = {0};
    for (int i = 0; i < 8; ++i) { A[i] = sub; B[i] = big; }
    f32x16 d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, z, 0, 0, 0);      // subnormal A: expect 16 * 2^-10
    for (int i = 0; i < 8; ++i) { A[i] = big; B[i] = sub; }
    f32x16 d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, z, 0, 0, 0);      // subnormal B
    for (int i = 0; i < 8; ++i) { A[i] = sub; B[i] = sub; }
    f32x16 d2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, z, 0, 0, 0);      // both: 16 * 2^-40
    const _Float16 tiny = __builtin_bit_cast(_Float16, (unsigned short)1);     // 2^-24, the smallest subnormal
    for (int i = 0; i < 8; ++i) { A[i] = tiny; B[i] = (_Float16)1.f; }
    f32x16 d3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, z, 0, 0, 0);      // 16 * 2^-24
    if (threadIdx.x == 0) { out[0] = d0[0]; out[1] = d1[0]; out[2] = d2[0]; out[3] = d3[0]; }
}

// ---- B / C / D -----------------------------------------------------------------------------------
template <int NOPS>
__global__ void k_split(const float* in, unsigned* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float g0 = in[2 * i], g1 = in[2 * i + 1];
    const h2 hv = {(_Float16)g0, (_Float16)g1};                                // v_cvt_pk_f16_f32
    const unsigned hb = __builtin_bit_cast(unsigned, hv);
    float r0 = g0, r1 = g1;
    // the dependent reader right behind the dot2c, NOPS wait states apart (NOPS < 0: none)
    if (NOPS >= 4) {
    } else if (NOPS < 0)
        asm volatile("v_dot2c_f32_f16 %0, %2, %4\n\tv_dot2c_f32_f16 %1, %3, %4\n\tv_mul_f32 %1, 1.0, %1\n\tv_mul_f32 %0, 1.0, %0"
                     : "+v"(r0), "+v"(r1) : "s"(0x0000bc00u), "s"(0xbc000000u), "v"(hb));
    else if (NOPS == 0)
        asm volatile("v_dot2c_f32_f16 %0, %2, %4\n\tv_dot2c_f32_f16 %1, %3, %4\n\ts_nop 0\n\tv_mul_f32 %1, 1.0, %1\n\tv_mul_f32 %0, 1.0, %0"
                     : "+v"(r0), "+v"(r1) : "s"(0x0000bc00u), "s"(0xbc000000u), "v"(hb));
    else if (NOPS == 1)
        asm volatile("v_dot2c_f32_f16 %0, %2, %4\n\tv_dot2c_f32_f16 %1, %3, %4\n\ts_nop 1\n\tv_mul_f32 %1, 1.0, %1\n\tv_mul_f32 %0, 1.0, %0"
                     : "+v"(r0), "+v"(r1) : "s"(0x0000bc00u), "s"(0xbc000000u), "v"(hb));
    else if (NOPS == 2)     // the production pattern of split8: the LAST dot2c, then s_nop 1, then the reader of ITS result
        asm volatile("v_dot2c_f32_f16 %0, %2, %4\n\tv_dot2c_f32_f16 %1, %3, %4\n\ts_nop 1\n\tv_mul_f32 %0, 1.0, %0\n\tv_mul_f32 %1, 1.0, %1"
                     : "+v"(r0), "+v"(r1) : "s"(0x0000bc00u), "s"(0xbc000000u), "v"(hb));
    else
        asm volatile("v_dot2c_f32_f16 %0, %2, %4\n\tv_dot2c_f32_f16 %1, %3, %4\n\ts_nop 3\n\tv_mul_f32 %1, 1.0, %1\n\tv_mul_f32 %0, 1.0, %0"
                     : "+v"(r0), "+v"(r1) : "s"(0x0000bc00u), "s"(0xbc000000u), "v"(hb));
    if (NOPS == 4) {        // v_fma_mix_f32 instead of v_dot2c: hi read as a half out of the packed register, reader back to back
        r0 = g0; r1 = g1;
        asm volatile("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_mul_f32 %1, 1.0, %1\n\tv_mul_f32 %0, 1.0, %0"
                     : "=&v"(r0), "=&v"(r1) : "v"(hb), "v"(g0), "v"(g1));
    }
    const float s0 = g0 - (float)hv[0], s1 = g1 - (float)hv[1];               // v_cvt_f32_f16 + v_sub_f32
    const h2 lv = {(_Float16)r0, (_Float16)r1};
    out[6 * i] = hb;
    unsigned lbits = __builtin_bit_cast(unsigned, lv);
    if (NOPS == 5) {        // the residual AND its conversion in one instruction per value: v_fma_mixlo_f16 / v_fma_mixhi_f16
        unsigned l;
        asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                     : "=&v"(l) : "v"(hb), "v"(g0), "v"(g1));
        lbits = l;
        r0 = g0 - (float)hv[0]; r1 = g1 - (float)hv[1];
    }
    out[6 * i + 1] = lbits;
    out[6 * i + 2] = __float_as_uint(r0); out[6 * i + 3] = __float_as_uint(r1);
    out[6 * i + 4] = __float_as_uint(s0); out[6 * i + 5] = __float_as_uint(s1);
}

// ---- E: the matrix pipe alone ---------------------------------------------------------------------
template <bool F16>
__global__ void __launch_bounds__(512, 2) k_mfma_power(const u32x4* frags, float* out, int iters) {
    // per wave: 4 A pairs (hi, lo) of "weights", 4 B pairs of "activations"; the production pass order of mfma3
    const int lane = threadIdx.x & 63;
    u32x4 a_hi[4], a_lo[4], b_hi[4], b_lo[4];
    for (int s = 0; s < 4; ++s) {
        a_hi[s] = frags[(0 * 4 + s) * 64 + lane]; a_lo[s] = frags[(1 * 4 + s) * 64 + lane];
        b_hi[s] = frags[(2 * 4 + s) * 64 + lane]; b_lo[s] = frags[(3 * 4 + s) * 64 + lane];
    }
    f32x16 acc0 = {0}, acc1 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (F16) {
                const h8 ah = __builtin_bit_cast(h8, a_hi[s]), al = __builtin_bit_cast(h8, a_lo[s]);
                const h8 bh = __builtin_bit_cast(h8, b_hi[s]), bl = __builtin_bit_cast(h8, b_lo[s]);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1, 0, 0, 0);
            } else {
                const b8 ah = __builtin_bit_cast(b8, a_hi[s]), al = __builtin_bit_cast(b8, a_lo[s]);
                const b8 bh = __builtin_bit_cast(b8, b_hi[s]), bl = __builtin_bit_cast(b8, b_lo[s]);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc1, 0, 0, 0);
            }
        }
        // keep the accumulators bounded without touching the operand stream
        if ((it & 63) == 63) { acc0 *= 1e-3f; acc1 *= 1e-3f; }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

struct Rsmi {
    void* lib = nullptr;
    int (*energy)(uint32_t, uint64_t*, float*, uint64_t*) = nullptr;
    bool ok = false;
    Rsmi() {
        lib = dlopen("/opt/rocm/lib/librocm_smi64.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("librocm_smi64.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) { printf("energy: cannot load librocm_smi64: %s\n", dlerror()); return; }
        auto init = reinterpret_cast<int (*)(uint64_t)>(dlsym(lib, "rsmi_init"));
        energy = reinterpret_cast<int (*)(uint32_t, uint64_t*, float*, uint64_t*)>(dlsym(lib, "rsmi_dev_energy_count_get"));
        if (!init || !energy || init(0) != 0) { printf("energy: rsmi_init failed\n"); return; }
        ok = true;
    }
    double joules() {
        uint64_t c = 0, ts = 0; float res = 0;
        if (!ok || energy(0, &c, &res, &ts) != 0) return -1;
        return (double)c * res * 1e-6;
    }
};

template <bool F16>
void power_run(Rsmi& smi, const char* name, double seconds, const u32x4* d_frags, float* d_out, int grid = 256) {
    const int iters = 4096;
    auto launch = [&] { hipLaunchKernelGGL((k_mfma_power<F16>), dim3(grid), dim3(512), 0, 0, d_frags, d_out, iters); };
    launch(); hipDeviceSynchronize();
    const double j0 = smi.joules();
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0; double dt = 0;
    do { for (int i = 0; i < 4; ++i) launch(); hipDeviceSynchronize(); n += 4;
         dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } while (dt < seconds);
    const double j1 = smi.joules();
    const double mfmas = (double)grid * 8 * iters * 24 * n;             // wave-level MFMA instructions
    printf("power %-44s grid %3d %7.1f W  %7.3f nJ/MFMA  %6.1f TFLOP/s  %6.2f ns per MFMA and SIMD  (%ld launches, %.2f s)\n", name, grid,
           (j1 - j0) / dt, (j1 - j0) / mfmas * 1e9, mfmas * 32768.0 / dt * 1e-12, dt * 1e9 / ((double)iters * 24 * 2 * n), n, dt);
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    using namespace pfhost;
    {   // A
        float* d; hipMalloc((void**)&d, 64);
        hipLaunchKernelGGL(k_mfma_subnormal, dim3(1), dim3(64), 0, 0, d);
        float o[4]; hipMemcpy(o, d, 16, hipMemcpyDeviceToHost);
        printf("A  MFMA f16 subnormal A operand: D = %.9g (kept: %.9g)  -> %s\n", o[0], 16 * std::ldexp(1.0, -10), o[0] == 16 * std::ldexp(1.f, -10) ? "KEPT" : "FLUSHED");
        printf("A  MFMA f16 subnormal B operand: D = %.9g (kept: %.9g)  -> %s\n", o[1], 16 * std::ldexp(1.0, -10), o[1] == 16 * std::ldexp(1.f, -10) ? "KEPT" : "FLUSHED");
        printf("A  both subnormal: D = %.9g (exact %.9g);  smallest subnormal x 1: D = %.9g (exact %.9g)\n", o[2], 16 * std::ldexp(1.0, -40), o[3], 16 * std::ldexp(1.0, -24));
    }
    {   // B C D
        const int n = 1 << 20;
        std::vector<float> in(n);
        std::mt19937_64 g(7);
        for (int i = 0; i < n; ++i) {
            // magnitudes 2^-32 .. 2^17 (past both ends of fp16), random significands, both signs; a few exact ties
            const int e = (int)(g() % 50) - 32;
            float m = 1.f + (float)(g() & 0x7fffff) / 8388608.f;
            if (i % 97 == 0) m = 1.f + (float)((g() % 2048) * 2 + 1) / 4096.f;      // exact fp16 rounding ties
            in[i] = std::ldexp(m, e) * ((g() & 1) ? 1.f : -1.f);
        }
        in[0] = 65504.f; in[1] = 65519.9f; in[2] = 65520.f; in[3] = -70000.f; in[4] = 0.f; in[5] = -0.f;
        float* di; unsigned* dou;
        hipMalloc((void**)&di, n * 4); hipMalloc((void**)&dou, (size_t)3 * n * 4);
        hipMemcpy(di, in.data(), n * 4, hipMemcpyHostToDevice);
        std::vector<unsigned> out((size_t)3 * n);
        for (int nops = -1; nops <= 5; ++nops) {
            if (nops < 0) hipLaunchKernelGGL(k_split<-1>, dim3(n / 2 / 256), dim3(256), 0, 0, di, dou, n);
            else if (nops == 0) hipLaunchKernelGGL(k_split<0>, dim3(n / 2 / 256), dim3(256), 0, 0, di, dou, n);
            else if (nops == 1) hipLaunchKernelGGL(k_split<1>, dim3(n / 2 / 256), dim3(256), 0, 0, di, dou, n);
            else if (nops == 2) hipLaunchKernelGGL(k_split<2>, dim3(n / 2 / 256), dim3(256), 0, 0, di, dou, n);
            else if (nops == 3) hipLaunchKernelGGL(k_split<3>, dim3(n / 2 / 256), dim3(256), 0, 0, di, dou, n);
            else if (nops == 4) hipLaunchKernelGGL(k_split<4>, dim3(n / 2 / 256), dim3(256), 0, 0, di, dou, n);
            else hipLaunchKernelGGL(k_split<5>, dim3(n / 2 / 256), dim3(256), 0, 0, di, dou, n);
            hipMemcpy(out.data(), dou, out.size() * 4, hipMemcpyDeviceToHost);
            long bad_cvt = 0, bad_dot = 0, bad_rec = 0, sub_hi = 0, finite = 0;
            double worst_rel = 0, worst_abs = 0;
            for (int i = 0; i < n / 2; ++i)
                for (int j = 0; j < 2; ++j) {
                    const float gv = in[2 * i + j];
                    const uint16_t hb = (uint16_t)(out[6 * i] >> (16 * j)), lb = (uint16_t)(out[6 * i + 1] >> (16 * j));
                    if (hb != f2h(gv)) { if (bad_cvt < 4) printf("   cvt mismatch g=%a device %04x host %04x\n", gv, hb, f2h(gv)); ++bad_cvt; }
                    const float hf = h2f(hb);
                    if (!std::isfinite(hf)) continue;
                    ++finite;
                    if ((hb & 0x7c00) == 0 && (hb & 0x3ff)) ++sub_hi;
                    if (out[6 * i + 2 + j] != out[6 * i + 4 + j]) {
                        if (bad_dot < 4) printf("   dot2c mismatch g=%a hi=%04x: dot2c %a  sub %a\n", gv, hb, __builtin_bit_cast(float, out[6 * i + 2 + j]), __builtin_bit_cast(float, out[6 * i + 4 + j]));
                        ++bad_dot;
                    }
                    if (nops == 5 && lb != f2h(gv - hf)) { if (bad_dot < 4) printf("   mixlo/hi lo limb %04x != RNE(g - hi) %04x for g=%a\n", lb, f2h(gv - hf), gv); ++bad_dot; }
                    const double rec = (double)hf + (double)h2f(lb), err = std::fabs((double)gv - rec);
                    const double bound = std::fmax(std::ldexp(std::fabs((double)gv), -22) * 1.0001, std::ldexp(1.0, -25));
                    if (err > bound) { if (bad_rec < 4) printf("   hi+lo off: g=%a err %.3g bound %.3g\n", gv, err, bound); ++bad_rec; }
                    if (std::fabs(gv) >= 0.125) worst_rel = std::fmax(worst_rel, err / std::fabs((double)gv)); else worst_abs = std::fmax(worst_abs, err);
                }
            printf("%s (reader %s): cvt_pk vs host RNE mismatches %ld / %d;  dot2c != sub: %ld;  hi+lo outside max(2^-22|g|, 2^-25): %ld "
                   "(worst rel %.3g = 2^%.2f for |g| >= 1/8, worst abs %.3g = 2^%.2f below; %ld subnormal hi among %ld finite)\n",
                   nops < 0 ? "B/C/D" : "  C  ", nops < 0 ? "back to back" : nops == 0 ? "after s_nop 0" : nops == 1 ? "after s_nop 1, last result first" : nops == 2 ? "after s_nop 1, as split8 reads" : nops == 3 ? "after s_nop 3" : nops == 4 ? "v_fma_mix_f32 instead of v_dot2c, back to back" : "lo limb by v_fma_mixlo/mixhi_f16", bad_cvt, n, bad_dot, bad_rec,
                   worst_rel, std::log2(worst_rel), worst_abs, std::log2(worst_abs), sub_hi, finite);
        }
    }
    {   // E
        Rsmi smi;
        if (!smi.ok) return 0;
        std::mt19937_64 g(11);
        std::normal_distribution<float> nd(0.f, 1.f);
        std::vector<uint16_t> fh((size_t)16 * 64 * 8), fb(fh.size());
        for (int grp = 0; grp < 2; ++grp)                    // 0: weights N(0, 0.15), 1: activations N(0, 1)
            for (int s = 0; s < 4; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 8; ++i) {
                        const float v = nd(g) * (grp == 0 ? 0.15f : 1.f);
                        const size_t hi_at = (((size_t)(2 * grp) * 4 + s) * 64 + lane) * 8 + i, lo_at = (((size_t)(2 * grp + 1) * 4 + s) * 64 + lane) * 8 + i;
                        const uint16_t hh = f2h(v), hb = f2bf(v);
                        fh[hi_at] = hh; fh[lo_at] = f2h(v - h2f(hh));
                        fb[hi_at] = hb; fb[lo_at] = f2bf(v - bf2f(hb));
                    }
        u32x4 *dh, *db; float* dout;
        hipMalloc((void**)&dh, fh.size() * 2); hipMalloc((void**)&db, fb.size() * 2); hipMalloc((void**)&dout, 256 * 512 * 4);
        hipMemcpy(dh, fh.data(), fh.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(db, fb.data(), fb.size() * 2, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) {                  // alternating, so that a warming box shows
            power_run<false>(smi, "bf16 hi/lo, 3 passes", seconds, db, dout);
            power_run<true>(smi, "fp16 hi/lo, 3 passes", seconds, dh, dout);
        }
        power_run<true>(smi, "fp16 MFMA on the bf16 bit patterns", seconds, db, dout);
        power_run<false>(smi, "bf16 MFMA on the fp16 bit patterns", seconds, dh, dout);
        // Do subnormal lo limbs cost the matrix pipe time or energy?  The same operands times 2^10 (every limb normal),
        // full chip (power-capped) and on 16 CUs (far below the cap: the pipe's own rate).
        std::vector<uint16_t> fs(fh.size());
        {
            std::mt19937_64 g2(11);
            std::normal_distribution<float> nd2(0.f, 1.f);
            for (int grp = 0; grp < 2; ++grp)
                for (int s = 0; s < 4; ++s)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int i = 0; i < 8; ++i) {
                            const float v = nd2(g2) * (grp == 0 ? 0.15f : 1.f) * 1024.f;
                            const size_t hi_at = (((size_t)(2 * grp) * 4 + s) * 64 + lane) * 8 + i, lo_at = (((size_t)(2 * grp + 1) * 4 + s) * 64 + lane) * 8 + i;
                            const uint16_t hh = f2h(v);
                            fs[hi_at] = hh; fs[lo_at] = f2h(v - h2f(hh));
                        }
        }
        u32x4* ds; hipMalloc((void**)&ds, fs.size() * 2);
        hipMemcpy(ds, fs.data(), fs.size() * 2, hipMemcpyHostToDevice);
        long nsub = 0, nsub_s = 0;
        for (size_t i = 0; i < fh.size(); ++i) { nsub += (fh[i] & 0x7c00) == 0 && (fh[i] & 0x3ff); nsub_s += (fs[i] & 0x7c00) == 0 && (fs[i] & 0x3ff); }
        printf("subnormal operand values: %ld of %zu as packed, %ld of %zu after scaling by 2^10\n", nsub, fh.size(), nsub_s, fs.size());
        for (int rep = 0; rep < 2; ++rep) {
            power_run<true>(smi, "fp16 hi/lo as packed (lo mostly subnormal)", seconds, dh, dout);
            power_run<true>(smi, "fp16 hi/lo x 2^10 (all limbs normal)", seconds, ds, dout);
        }
        power_run<true>(smi, "fp16 hi/lo as packed", seconds / 2, dh, dout, 16);
        power_run<true>(smi, "fp16 hi/lo x 2^10", seconds / 2, ds, dout, 16);
        power_run<false>(smi, "bf16 hi/lo", seconds / 2, db, dout, 16);
    }
    return 0;
}
