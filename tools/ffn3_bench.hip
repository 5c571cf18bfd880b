// Micro-benchmark + bit-exactness check of the hand-scheduled FFN hidden loop (tools/gen_hidden_asm.py).
//   python tools/gen_hidden_asm.py 6 base > /tmp/hid.inc       (both streams; modes: base nodot nosplit poly3 noexp movonly mfmaonly)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DPF_SH_CONST=0xbf800000u \
//         -DPF_HID_INC='"/tmp/hid.inc"' tools/ffn3_bench.hip -o tools/ffn3_bench_base      (nodot: -DPF_SH_CONST=0xffff0000u)
//   tools/run_ffn3.sh runs the prebuilt tools/ffn3_bench_<mode> binaries on the GPU box
// Variants: plain C++ loop (the production body of round 1) with 1 or 2 waves per SIMD, the two-tile asm stream
// (one wave per SIMD) and the one-tile software-pipelined asm stream (two waves per SIMD).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <chrono>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../phyloformer_amd/csrc/pf_device.hip.h"
#include "../phyloformer_amd/csrc/pf_host_prep.h"
#include PF_HID_INC

using namespace pfk;
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ u32x16 pack4(const frag_t (&f)[4]) {
    u32x16 r;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const u32x4 q = __builtin_bit_cast(u32x4, f[s]);
#pragma unroll
        for (int i = 0; i < 4; ++i) r[4 * s + i] = q[i];
    }
    return r;
}

static unsigned long long* g_cyc = nullptr;
static double mean_cycles() {
    std::vector<unsigned long long> c(256);
    hipMemcpy(c.data(), g_cyc, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : c) s += (double)v;
    return s / 256.0;
}

enum { M_PLAIN = 0, M_HID2 = 1, M_HID1 = 2 };

template <int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS, THREADS / 256) k_hid(const frag_t* wimg, const float* consts, float* out,
                                                               float* dump, int iters, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_frag_t lw = (lds_frag_t)smem;
    lds_f32_t lc = (lds_f32_t)(smem + FRAG_END * 16);
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
    lds_frag_t w1p = lw + FRAG_W1 + lane;
    lds_frag_t w2p = lw + FRAG_W2 + lane;
    lds_f32_t lch = lc + 4 * h;
    constexpr int NT = (MODE == M_HID1) ? 1 : 2;       // tiles per iteration
    float x[NT][32];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 32; ++j)
            x[t][j] = 0.01f * (float)((lane * (7 - 2 * t) + j * (3 + 8 * t) + 13 * (threadIdx.x >> 6)) % 97) - 0.4f - 0.1f * t;

    const unsigned long long tc0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        frag_t xh[NT][4], xl[NT][4];
        f32x16 oa[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float xn[32];
            ln_pair(x[t], xn);
#pragma unroll
            for (int s = 0; s < 4; ++s) split8(&xn[8 * s], xh[t][s], xl[t][s]);
            load_acc_bias(oa[t][0], lch + CONST_B2);
            load_acc_bias(oa[t][1], lch + CONST_B2 + 32);
        }
        if (MODE == M_PLAIN) {
#pragma unroll 1
            for (int t = 0; t < NT; ++t) {
#pragma unroll 1
                for (int T = 0; T < 8; ++T) {
                    lds_frag_t f1 = w1p + T * 512;
                    lds_frag_t f2 = w2p + T * 256;
                    lds_f32_t bp = lch + CONST_B1 + 32 * T;
                    PF_OPAQUE(f1); PF_OPAQUE(f2); PF_OPAQUE(bp);
                    f32x16 ha;
                    load_acc_bias(ha, bp);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const frag_t fh = f1[s * 128], fl = f1[s * 128 + 64];
                        mfma3(ha, fh, fl, xh[t][s], xl[t][s]);
                    }
                    frag_t g_hi[2], g_lo[2];
                    gelu_split8(ha, 0, g_hi[0], g_lo[0]);
                    gelu_split8(ha, 8, g_hi[1], g_lo[1]);
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const int u = st >> 1, To = st & 1;
                        const frag_t fh = f2[(To * 32 + u * 2) * 64], fl = f2[(To * 32 + u * 2) * 64 + 64];
                        mfma3(oa[t][To], fh, fl, g_hi[u], g_lo[u]);
                    }
                }
            }
        } else {
            int aw1 = lane * 16, aw2 = FRAG_W2 * 16 + lane * 16, ab = FRAG_END * 16 + (CONST_B1 + 4 * h) * 4;
            int tcount;
            const float c4 = 0.0136151873f;
            if constexpr (MODE == M_HID2) {
                const u32x16 xAh = pack4(xh[0]), xAl = pack4(xl[0]), xBh = pack4(xh[NT - 1]), xBl = pack4(xl[NT - 1]);
                f32x16 o0, o1, o2, o3;
                asm volatile(PF_HID2_ASM
                             : PF_HID2_OUT0_A(o0), PF_HID2_OUT1_A(o1), PF_HID2_OUT0_B(o2), PF_HID2_OUT1_B(o3),
                               PF_HID2_INIT0_A(oa[0][0]), PF_HID2_INIT1_A(oa[0][1]), PF_HID2_INIT0_B(oa[NT - 1][0]),
                               PF_HID2_INIT1_B(oa[NT - 1][1]), PF_HID2_AW1(aw1), PF_HID2_AW2(aw2), PF_HID2_AB(ab),
                               [t] "=&s"(tcount)
                             : PF_HID2_XH_A(xAh), PF_HID2_XL_A(xAl), PF_HID2_XH_B(xBh), PF_HID2_XL_B(xBl), PF_HID2_C4(c4),
                               [c5] "s"(-0.00107098569f), [c3] "s"(-0.084594565f), [c2] "s"(-0.637684925f),
                               [c1] "s"(-1.35494915f), [c0] "s"(-0.00003762f), [sl] "s"(0x0000bf80u), [sh] "s"(PF_SH_CONST)
                             : PF_HID2_CLOBBERS, "scc");
                oa[0][0] = o0; oa[0][1] = o1; oa[NT - 1][0] = o2; oa[NT - 1][1] = o3;
            } else {
                const u32x16 xAh = pack4(xh[0]), xAl = pack4(xl[0]);
                f32x16 o0, o1;
                asm volatile(PF_HID1_ASM
                             : PF_HID1_OUT0_A(o0), PF_HID1_OUT1_A(o1), PF_HID1_INIT0_A(oa[0][0]), PF_HID1_INIT1_A(oa[0][1]),
                               PF_HID1_AW1(aw1), PF_HID1_AW2(aw2), PF_HID1_AB(ab), [t] "=&s"(tcount)
                             : PF_HID1_XH_A(xAh), PF_HID1_XL_A(xAl), PF_HID1_C4(c4),
                               [c5] "s"(-0.00107098569f), [c3] "s"(-0.084594565f), [c2] "s"(-0.637684925f),
                               [c1] "s"(-1.35494915f), [c0] "s"(-0.00003762f), [sl] "s"(0x0000bf80u), [sh] "s"(PF_SH_CONST)
                             : PF_HID1_CLOBBERS, "scc");
                oa[0][0] = o0; oa[0][1] = o1;
            }
        }
        if (it == 0 && blockIdx.x == 0 && dump) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 32; ++j) dump[((size_t)t * THREADS + threadIdx.x) * 32 + j] = oa[t][j >> 4][j & 15];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 32; ++j) x[t][j] = 0.01f * (float)((lane + j * 5) % 61) + 1e-2f * oa[t][j >> 4][j & 15];   // nothing but oa lives across the loop
    }
    const unsigned long long tc1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 32; ++j) s += x[t][j];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = tc1 - tc0;
}

// Lean harness for the one-tile stream: operands are loaded from / stored to (L2-resident) global memory
// so that the compiler-generated code around the asm needs only a handful of registers and the kernel
// fits 2 waves per SIMD (176 pinned VGPRs + 64 pinned AGPRs).
template <int THREADS>
__global__ void __launch_bounds__(THREADS, THREADS / 256) k_hid1_lean(const frag_t* wimg, const float* consts,
                                                                     const u32x16* opnd, float* out, int iters,
                                                                     unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
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
    const u32x16* op = opnd + (size_t)lane * 4;
    const unsigned long long tc0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        u32x16 xAh = op[0], xAl = op[1];
        f32x16 i0 = __builtin_bit_cast(f32x16, op[2]), i1 = __builtin_bit_cast(f32x16, op[3]);
        int aw1 = lane * 16, aw2 = FRAG_W2 * 16 + lane * 16, ab = FRAG_END * 16 + (CONST_B1 + 4 * h) * 4;
        int tcount;
        const float c4 = 0.0136151873f;
        f32x16 o0, o1;
        asm volatile(PF_HID1_ASM
                     : PF_HID1_OUT0_A(o0), PF_HID1_OUT1_A(o1), PF_HID1_INIT0_A(i0), PF_HID1_INIT1_A(i1),
                       PF_HID1_AW1(aw1), PF_HID1_AW2(aw2), PF_HID1_AB(ab), [t] "=&s"(tcount)
                     : PF_HID1_XH_A(xAh), PF_HID1_XL_A(xAl), PF_HID1_C4(c4),
                       [c5] "s"(-0.00107098569f), [c3] "s"(-0.084594565f), [c2] "s"(-0.637684925f),
                       [c1] "s"(-1.35494915f), [c0] "s"(-0.00003762f), [sl] "s"(0x0000bf80u), [sh] "s"(PF_SH_CONST)
                     : PF_HID1_CLOBBERS, "scc", "memory");
        f32x16* o = reinterpret_cast<f32x16*>(out) + ((size_t)blockIdx.x * THREADS + threadIdx.x) * 2;
        o[0] = o0; o[1] = o1;
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - tc0;
}

template <int THREADS>
void run_lean(const char* name, const frag_t* wimg, const float* consts, const u32x16* opnd, float* out, int iters) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hid1_lean<THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        MAIN_LDS_BYTES);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_hid1_lean<THREADS>), dim3(256), dim3(THREADS), MAIN_LDS_BYTES, 0, wimg, consts, opnd, out, iters, g_cyc);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    hipError_t e = hipGetLastError();
    const double tiles = 256.0 * (THREADS / 64) * iters;
    const double wc = mean_cycles();                                  // shader cycles of one wave for the whole loop
    const double ghz = wc / (best * 1e-3) * 1e-9;                     // effective shader clock
    const double real = wc / iters / (THREADS / 256);                 // shader cycles per tile per SIMD
    printf("%-22s thr %3d %8.3f ms  %6.2f us/tile/SIMD  %7.0f shader cycles/tile/SIMD (MFMA floor 6144)  clock %.2f GHz %s\n", name,
           THREADS, best, best * 1e3 * 1024.0 / tiles, real, ghz, e == hipSuccess ? "" : hipGetErrorString(e));
}

static std::vector<float> g_ref;


template <int MODE, int THREADS>
void run(const char* name, const frag_t* wimg, const float* consts, float* out, float* dump, int iters) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hid<MODE, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        MAIN_LDS_BYTES);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    hipMemset(dump, 0, 2 * 512 * 32 * 4);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_hid<MODE, THREADS>), dim3(256), dim3(THREADS), MAIN_LDS_BYTES, 0, wimg, consts, out, dump, iters, g_cyc);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    hipError_t e = hipGetLastError();
    constexpr int NT = (MODE == M_HID1) ? 1 : 2;
    std::vector<float> d((size_t)2 * 512 * 32, 0.f);
    hipMemcpy(d.data(), dump, d.size() * 4, hipMemcpyDeviceToHost);
    // the first wave's tile 0 starts from the same x in every variant: compare it with the plain result
    double maxdiff = -1;
    if (MODE == M_PLAIN && THREADS == 256) g_ref = d;
    if (!g_ref.empty()) {
        maxdiff = 0;
        for (int i = 0; i < 64 * 32; ++i) maxdiff = std::fmax(maxdiff, std::fabs((double)d[i] - (double)g_ref[i]));
        if (NT == 2 && MODE != M_HID1 && THREADS == 256)
            for (int i = 0; i < 64 * 32; ++i)
                maxdiff = std::fmax(maxdiff, std::fabs((double)d[(size_t)256 * 32 + i] - (double)g_ref[(size_t)256 * 32 + i]));
    }
    const double tiles = 256.0 * (THREADS / 64) * iters * NT;
    const double cyc = best * 1e-3 * 2.4e9 * 1024.0 / tiles;
    const double wc = mean_cycles();
    const double ghz = wc / (best * 1e-3) * 1e-9;
    const double real = wc / iters / NT / (THREADS / 256);
    printf("%-22s thr %3d %8.3f ms  %6.2f us/tile/SIMD  %7.0f shader cycles/tile/SIMD (MFMA floor 6144)  clock %.2f GHz  maxdiff %.3g %s\n",
           name, THREADS, best, best * 1e3 * 1024.0 / tiles, real, ghz, maxdiff, e == hipSuccess ? "" : hipGetErrorString(e));
    (void)cyc;
}

// ---- energy mode: `ffn3_bench_<mode> energy [seconds]` -------------------------------------------
// Runs one variant back to back for a few seconds and reads the socket energy counter
// (rsmi_dev_energy_count_get, librocm_smi64 through dlopen: sysfs reads, no subprocess) before and after:
// joules per 32-token tile, average watts and the effective shader clock under sustained load.
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

template <typename Launch>
void energy_run(Rsmi& smi, const char* name, double seconds, double tiles_per_launch, int waves_per_simd, Launch launch) {
    launch();                                       // warm-up: clocks up, LDS image resident in L2
    hipDeviceSynchronize();
    const double j0 = smi.joules();
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    double dt = 0;
    do {
        for (int i = 0; i < 8; ++i) launch();
        hipDeviceSynchronize();
        n += 8;
        dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (dt < seconds);
    const double j1 = smi.joules();
    const double wc = mean_cycles();                // shader cycles of one wave for the last launch
    const double tiles = tiles_per_launch * n;
    const double per_launch_s = dt / n;
    printf("energy %-22s %7.1f W  %8.3f uJ/tile  %6.2f us/tile/SIMD  %7.0f cycles/tile/SIMD  clock %.2f GHz  (%ld launches, %.2f s, %.1f J)\n",
           name, (j1 - j0) / dt, (j1 - j0) / tiles * 1e6, dt * 1e6 * 1024.0 / tiles,
           wc / (tiles_per_launch / 1024.0), wc / per_launch_s * 1e-9, n, dt, j1 - j0);
    (void)waves_per_simd;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const bool energy = argc > 1 && std::strcmp(argv[1], "energy") == 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    const bool zero = getenv("PF_ZERO") != nullptr;      // all-zero weights: the data-dependent part of the power
    const int iters = energy ? 2048 : 64;
    std::vector<uint16_t> img((size_t)FRAG_END * 8);
    for (size_t i = 0; i < img.size(); ++i) {
        // W ~ +-[0.03, 0.12], deterministic; hi fragments hold the value rounded to the operand format, lo fragments
        // (every second group of 64 in the W1 / W2 / Wo regions) its rounding residual - what the product image holds
        const uint32_t r = (uint32_t)((i / 8 % 64 + (i / 1024) * 64) * 8 + i % 8) * 2654435761u;
        const float w = (0.03f + 0.09f * (float)((r >> 20) % 256) / 256.f + 1e-5f * (float)((r >> 3) % 97)) * ((r >> 9) & 1 ? -1.f : 1.f);
        const uint16_t hi = pfhost::f2x(w);
        const bool lo_frag = (i / 8 / 64) % 2 == 1;
        img[i] = zero ? 0 : (lo_frag ? pfhost::f2x(w - pfhost::x2f(hi)) : hi);
    }
    std::vector<float> cst(CONST_LEN);
    for (int i = 0; i < CONST_LEN; ++i) cst[i] = 0.05f * (float)((i * 37) % 21 - 10);
    frag_t* d_img; float *d_c, *d_out, *d_dump;
    hipMalloc((void**)&d_img, img.size() * 2);
    hipMalloc((void**)&d_c, cst.size() * 4);
    hipMalloc((void**)&d_out, 256 * 512 * 4);
    hipMalloc((void**)&d_dump, 2 * 512 * 32 * 4);
    hipMalloc((void**)&g_cyc, 256 * 8);
    hipMemcpy(d_img, img.data(), img.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(d_c, cst.data(), cst.size() * 4, hipMemcpyHostToDevice);
    if (energy) {
        Rsmi smi;
        if (!smi.ok) return 1;
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hid<M_HID2, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, MAIN_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hid<M_PLAIN, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, MAIN_LDS_BYTES);
        {   // idle baseline
            const double j0 = smi.joules();
            const auto t0 = std::chrono::steady_clock::now();
            hipDeviceSynchronize();
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.0) {}
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("energy idle                   %7.1f W\n", (smi.joules() - j0) / dt);
        }
        if (!PF_F16)      // (the generated streams are written with the bf16 mnemonics)
        energy_run(smi, "asm two-tile 1w/SIMD", seconds, 256.0 * 4 * iters * 2, 1, [&] {
            hipLaunchKernelGGL((k_hid<M_HID2, 256>), dim3(256), dim3(256), MAIN_LDS_BYTES, 0, d_img, d_c, d_out, (float*)nullptr, iters, g_cyc);
        });
        if (getenv("PF_PLAIN"))
            energy_run(smi, "plain C++ 2w/SIMD", seconds, 256.0 * 8 * iters * 2, 2, [&] {
                hipLaunchKernelGGL((k_hid<M_PLAIN, 512>), dim3(256), dim3(512), MAIN_LDS_BYTES, 0, d_img, d_c, d_out, (float*)nullptr, iters, g_cyc);
            });
        return 0;
    }
    run<M_PLAIN, 256>("plain C++ 1w/SIMD", d_img, d_c, d_out, d_dump, iters);
    run<M_PLAIN, 512>("plain C++ 2w/SIMD", d_img, d_c, d_out, d_dump, iters);
    run<M_HID2, 256>("asm two-tile 1w/SIMD", d_img, d_c, d_out, d_dump, iters);
    run<M_HID1, 256>("asm one-tile 1w/SIMD", d_img, d_c, d_out, d_dump, iters);
    {
        std::vector<uint32_t> opv((size_t)64 * 64);
        for (size_t i = 0; i < opv.size(); ++i) {
            const uint32_t r = (uint32_t)(i * 2246822519u);
            // bf16 pairs ~ +-[0.25, 1] for the x fragments (first 32 dwords per lane), small floats for the start values
            opv[i] = ((i & 63) < 32) ? (0x3e803e80u + ((r >> 8) & 0x007f007fu) + ((r & 1) ? 0x80000000u : 0) + ((r & 2) ? 0x8000u : 0))
                                     : 0x3c000000u + (r >> 12);
        }
        u32x16* d_op; float* d_o2;
        hipMalloc((void**)&d_op, opv.size() * 4);
        hipMalloc((void**)&d_o2, (size_t)256 * 512 * 32 * 4);
        hipMemcpy(d_op, opv.data(), opv.size() * 4, hipMemcpyHostToDevice);
        run_lean<256>("lean one-tile 1w/SIMD", d_img, d_c, d_op, d_o2, iters);
        run_lean<512>("lean one-tile 2w/SIMD", d_img, d_c, d_op, d_o2, iters);
    }
    return 0;
}
