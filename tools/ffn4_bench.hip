// Energy / time per 32-token tile of the FFN hidden loop as hipcc compiles it at two waves per SIMD:
//   NT = 1  the loop of k_main (one tile per wave, every weight fragment read from LDS per tile)
//   NT = 2  two tiles per wave, every LDS-read weight fragment feeds BOTH tiles' MFMAs (k_main's k_pair form)
// Same arithmetic per tile (bit-identical sums); what changes is LDS bytes per tile (128 KB -> 64 KB).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/ffn4_bench.hip -o tools/ffn4_bench -ldl
//   tools/ffn4_bench [seconds]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../phyloformer_amd/csrc/pf_device.hip.h"

using namespace pfk;

template <int NT>
__global__ void __launch_bounds__(512, 2) k_ffn(const bf16x8* wimg, const float* consts, const float* xin, float* out,
                                                int iters, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_frag_t lw = (lds_frag_t)smem;
    lds_f32_t lc = (lds_f32_t)(smem + FRAG_END * 16);
    {
        uint4* dst = reinterpret_cast<uint4*>(smem);
        const uint4* src = reinterpret_cast<const uint4*>(wimg);
        for (int i = threadIdx.x; i < FRAG_END; i += 512) dst[i] = src[i];
        float* dc = reinterpret_cast<float*>(smem + FRAG_END * 16);
        for (int i = threadIdx.x; i < CONST_LEN; i += 512) dc[i] = consts[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    lds_frag_t w1p = lw + FRAG_W1 + lane;
    lds_frag_t w2p = lw + FRAG_W2 + lane;
    lds_f32_t lch = lc + 4 * h;
    float x[NT][32];
    const float* xp = xin + ((size_t)blockIdx.x * 512 + threadIdx.x) * 64;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 32; ++j) x[t][j] = xp[t * 32 + j];
    const unsigned long long tc0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += NT) {
        bf16x8 xh[NT][4], xl[NT][4];
        f32x16 oa[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float xn[32];
            ln_pair(x[t], xn);
#pragma unroll
            for (int s = 0; s < 4; ++s) split8(&xn[8 * s], xh[t][s], xl[t][s]);
            load_acc_bias(oa[t][0], lch + CONST_B2);
            load_acc_bias(oa[t][1], lch + CONST_B2 + 32);
#pragma unroll
            for (int j = 0; j < 32; ++j) oa[t][j >> 4][j & 15] += x[t][j];
        }
#pragma unroll 1
        for (int T = 0; T < 8; ++T) {
            lds_frag_t f1 = w1p + T * 512;
            lds_frag_t f2 = w2p + T * 256;
            lds_f32_t bp = lch + CONST_B1 + 32 * T;
            PF_OPAQUE(f1); PF_OPAQUE(f2); PF_OPAQUE(bp);
            f32x16 ha[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) load_acc_bias(ha[t], bp);
            bf16x8 fh = f1[0], fl = f1[64];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 nh, nl;
                if (s < 3) { nh = f1[(s + 1) * 128]; nl = f1[(s + 1) * 128 + 64]; }
                else { nh = f2[0]; nl = f2[64]; }
#pragma unroll
                for (int t = 0; t < NT; ++t) mfma3(ha[t], fh, fl, xh[t][s], xl[t][s], t == 1);
                fh = nh; fl = nl;
            }
            bf16x8 g_hi[NT][2], g_lo[NT][2];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                gelu_split8(ha[t], 0, g_hi[t][0], g_lo[t][0]);
                gelu_split8(ha[t], 8, g_hi[t][1], g_lo[t][1]);
            }
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int u = st >> 1, To = st & 1;
                bf16x8 nh = fh, nl = fl;
                if (st < 3) {
                    const int nu = (st + 1) >> 1, nTo = (st + 1) & 1;
                    nh = f2[(nTo * 32 + nu * 2) * 64];
                    nl = f2[(nTo * 32 + nu * 2) * 64 + 64];
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) mfma3(oa[t][To], fh, fl, g_hi[t][u], g_lo[t][u], (To == 1) != (t == 1));
                fh = nh; fl = nl;
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 32; ++j) x[t][j] = 0.01f * (float)((lane + j * 5) % 61) + 1e-2f * oa[t][j >> 4][j & 15];
    }
    const unsigned long long tc1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 32; ++j) s += x[t][j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = tc1 - tc0;
}

struct Rsmi {
    void* lib = nullptr;
    int (*energy)(uint32_t, uint64_t*, float*, uint64_t*) = nullptr;
    bool ok = false;
    Rsmi() {
        lib = dlopen("/opt/rocm/lib/librocm_smi64.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("librocm_smi64.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) { printf("cannot load librocm_smi64: %s\n", dlerror()); return; }
        auto init = reinterpret_cast<int (*)(uint64_t)>(dlsym(lib, "rsmi_init"));
        energy = reinterpret_cast<int (*)(uint32_t, uint64_t*, float*, uint64_t*)>(dlsym(lib, "rsmi_dev_energy_count_get"));
        if (!init || !energy || init(0) != 0) { printf("rsmi_init failed\n"); return; }
        ok = true;
    }
    double joules() {
        uint64_t c = 0, ts = 0; float res = 0;
        if (!ok || energy(0, &c, &res, &ts) != 0) return -1;
        return (double)c * res * 1e-6;
    }
};

static unsigned long long* g_cyc = nullptr;

template <int NT>
void run(Rsmi& smi, double seconds, const bf16x8* wimg, const float* consts, const float* xin, float* out) {
    const int iters = 1024;        // tiles per wave and launch
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ffn<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, MAIN_LDS_BYTES);
    auto launch = [&] { hipLaunchKernelGGL((k_ffn<NT>), dim3(256), dim3(512), MAIN_LDS_BYTES, 0, wimg, consts, xin, out, iters, g_cyc); };
    launch();
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
    std::vector<unsigned long long> c(256);
    hipMemcpy(c.data(), g_cyc, 256 * 8, hipMemcpyDeviceToHost);
    double wc = 0;
    for (auto v : c) wc += (double)v;
    wc /= 256.0;
    std::vector<float> o(256 * 512);
    hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
    double chk = 0;
    for (float v : o) chk += v;
    const double tiles = 256.0 * 8 * iters * n;
    printf("NT=%d  %7.1f W  %8.3f uJ/tile  %6.2f us/tile/SIMD  %7.0f cycles/tile/SIMD  clock %.2f GHz  checksum %.6e  (%s)\n", NT,
           (j1 - j0) / dt, (j1 - j0) / tiles * 1e6, dt * 1e6 * 1024.0 / tiles, wc / iters / 2.0 * 1.0, wc / (dt / n) * 1e-9, chk,
           hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    std::vector<uint16_t> img((size_t)FRAG_END * 8);
    for (size_t i = 0; i < img.size(); ++i) {
        const uint32_t r = (uint32_t)(i * 2654435761u);
        img[i] = (uint16_t)(0x3d00 + ((r >> 20) % 256) + ((r >> 9) & 1 ? 0x8000 : 0));
    }
    std::vector<float> cst(CONST_LEN);
    for (int i = 0; i < CONST_LEN; ++i) cst[i] = 0.05f * (float)((i * 37) % 21 - 10);
    std::vector<float> xin((size_t)256 * 512 * 64);
    for (size_t i = 0; i < xin.size(); ++i) xin[i] = 0.01f * (float)((i * 7919u) % 197) - 0.9f;
    bf16x8* d_img; float *d_c, *d_x, *d_out;
    hipMalloc((void**)&d_img, img.size() * 2);
    hipMalloc((void**)&d_c, cst.size() * 4);
    hipMalloc((void**)&d_x, xin.size() * 4);
    hipMalloc((void**)&d_out, 256 * 512 * 4);
    hipMalloc((void**)&g_cyc, 256 * 8);
    hipMemcpy(d_img, img.data(), img.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(d_c, cst.data(), cst.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_x, xin.data(), xin.size() * 4, hipMemcpyHostToDevice);
    Rsmi smi;
    if (!smi.ok) return 1;
    for (int rep = 0; rep < 2; ++rep) {
        run<1>(smi, seconds, d_img, d_c, d_x, d_out);
        run<2>(smi, seconds, d_img, d_c, d_x, d_out);
    }
    return 0;
}
