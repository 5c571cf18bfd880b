// Cost of a grid-wide barrier inside one launch against the cost of a kernel boundary (is
// fusing the per-block trio k_rowfin -> k_colstats -> k_colfin of a LONE small alignment worth it?).
//   hipcc --offload-arch=gfx950 -O3 -o tools/gridbar_bench tools/gridbar_bench.hip && tools/gridbar_bench
// Prints, for grids of 64 .. 512 blocks of 256 threads: us per dependent (almost empty) kernel launch on one stream,
// and us per grid barrier (release fence + device-scope atomic + spin + acquire fence) inside one launch, with every
// block writing a value its successor block reads after the barrier (cross-XCD visibility is part of the price).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_step(float* buf, int n, int it) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0) buf[(size_t)(it & 1) * n + b] = buf[(size_t)((it + 1) & 1) * n + (b + 1) % n] + 1.f;
}

__device__ __forceinline__ void grid_barrier(unsigned long long* ctr, unsigned long long target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                   // release: this block's writes reach device scope
        atomicAdd(ctr, 1ull);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __threadfence();                                   // acquire
    }
    __syncthreads();
}

__global__ void k_fused(float* buf, int n, int iters, unsigned long long* ctr, unsigned long long base) {
    const int b = blockIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (threadIdx.x == 0) buf[(size_t)(it & 1) * n + b] = buf[(size_t)((it + 1) & 1) * n + (b + 1) % n] + 1.f;
        grid_barrier(ctr, base + (unsigned long long)(it + 1) * n);
    }
}

int main() {
    hipStream_t st;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float* buf;
    unsigned long long* ctr;
    CHK(hipMalloc(&buf, 2 * 4096 * sizeof(float)));
    CHK(hipMalloc(&ctr, 8));
    CHK(hipMemset(ctr, 0, 8));
    unsigned long long base = 0;
    const int iters = 200;
    for (int n : {64, 128, 256, 512}) {
        CHK(hipMemset(buf, 0, 2 * 4096 * sizeof(float)));
        for (int rep = 0; rep < 2; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(k_step, dim3(n), dim3(256), 0, st, buf, n, it);
            CHK(hipStreamSynchronize(st));
            const double us_launch = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
            t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_fused, dim3(n), dim3(256), 0, st, buf, n, iters, ctr, base);
            CHK(hipStreamSynchronize(st));
            base += (unsigned long long)iters * n;
            const double us_bar = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
            std::vector<float> h(n);
            CHK(hipMemcpy(h.data(), buf + (size_t)((iters - 1) & 1) * n, n * sizeof(float), hipMemcpyDeviceToHost));
            if (rep) printf("grid %4d x 256: %.2f us per dependent launch, %.2f us per grid barrier (check: %.0f = %d)\n", n, us_launch,
                            us_bar, h[0], 2 * iters * 2);
        }
    }
    return 0;
}
