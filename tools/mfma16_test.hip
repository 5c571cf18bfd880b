// Layout facts for the softmax operator's PV product on v_mfma_f32_16x16x32_bf16 (gfx950):
//  (1) v_permlane16_swap_b32 x, y: x.row1 <-> y.row0, x.row3 <-> y.row2 (rows = 16 lanes)
//  (2) A[m][k]: lane = 16*(k/8) + m holds k%8 = 0..7;  B[k][n]: lane = 16*(k/8) + n;  D[m][n]: lane = 16*(m/4) + n, reg m%4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    unsigned x = 1000 + lane, y = 2000 + lane;
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    out[lane] = (float)x; out[64 + lane] = (float)y;
    // A[m][k] = m + 16*(k%4) + (k/4 == 3 ? 0 : 0)  -> use A[m][k] = (k == 5 ? m : 0), B[k][n] = (k == 5 ? n+1 : 0) -> D = m*(n+1)
    bf16x8 a, b;
    const int m = lane & 15, g = lane >> 4;
    for (int i = 0; i < 8; ++i) {
        const int kk = 8 * g + i;
        a[i] = (__bf16)(float)((kk == 13) ? m : (kk == 30 ? 1 : 0));
        b[i] = (__bf16)(float)((kk == 13) ? (m + 1) : (kk == 30 ? 100 : 0));
    }
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[128 + lane * 4 + r] = d[r];
}
int main() {
    float* d; hipMalloc(&d, 4096); k<<<1, 64>>>(d); float h[1024]; hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int row = l >> 4;
        const float ex = (row & 1) ? 2000 + (l - 16) : 1000 + l;      // x' = [x.r0, y.r0, x.r2, y.r2]
        const float ey = (row & 1) ? 2000 + l : 1000 + (l + 16);      // y' = [x.r1, y.r1, x.r3, y.r3]
        if (h[l] != ex || h[64 + l] != ey) { if (bad < 8) printf("swap lane %d: x %g (want %g) y %g (want %g)\n", l, h[l], ex, h[64 + l], ey); ++bad; }
        for (int r = 0; r < 4; ++r) {
            const int mm = 4 * (l >> 4) + r, nn = l & 15;
            const float want = (float)(mm * (nn + 1) + 100);
            if (h[128 + l * 4 + r] != want) { if (bad < 16) printf("mfma lane %d r %d: %g want %g\n", l, r, h[128 + l * 4 + r], want); ++bad; }
        }
    }
    printf(bad ? "FAILED %d\n" : "ok: permlane16_swap and 16x16x32 layouts as assumed\n", bad);
    return bad != 0;
}
