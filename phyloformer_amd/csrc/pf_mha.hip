// Kernels of the softmax multi-head self-attention operator (interface, layouts and the reference it follows:
// pf_mha.hip.h).  Its own translation unit - see the note on the launchers there.
#define PF_DEVICE_HELPERS_ONLY          // the operand-split / MFMA / cross-lane helpers of pf_device.hip.h, not its kernels
#include "pf_device.hip.h"
#include "pf_mha.hip.h"

namespace pfk {

// 32 channels of the lane's token -> hi / lo fragments of the four K steps
__device__ __forceinline__ void mha_load_tile(const float* __restrict__ src, int h, frag_t (&xh)[4], frag_t (&xl)[4]) {
    float xv[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + 8 * q + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[4 * q + i] = v[i];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) split8(&xv[8 * s], xh[s], xl[s]);
}

// acc[mt][j] = bias[32 mt + kmap(j,h)] + sum_k W[32 mt + kmap(j,h)][k] x[token][k]
__device__ __forceinline__ void mha_linear(const frag_t* wl, const float* bl, int lane, int h,
                                           const frag_t (&xh)[4], const frag_t (&xl)[4], f32x16 (&acc)[2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        load_acc_bias(acc[mt], bl + 32 * mt, h);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const frag_t a_hi = wl[((mt * 4 + s) * 2 + 0) * 64 + lane];
            const frag_t a_lo = wl[((mt * 4 + s) * 2 + 1) * 64 + lane];
            mfma3(acc[mt], a_hi, a_lo, xh[s], xl[s]);
        }
    }
}

__global__ void __launch_bounds__(256) k_mha_qkv(MhaArgs a) {
    __shared__ frag_t wl[MHA_OFF_WO];
    __shared__ float bl[3 * 64];
    for (int i = threadIdx.x; i < MHA_OFF_WO; i += 256) wl[i] = a.wfrag[i];
    if (threadIdx.x < 192) bl[threadIdx.x] = a.bias[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, t = lane & 31, h = lane >> 5;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int Cpad = a.ntiles * 32;
    const size_t plane_qk = (size_t)a.rows * MHA_H * Cpad * 2;
    const size_t plane_v = (size_t)a.rows * MHA_H * a.ntiles * 64;
    for (int tile = wave; tile < a.rows * a.ntiles; tile += nwaves) {
        const int row = tile / a.ntiles, kt = tile - row * a.ntiles;
        const int c = kt * 32 + t, cc = min(c, a.C - 1);
        frag_t xh[4], xl[4];
        mha_load_tile(a.x + ((size_t)row * a.C + cc) * 64, h, xh, xl);
        // Q and K: weights on M, tokens on N -> lane (t,h) holds 8 channels of each head
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            f32x16 acc[2];
            mha_linear(wl + which * MHA_WFRAGS, bl + which * 64, lane, h, xh, xl, acc);
            frag_t* dst = which == 0 ? a.qp : a.kp;
            const float sc = which == 0 ? a.qscale : 1.f;
#pragma unroll
            for (int hd = 0; hd < MHA_H; ++hd) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = acc[hd >> 1][8 * (hd & 1) + i] * sc;
                frag_t fh, fl;
                split8(v, fh, fl);
                const size_t o = (((size_t)row * MHA_H + hd) * Cpad + c) * 2 + h;
                dst[o] = fh;
                dst[plane_qk + o] = fl;
            }
        }
        // V: tokens on M, channels on N -> lane (n = channel, h) holds 16 keys of its channel
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f32x16 acc;
            const float bv = bl[128 + 32 * nt + t];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = bv;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const frag_t b_hi = wl[MHA_OFF_WV + ((nt * 4 + s) * 2 + 0) * 64 + lane];
                const frag_t b_lo = wl[MHA_OFF_WV + ((nt * 4 + s) * 2 + 1) * 64 + lane];
                mfma3(acc, xh[s], xl[s], b_hi, b_lo);
            }
            const int hd = 2 * nt + (t >> 4), d = t & 15;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = acc[8 * s2 + i];
                frag_t fh, fl;
                split8(v, fh, fl);
                // A operand of v_mfma_f32_16x16x32_f16: lane 16 g + d holds 8 keys of channel d; key group
                // g = 2 h + s2 is the set kmap(8 s2 .. 8 s2 + 7, h) - the order the P fragments of k_mha_attn
                // arrive in after their permlane16 swap
                const size_t o = (((size_t)row * MHA_H + hd) * a.ntiles + kt) * 64 + 16 * (2 * h + s2) + d;
                a.vp[o] = fh;
                a.vp[plane_v + o] = fl;
            }
        }
    }
}

// x <-> y exchange of 16-lane rows: x.row1 <-> y.row0, x.row3 <-> y.row2 (tools/mfma16_test.hip).  With x, y
// the lane's values for two key groups, x' then holds queries 0-15 in every row and y' queries 16-31.
__device__ __forceinline__ void swap_rows16(unsigned& x, unsigned& y) {
    const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    x = r[0]; y = r[1];
}
__device__ __forceinline__ void swap_rows16(float& x, float& y) {
    unsigned a = __builtin_bit_cast(unsigned, x), b = __builtin_bit_cast(unsigned, y);
    swap_rows16(a, b);
    x = __builtin_bit_cast(float, a); y = __builtin_bit_cast(float, b);
}
__device__ __forceinline__ void swap_rows16(frag_t& x, frag_t& y) {
    u32x4 a = __builtin_bit_cast(u32x4, x), b = __builtin_bit_cast(u32x4, y);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned u = a[i], v = b[i];
        swap_rows16(u, v);
        a[i] = u; b[i] = v;
    }
    x = __builtin_bit_cast(frag_t, a); y = __builtin_bit_cast(frag_t, b);
}
#if PF_F16
#define PF_MFMA32_ASM "v_mfma_f32_32x32x16_f16"
#define PF_MFMA16_ASM "v_mfma_f32_16x16x32_f16"
#else
#define PF_MFMA32_ASM "v_mfma_f32_32x32x16_bf16"
#define PF_MFMA16_ASM "v_mfma_f32_16x16x32_bf16"
#endif
constexpr float MHA_DEFER = 6.f;   // the running maximum is raised only when a tile exceeds it by 2^6

__global__ void __launch_bounds__(256) k_mha_attn(MhaArgs a) {
    // per key tile: K hi / lo, V hi / lo: 64 fragments of 16 B each
    __shared__ frag_t stage[2][MHA_KB][4][64];
    const int lane = threadIdx.x & 63, t = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
    const int nqb = (a.ntiles + 3) / 4;
    const int qb = blockIdx.x % nqb, rh = blockIdx.x / nqb;      // rh = row * H + head
    const int hd = rh % MHA_H, row = rh / MHA_H;
    const int Cpad = a.ntiles * 32;
    const size_t plane_qk = (size_t)a.rows * MHA_H * Cpad * 2;
    const size_t plane_v = (size_t)a.rows * MHA_H * a.ntiles * 64;
    const int qt = min(qb * 4 + w, a.ntiles - 1);
    const size_t qo = ((size_t)rh * Cpad + qt * 32 + t) * 2 + h;
    const frag_t qh = a.qp[qo], ql = a.qp[plane_qk + qo];
    // wave w of the workgroup stages key tile w of every stage (MHA_KB == waves per workgroup)
    const frag_t* ksrc = a.kp + (size_t)rh * Cpad * 2 + lane;
    const frag_t* vsrc = a.vp + (size_t)rh * a.ntiles * 64 + lane;
    const int nst = (a.ntiles + MHA_KB - 1) / MHA_KB;
    // global_load_lds_dwordx4: the four fragment planes of the wave's key tile go straight to LDS
    // (destination = wave-uniform base + lane * 16), no staging registers
    auto fetch = [&](int st, int buf) {
        const size_t kt = (size_t)min(st * MHA_KB + w, a.ntiles - 1) * 64;
        __builtin_amdgcn_global_load_lds(ksrc + kt, &stage[buf][w][0][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds(ksrc + plane_qk + kt, &stage[buf][w][1][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds(vsrc + kt, &stage[buf][w][2][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds(vsrc + plane_v + kt, &stage[buf][w][3][0], 16, 0, 0);
    };
    fetch(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // O^T for queries 0-15 / 16-31 of the tile: lane 16 g + n holds channels 4 g .. 4 g + 3 of query n
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    float m_ref = -INFINITY, lsum = 0.f;
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        if (st + 1 < nst) fetch(st + 1, buf ^ 1);
        const int ktn = min(MHA_KB, a.ntiles - st * MHA_KB);
        for (int k = 0; k < ktn; ++k) {
            const frag_t kh = stage[buf][k][0][2 * t + h], kl = stage[buf][k][1][2 * t + h];
            // s[j] = S[key kmap(j,h)][query t], log2 units.  The three passes (small terms first) written out with the
            // accumulator in VGPRs: hipcc would park it in AGPRs and pay a v_accvgpr_read per score (16 of the
            // ~140 instructions of a tile).  The trailing s_nop is the 8-pass XDL-write -> VALU-read distance
            // (11 wait states), which hipcc cannot see inside an asm statement.
            f32x16 s;
            asm volatile(
                PF_MFMA32_ASM " %0, %1, %4, 0\n\t"
                PF_MFMA32_ASM " %0, %2, %3, %0\n\t"
                PF_MFMA32_ASM " %0, %1, %3, %0\n\t"
                "s_nop 10"
                : "=&v"(s) : "v"(kh), "v"(kl), "v"(qh), "v"(ql));
            const int key0 = (st * MHA_KB + k) * 32;
            if (key0 + 32 > a.C) {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (key0 + kmap(j, h) >= a.C) s[j] = -INFINITY;
            }
            float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);      // (v_max3_f32)
#pragma unroll
            for (int j = 3; j < 15; j += 2) mx = fmaxf(fmaxf(mx, s[j]), s[j + 1]);
            mx = fmaxf(mx, s[15]);
            mx = fmaxf(mx, pair_other(mx, h));
            // Deferred rescaling: the reference maximum moves only when some query of the tile beats it by
            // 2^MHA_DEFER (always on the first tile); p <= 2^MHA_DEFER otherwise, which fp32 / the split carry
            // exactly as well.  Wave-uniform branch; lanes that did not trip it rescale by alpha <= 1 too.
            if (__builtin_amdgcn_ballot_w64(mx > m_ref + MHA_DEFER) != 0) {
                const float m_new = fmaxf(m_ref, mx);
                const float alpha = __builtin_amdgcn_exp2f(m_ref - m_new);
                m_ref = m_new;
                lsum *= alpha;
                float a0 = alpha, a1 = alpha;               // alpha of query n / 16 + n for every row of lanes
                swap_rows16(a0, a1);
#pragma unroll
                for (int i = 0; i < 4; ++i) { o0[i] *= a0; o1[i] *= a1; }
            }
            float p[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) p[j] = __builtin_amdgcn_exp2f(s[j] - m_ref);
            lsum += (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) +
                    (((p[8] + p[9]) + (p[10] + p[11])) + ((p[12] + p[13]) + (p[14] + p[15])));
            frag_t ph0, pl0, ph1, pl1;
            split8(&p[0], ph0, pl0);                        // keys kmap(0..7, h)
            split8(&p[8], ph1, pl1);                        // keys kmap(8..15, h)
            swap_rows16(ph0, ph1);                          // -> B operands: queries 0-15 | 16-31, key group per row
            swap_rows16(pl0, pl1);
            const frag_t vh = stage[buf][k][2][lane], vl = stage[buf][k][3][lane];
            // O^T += V^T P^T, small terms first, the two query halves alternating; accumulators in VGPRs (their
            // next VALU reader - a rescale or the epilogue - is more than the 7 wait states of a 4-pass MFMA away)
            asm volatile(
                "s_nop 1\n\t"                               // (the P fragments were just written by v_permlane16_swap)
                PF_MFMA16_ASM " %0, %3, %4, %0\n\t"
                PF_MFMA16_ASM " %1, %3, %6, %1\n\t"
                PF_MFMA16_ASM " %0, %2, %5, %0\n\t"
                PF_MFMA16_ASM " %1, %2, %7, %1\n\t"
                PF_MFMA16_ASM " %0, %2, %4, %0\n\t"
                PF_MFMA16_ASM " %1, %2, %6, %1"
                : "+v"(o0), "+v"(o1) : "v"(vh), "v"(vl), "v"(ph0), "v"(pl0), "v"(ph1), "v"(pl1));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const float l = lsum + pair_other(lsum, h);
    float i0 = 1.f / l, i1 = i0;
    swap_rows16(i0, i1);
    const int n = lane & 15, g = lane >> 4;
    const int c0 = (qb * 4 + w) * 32 + n;
    if (qb * 4 + w < a.ntiles) {
        float* dst = a.att + ((size_t)row * a.C + c0) * 64 + hd * 16 + 4 * g;
        if (c0 < a.C) *reinterpret_cast<f32x4*>(dst) = o0 * i0;
        if (c0 + 16 < a.C) *reinterpret_cast<f32x4*>(dst + 16 * 64) = o1 * i1;
    }
}

__global__ void __launch_bounds__(256) k_mha_out(MhaArgs a) {
    __shared__ frag_t wl[MHA_WFRAGS];
    __shared__ float bl[64];
    for (int i = threadIdx.x; i < MHA_WFRAGS; i += 256) wl[i] = a.wfrag[MHA_OFF_WO + i];
    if (threadIdx.x < 64) bl[threadIdx.x] = a.bias[192 + threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, t = lane & 31, h = lane >> 5;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const size_t ntok = (size_t)a.rows * a.C;
    const int nt = (int)((ntok + 31) / 32);
    for (int tile = wave; tile < nt; tile += nwaves) {
        const size_t tok = (size_t)tile * 32 + t;
        const size_t tc = tok < ntok ? tok : ntok - 1;
        frag_t xh[4], xl[4];
        mha_load_tile(a.att + tc * 64, h, xh, xl);
        f32x16 acc[2];
        mha_linear(wl, bl, lane, h, xh, xl, acc);
        if (tok < ntok) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = acc[mt][4 * q4 + i];
                    *reinterpret_cast<f32x4*>(a.y + tok * 64 + 32 * mt + 8 * q4 + 4 * h) = v;
                }
        }
    }
}


void launch_mha_qkv(hipStream_t s, unsigned grid, const MhaArgs& a) { hipLaunchKernelGGL(k_mha_qkv, dim3(grid), dim3(256), 0, s, a); }
void launch_mha_attn(hipStream_t s, unsigned grid, const MhaArgs& a) { hipLaunchKernelGGL(k_mha_attn, dim3(grid), dim3(256), 0, s, a); }
void launch_mha_out(hipStream_t s, unsigned grid, const MhaArgs& a) { hipLaunchKernelGGL(k_mha_out, dim3(grid), dim3(256), 0, s, a); }

}  // namespace pfk
