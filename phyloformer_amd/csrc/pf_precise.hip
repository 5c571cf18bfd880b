// Kernels of the PRECISE (float64) path; rationale, layout and the host-side sequence: pf_precise.hip.h,
// pf_precise_host.hip.h.
#include "pf_precise.hip.h"

namespace pfp {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ double bcast(double v, int lane) {      // `lane` is wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double elu1(double z) { return z > 0.0 ? z + 1.0 : exp(z); }   // attention.py:179-180

// nn.LayerNorm(64), biased variance, eps inside the sqrt (model.py:64-66); lane c holds channel c
__device__ __forceinline__ double layer_norm(double x, double g, double b) {
    const double mu = wave_sum(x) * (1.0 / 64.0);
    const double xc = x - mu;
    const double var = wave_sum(xc * xc) * (1.0 / 64.0);
    return xc / sqrt(var + 1e-5) * g + b;
}

// ---- embedding + pair expansion (model.py:138-143, 173-175) ------------------------------------
__global__ void __launch_bounds__(PT) kp_embed(EmbedArgs a) {
    const size_t total = (size_t)a.B * a.P * a.L * E;
    for (size_t i = (size_t)blockIdx.x * PT + threadIdx.x; i < total; i += (size_t)gridDim.x * PT) {
        const int c = (int)(i & 63);
        const size_t tok = i >> 6;
        const int l = (int)(tok % a.L);
        const size_t bp = tok / a.L;
        const int p = (int)(bp % a.P), b = (int)(bp / a.P);
        int ri = a.idx[((size_t)b * a.N + a.pi[p]) * a.L + l], rj = a.idx[((size_t)b * a.N + a.pj[p]) * a.L + l];
        if ((ri >= NA || rj >= NA) && a.bad) *a.bad = 1u;          // sticky flag, as k_embed (pf_device.hip.h)
        ri = min(ri, NA - 1); rj = min(rj, NA - 1);
        a.x[i] = a.table[ri * E + c] + a.table[rj * E + c];
    }
}

// ---- attention statistics over one axis (attention.py:163-190) ------------------------------------
// A "line" is what the attention reduces over: the Lloc sites of a pair (row attention, model.py:91) or the P
// pairs of a site (column attention, model.py:97).  Block = (line, chunk of CHUNK elements); its four waves take
// the chunk's elements round-robin and leave part[line][chunk][72] = S_kv[64] | S_q[4] | S_k[4] summed in the
// fixed order wave 0 + wave 1 + wave 2 + wave 3.  q' is kept per token for the apply kernel.
__device__ __forceinline__ size_t token_of(int col, int line, int e, int P, int L) {
    if (!col) return (size_t)line * L + e;                       // line = b * P + p, e = l
    const int b = line / L, l = line - b * L;                    // line = b * L + l, e = p
    return ((size_t)b * P + e) * L + l;
}
__global__ void __launch_bounds__(PT) kp_attn_stats(StatsArgs a) {
    __shared__ double wv[E * E];          // 32 KB: WvT
    __shared__ double red[4][SROW];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int line = blockIdx.x / a.nchunk, ch = blockIdx.x - line * a.nchunk;
    const int nelem = a.col ? a.P : a.L;
    for (int i = tid; i < E * E; i += PT) wv[i] = a.w.wvT[i];
    const double g = a.w.g[lane], beta = a.w.b[lane], bv = a.w.bv[lane];
    double wqk[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) wqk[m] = a.w.wqk[m * E + lane];
    const double bqk = a.w.bqk[lane & 7];
    __syncthreads();
    double skv = 0.0, s8 = 0.0;           // lane c: S_kv[c];  lanes 0..7: S_q[0..3], S_k[0..3]
    const int e_end = min(nelem, (ch + 1) * CHUNK);
    for (int e = ch * CHUNK + w; e < e_end; e += 4) {
        const size_t tok = token_of(a.col, line, e, a.P, a.L);
        const double xn = layer_norm(a.x[tok * E + lane], g, beta);
        double qk = 0.0;                  // lane m < 8 ends up with elu(.)+1 of projection m
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const double s = wave_sum(wqk[m] * xn);
            if ((lane & 7) == m) qk = s;
        }
        qk = elu1(qk + bqk);
        double v = bv;
#pragma unroll 8
        for (int k = 0; k < E; ++k) v = fma(wv[k * E + lane], bcast(xn, k), v);
        const double k0 = bcast(qk, 4), k1 = bcast(qk, 5), k2 = bcast(qk, 6), k3 = bcast(qk, 7);
        const double kh = (lane >> 4) == 0 ? k0 : (lane >> 4) == 1 ? k1 : (lane >> 4) == 2 ? k2 : k3;
        skv += kh * v;                    // attention.py:187-188: sum of k'[h] * v[h, d], channel h * 16 + d
        if (lane < 8) s8 += qk;
        if (lane < 4) a.q[tok * 4 + lane] = qk;
    }
    red[w][lane] = skv;
    if (lane < 8) red[w][64 + lane] = s8;
    __syncthreads();
    if (tid < SROW) a.part[((size_t)line * a.nchunk + ch) * SROW + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

// The same statistics with the projections on the matrix cores (v_mfma_f64_16x16x4_f64; layouts as kp_ffn_mfma below):
// one wave = 16 elements of the line at a time, lane (g, j) holds channels 16 g ... 16 g + 15 of element j.  The fused
// 72 x 64 projection [Wv; Wq; Wk] is five 16-row tiles: tile T < 4 is head T of v (register r of lane (g, j) = channel
// 16 T + g + 4 r), tile 4 leaves q[g] in register 0 and k[g] in register 1 of lane (g, j).  Each lane accumulates the
// contributions of its own elements (k'[T] comes from lane (T, j) by one shuffle); the 16 element lanes of a lane group
// are summed once per wave at the end, the four waves in fixed order.  Block = (line, chunk of CHUNK_MFMA elements).
typedef double d4s __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(PT) kp_attn_stats_mfma(StatsArgs a) {
    __shared__ double red[4][SROW];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, j = lane & 15;
    const int line = blockIdx.x / a.nchunk, ch = blockIdx.x - line * a.nchunk;
    const int nelem = a.col ? a.P : a.L;
    const double* gam = a.w.g + 16 * g;       // (re-read per tile from L1: 64 registers are worth more than 32 loads)
    const double* bet = a.w.b + 16 * g;
    const double bq = a.w.bqk[g], bk = a.w.bqk[4 + g];
    d4s skv[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) skv[T] = d4s{0.0, 0.0, 0.0, 0.0};
    double sq = 0.0, sk = 0.0;
    const int e_end = min(nelem, (ch + 1) * CHUNK_MFMA);
    for (int e0 = ch * CHUNK_MFMA + 16 * w; e0 < e_end; e0 += 64) {
        const int e = e0 + j;
        const bool valid = e < e_end;
        const size_t tok = token_of(a.col, line, valid ? e : e0, a.P, a.L);
        double xn[16];
        {
            const double* p = a.x + tok * E + 16 * g;
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < 16; ++m) { xn[m] = p[m]; s += xn[m]; }
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            const double mu = s * (1.0 / 64.0);
            double v = 0.0;
#pragma unroll
            for (int m = 0; m < 16; ++m) { xn[m] -= mu; v = fma(xn[m], xn[m], v); }
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const double sd = sqrt(v * (1.0 / 64.0) + 1e-5);
#pragma unroll
            for (int m = 0; m < 16; ++m) xn[m] = xn[m] / sd * gam[m] + bet[m];
        }
        d4s qk = d4s{bq, bk, 0.0, 0.0};
        {
            const double* a4 = a.w.a72 + (size_t)4 * 16 * 64 + lane;
#pragma unroll 4
            for (int s = 0; s < 16; ++s) qk = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[s * 64], xn[s], qk, 0, 0, 0);
        }
        const double qp = valid ? elu1(qk[0]) : 0.0, kp = valid ? elu1(qk[1]) : 0.0;     // q'[g], k'[g] of element j
        sq += qp;
        sk += kp;
        if (valid) a.q[tok * 4 + g] = qp;
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            d4s v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = a.w.bv[16 * T + g + 4 * r];
            const double* aT = a.w.a72 + (size_t)T * 16 * 64 + lane;
#pragma unroll 4
            for (int s = 0; s < 16; ++s) v = __builtin_amdgcn_mfma_f64_16x16x4f64(aT[s * 64], xn[s], v, 0, 0, 0);
            const double kT = __shfl(kp, 16 * T + j, 64);          // k'[T] of element j (0 for an element past the end)
#pragma unroll
            for (int r = 0; r < 4; ++r) skv[T][r] = fma(kT, v[r], skv[T][r]);   // attention.py:187-188, channel 16 T + g + 4 r
        }
    }
    // sum over the 16 element lanes of each lane group, then over the four waves (fixed order)
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 4; ++r) skv[T][r] += __shfl_xor(skv[T][r], m, 64);
        sq += __shfl_xor(sq, m, 64);
        sk += __shfl_xor(sk, m, 64);
    }
    if (j == 0) {
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[w][16 * T + g + 4 * r] = skv[T][r];
        red[w][64 + g] = sq;
        red[w][68 + g] = sk;
    }
    __syncthreads();
    if (tid < SROW) a.part[((size_t)line * a.nchunk + ch) * SROW + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

// part[line][nchunk][72] -> stats[line][72], chunks in index order
__global__ void __launch_bounds__(PT) kp_stats_fin(const double* part, double* stats, int nlines, int nchunk) {
    const int i = blockIdx.x * PT + threadIdx.x;
    if (i >= nlines * SROW) return;
    const int line = i / SROW, j = i - line * SROW;
    double s = 0.0;
    for (int c = 0; c < nchunk; ++c) s += part[((size_t)line * nchunk + c) * SROW + j];
    stats[i] = s;
}

// ---- attention apply (attention.py:183-195) + residual --------------------------------------------
// o[h, d] = q'[h] / (S_q[h] / count) * S_kv[h, d] / S_k[h];  y = Wo o + bo;  x += y.
// The line's mix M[h][c] = sum_d Wo[c][16 h + d] ctx[16 h + d], ctx = S_kv / S_k / (S_q / count), is formed once
// per block; a token then costs four multiply-adds per channel.
__global__ void __launch_bounds__(PT) kp_attn_apply(ApplyArgs a) {
    __shared__ double ctx[E];
    __shared__ double M[NH][E];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int line = blockIdx.x / a.nchunk, ch = blockIdx.x - line * a.nchunk;
    const int nelem = a.col ? a.P : a.L;
    const double* s = a.stats + (size_t)line * SROW;
    if (tid < E) ctx[tid] = s[tid] / s[68 + (tid >> 4)] / (s[64 + (tid >> 4)] / a.count);
    __syncthreads();
    {
        double m = 0.0;                   // thread (h = w, c = lane)
#pragma unroll
        for (int d = 0; d < 16; ++d) m = fma(a.w.woT[(size_t)(16 * w + d) * E + lane], ctx[16 * w + d], m);
        M[w][lane] = m;
    }
    __syncthreads();
    const double bo = a.w.bo[lane];
    const int e_end = min(nelem, (ch + 1) * CHUNK);
    for (int e = ch * CHUNK + w; e < e_end; e += 4) {
        const size_t tok = token_of(a.col, line, e, a.P, a.L);
        const double* q = a.q + tok * 4;
        double y = bo;
#pragma unroll
        for (int h = 0; h < NH; ++h) y = fma(q[h], M[h][lane], y);
        a.x[tok * E + lane] += y;
    }
}

// ---- feed-forward (model.py:69-85, 101-104): x += W2 gelu_erf(W1 LN(x) + b1) + b2 -----------------------
__global__ void __launch_bounds__(PT) kp_ffn(FfnArgs a) {
    __shared__ double xn[FFN_NT][E];
    __shared__ double hid[FFN_NT][FF];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t t0 = (size_t)blockIdx.x * FFN_NT;
    const double g = a.w.g[lane], beta = a.w.b[lane];
    for (int t = w; t < FFN_NT; t += 4) {
        const size_t tok = t0 + t;
        const double xv = tok < a.ntok ? a.x[tok * E + lane] : 0.0;
        xn[t][lane] = layer_norm(xv, g, beta);
    }
    __syncthreads();
    {
        double acc[FFN_NT];
        const double b1 = a.w.b1[tid];
#pragma unroll
        for (int t = 0; t < FFN_NT; ++t) acc[t] = b1;
        for (int k = 0; k < E; ++k) {
            const double wk = a.w.w1T[(size_t)k * FF + tid];
#pragma unroll
            for (int t = 0; t < FFN_NT; ++t) acc[t] = fma(wk, xn[t][k], acc[t]);
        }
#pragma unroll
        for (int t = 0; t < FFN_NT; ++t) hid[t][tid] = 0.5 * acc[t] * (1.0 + erf(acc[t] * 0.70710678118654752440));   // nn.GELU(): erf form
    }
    __syncthreads();
    {
        // thread (c = lane, token pair w, w + 4)
        double y0 = a.w.b2[lane], y1 = y0;
        for (int j = 0; j < FF; ++j) {
            const double wj = a.w.w2T[(size_t)j * E + lane];
            y0 = fma(wj, hid[w][j], y0);
            y1 = fma(wj, hid[w + 4][j], y1);
        }
        if (t0 + w < a.ntok) a.x[(t0 + w) * E + lane] += y0;
        if (t0 + w + 4 < a.ntok) a.x[(t0 + w + 4) * E + lane] += y1;
    }
}

// ---- the same on the matrix cores: v_mfma_f64_16x16x4_f64 -------------------------------------------------
// One wave = 16 tokens on the N side.  Lane (g = lane >> 4, j = lane & 15) holds the 16 channels 16 g ... 16 g + 15
// of token j - 128 contiguous bytes of x.  Operand layouts (cdna_hip_programming.md): A[i][k] in lane i + 16 k,
// B[k][j] in lane j + 16 k, D[row = g + 4 r][col = j] in register r of lane (g, j).  The K order of each product and
// the row order of each output tile are free, and chosen so that activations never move between lanes:
//   GEMM1  K step s takes channel 16 kq + s from lane group kq            -> B operand = the lane's own xn[s]
//   GEMM1  D tile T: register r of lane (g, j) = hidden unit 16 T + g + 4 r
//   GEMM2  K step (T, r) takes hidden unit 16 T + kq + 4 r from lane group kq  -> B operand = GELU of D register r
//   GEMM2  D tile Tc: row i stands for channel 16 (i & 3) + 4 Tc + (i >> 2), so register r of lane (g, j) is channel
//          16 g + 4 Tc + r: the residual's own layout
// (the A fragments are packed accordingly on the host, pf_precise_host.hip.h).  512 MFMAs per 16 tokens; the
// 256 erf evaluations per token run on the VALU beside another wave's MFMAs.
// erf-GELU in double without ocml's erf (four divergent ranges, ~2,000 cycles per wave: it was 70 % of the FFN kernel):
//   gelu(h) = max(h, 0) - |h| Q(|h|),  Q(u) = erfc(u / sqrt 2) / 2 = exp(-u^2 / 2) R(u),
//   R(u) (1 + u) = a degree-22 polynomial in t = (u - 4) / (u + 4)  (Chebyshev fit on u in [0, inf), |relative error|
//   of R <= 2.4e-15, coefficients generated with scipy's erfcx; |gelu error| <= 1.8e-15 over |h| <= 40 against
//   0.5 h (1 + erf(h / sqrt 2)) evaluated in double).  Branch-free: one division, one exp, 23 FMAs.
__device__ __forceinline__ double gelu_f64(double h) {
    constexpr double Q[23] = {0x1.e361ea6fba145p-2, -0x1.8c18f2086e47cp-4, 0x1.cabd72a6120b9p-7, 0x1.d4969f10f90d4p-6,
                              -0x1.07c3c25842975p-5, 0x1.25dd720375999p-6, -0x1.47d5fc6944b2cp-8, -0x1.2b7f5644197fap-12,
                              0x1.6c5380196e928p-11, -0x1.8c1283b1235e3p-14, -0x1.707b3dae24d79p-14, 0x1.64919115d4a57p-16,
                              0x1.c9344f4725c3dp-17, -0x1.cdc5363466f39p-19, -0x1.5e69413cc4adcp-19, 0x1.c1cd90ff96cf7p-22,
                              0x1.26fb2b6228421p-21, -0x1.005dbc607bf3dp-26, -0x1.d50a583370aa4p-24, -0x1.1cca57b6a492fp-27,
                              0x1.241e7aeedd9aap-26, 0x1.b830e247ca68bp-30, -0x1.925d735408ab7p-30};
    const double u = fabs(h);
    const double r = 1.0 / ((u + 4.0) * (u + 1.0));
    const double t = (u - 4.0) * (u + 1.0) * r;
    double p = Q[22];
#pragma unroll
    for (int k = 21; k >= 0; --k) p = fma(p, t, Q[k]);
    const double q = exp(-0.5 * u * u) * p * (u + 4.0) * r;
    return fmax(h, 0.0) - u * q;
}

typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(PT) kp_ffn_mfma(FfnArgs a) {
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const size_t tok = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + j;
    const bool valid = tok < a.ntok;
    double x[16], xn[16];
    {
        const double* p = a.x + (valid ? tok : 0) * E + 16 * g;
#pragma unroll
        for (int m = 0; m < 16; ++m) x[m] = valid ? p[m] : 0.0;
    }
    {   // nn.LayerNorm(64): the token's channels sit in the four lanes (g, j), g = 0..3
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < 16; ++m) s += x[m];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const double mu = s * (1.0 / 64.0);
        double v = 0.0;
#pragma unroll
        for (int m = 0; m < 16; ++m) { xn[m] = x[m] - mu; v = fma(xn[m], xn[m], v); }
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        const double sd = sqrt(v * (1.0 / 64.0) + 1e-5);
#pragma unroll
        for (int m = 0; m < 16; ++m) xn[m] = xn[m] / sd * a.w.g[16 * g + m] + a.w.b[16 * g + m];
    }
    d4 y[4];
#pragma unroll
    for (int tc = 0; tc < 4; ++tc)
#pragma unroll
        for (int r = 0; r < 4; ++r) y[tc][r] = a.w.b2[16 * g + 4 * tc + r];
    // The A fragments of one hidden tile T (16 of W1 + 16 of W2 = 16 KB) are the same for the block's four waves: they
    // are staged through LDS, double-buffered - every thread fetches 8 doubles of tile T + 1 while tile T is consumed.
    // (Read straight from L2 by every wave the fragments were 16 KB per TOKEN of L2 traffic: 6.5 TB/s, the kernel's bound.)
    __shared__ double frag[2][2 * 16 * 64];
    const int tid = threadIdx.x;
    {
        const double *s1 = a.w.a1, *s2 = a.w.a2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { frag[0][tid + 256 * k] = s1[tid + 256 * k]; frag[0][1024 + tid + 256 * k] = s2[tid + 256 * k]; }
    }
    __syncthreads();
    for (int T = 0; T < 16; ++T) {
        double n1[4], n2[4];
        const int Tn = min(T + 1, 15);
        {
            const double *s1 = a.w.a1 + (size_t)Tn * 1024, *s2 = a.w.a2 + (size_t)Tn * 1024;
#pragma unroll
            for (int k = 0; k < 4; ++k) { n1[k] = s1[tid + 256 * k]; n2[k] = s2[tid + 256 * k]; }
        }
        const double* f = frag[T & 1];
        d4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = a.w.b1[16 * T + g + 4 * r];
#pragma unroll
        for (int s = 0; s < 16; ++s) h = __builtin_amdgcn_mfma_f64_16x16x4f64(f[s * 64 + lane], xn[s], h, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double act = gelu_f64(h[r]);                                                   // nn.GELU(): erf form
#pragma unroll
            for (int tc = 0; tc < 4; ++tc)
                y[tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[1024 + (r * 4 + tc) * 64 + lane], act, y[tc], 0, 0, 0);
        }
        double* fn = frag[(T + 1) & 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) { fn[tid + 256 * k] = n1[k]; fn[1024 + tid + 256 * k] = n2[k]; }
        __syncthreads();
    }
    if (valid) {
        double* p = a.x + tok * E + 16 * g;
#pragma unroll
        for (int tc = 0; tc < 4; ++tc)
#pragma unroll
            for (int r = 0; r < 4; ++r) p[4 * tc + r] = x[4 * tc + r] + y[tc][r];
    }
}

// ---- head (model.py:158-164, 182-185): per pair, sum over this rank's sites of softplus(w . x + b) ------
__global__ void __launch_bounds__(PT) kp_head(HeadArgs a) {
    const int lane = threadIdx.x & 63, line = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (line >= a.nlines) return;
    const double hw = a.hw[lane], hb = a.hb[0];
    double acc = 0.0;
    for (int l = 0; l < a.L; ++l) {
        const double z = wave_sum(hw * a.x[((size_t)line * a.L + l) * E + lane]) + hb;
        acc += z > 20.0 ? z : log1p(exp(z));                      // nn.Softplus(beta = 1, threshold = 20)
    }
    if (lane == 0) a.osum[line] = acc;
}
__global__ void __launch_bounds__(PT) kp_out(const double* osum, float* out, int n, double l_total) {
    const int i = blockIdx.x * PT + threadIdx.x;
    if (i < n) out[i] = (float)(osum[i] / l_total);               // model.py:185: mean over ALL sites
}
__global__ void __launch_bounds__(PT) kp_accumulate(double* dst, const double* src, size_t n) {   // shard emulation
    const size_t i = (size_t)blockIdx.x * PT + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
__global__ void __launch_bounds__(PT) kp_to_float(const double* src, float* dst, size_t n) {      // debug taps
    const size_t i = (size_t)blockIdx.x * PT + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}


#define PFP_GRID(n) dim3((unsigned)(((n) + PT - 1) / PT))
void launch_embed(hipStream_t s, size_t grid, const EmbedArgs& a) { hipLaunchKernelGGL(kp_embed, dim3((unsigned)grid), dim3(PT), 0, s, a); }
void launch_attn_stats(hipStream_t s, size_t grid, const StatsArgs& a, bool valu) {
    if (valu) hipLaunchKernelGGL(kp_attn_stats, dim3((unsigned)grid), dim3(PT), 0, s, a);
    else hipLaunchKernelGGL(kp_attn_stats_mfma, dim3((unsigned)grid), dim3(PT), 0, s, a);
}
void launch_stats_fin(hipStream_t s, const double* part, double* stats, int nlines, int nchunk) {
    hipLaunchKernelGGL(kp_stats_fin, PFP_GRID((size_t)nlines * SROW), dim3(PT), 0, s, part, stats, nlines, nchunk);
}
void launch_attn_apply(hipStream_t s, size_t grid, const ApplyArgs& a) { hipLaunchKernelGGL(kp_attn_apply, dim3((unsigned)grid), dim3(PT), 0, s, a); }
void launch_ffn(hipStream_t s, const FfnArgs& a, bool valu) {
    if (valu) hipLaunchKernelGGL(kp_ffn, dim3((unsigned)((a.ntok + FFN_NT - 1) / FFN_NT)), dim3(PT), 0, s, a);
    else hipLaunchKernelGGL(kp_ffn_mfma, dim3((unsigned)((a.ntok + 63) / 64)), dim3(PT), 0, s, a);
}
void launch_head(hipStream_t s, const HeadArgs& a) { hipLaunchKernelGGL(kp_head, dim3((unsigned)((a.nlines + 3) / 4)), dim3(PT), 0, s, a); }
void launch_out(hipStream_t s, const double* osum, float* out, int n, double l_total) {
    hipLaunchKernelGGL(kp_out, PFP_GRID((size_t)n), dim3(PT), 0, s, osum, out, n, l_total);
}
void launch_accumulate(hipStream_t s, double* dst, const double* src, size_t n) { hipLaunchKernelGGL(kp_accumulate, PFP_GRID(n), dim3(PT), 0, s, dst, src, n); }
void launch_to_float(hipStream_t s, const double* src, float* dst, size_t n) { hipLaunchKernelGGL(kp_to_float, PFP_GRID(n), dim3(PT), 0, s, src, dst, n); }

}  // namespace pfp
