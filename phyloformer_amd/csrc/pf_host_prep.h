// Host-side preparation of the device operands: 16-bit hi/lo MFMA fragment images (fp16; bf16 with PF_F16 = 0), LayerNorm folding, the block-0
// residue-pair table, and the shape-only launch plans (k_main's tiling, k_colstats' summation tree).
// Plain C++ (no HIP): included by pf_lib.hip, and compiled on its own with g++ -fsanitize=address,undefined for the
// fuzz tests (tests/test_native_sanitizers.py, tests/native/pf_host_prep_shim.cpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "pf_layout.h"

namespace pfhost {
using namespace pfk;

// ---- bf16 helpers (host) --------------------------------------------------------------------
inline uint16_t f2bf(float f) {  // round to nearest even
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return uint16_t((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return uint16_t(u >> 16);
}
inline float bf2f(uint16_t b) {
    uint32_t u = uint32_t(b) << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// ---- IEEE half helpers (host): round to nearest even, subnormals kept, overflow -> inf ----------
inline uint16_t f2h(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    const uint16_t sign = uint16_t((u >> 16) & 0x8000u);
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) return uint16_t(sign | 0x7c00u | (u > 0x7f800000u ? 0x200u : 0u));   // inf / nan
    if (u >= 0x477ff000u) return uint16_t(sign | 0x7c00u);          // >= 65520 rounds to inf
    if (u < 0x33000001u) return sign;                                // <= 2^-25 rounds to zero (2^-25 itself: ties to even)
    const int e = int(u >> 23) - 127;                                // unbiased exponent
    uint32_t m = (u & 0x7fffffu) | 0x800000u;                        // 24-bit significand
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;                   // bits dropped (subnormal results drop more)
    const uint32_t half = 1u << (shift - 1), rest = m & ((1u << shift) - 1);
    uint32_t r = m >> shift;
    if (rest > half || (rest == half && (r & 1u))) ++r;
    // r carries the implicit bit for normal results: exponent field = e + 15 - 1, added with the carry
    const uint32_t out = (e < -14) ? r : (uint32_t(e + 14) << 10) + r;
    return uint16_t(sign | out);
}
inline float h2f(uint16_t hbits) {
    const uint32_t sign = uint32_t(hbits & 0x8000u) << 16;
    const int e = (hbits >> 10) & 0x1f;
    const uint32_t m = hbits & 0x3ffu;
    float mag;
    if (e == 0) mag = std::ldexp((float)m, -24);
    else if (e == 31) mag = m ? NAN : INFINITY;
    else mag = std::ldexp((float)(m | 0x400u), e - 25);
    uint32_t u;
    std::memcpy(&u, &mag, 4);
    u |= sign;
    std::memcpy(&mag, &u, 4);
    return mag;
}
// the format the device's split operands use (pf_layout.h: PF_F16)
#if PF_F16
inline uint16_t f2x(float f) { return f2h(f); }
inline float x2f(uint16_t b) { return h2f(b); }
#else
inline uint16_t f2x(float f) { return f2bf(f); }
inline float x2f(uint16_t b) { return bf2f(b); }
#endif
inline int kmap_h(int j, int h) { return 8 * (j >> 2) + 4 * h + (j & 3); }

// Pack W[M][K] (row-major fp32) into MFMA A fragments, hi/lo split:
//   out[((T * (K/16) + s) * 2 + hl) * 64 + lane][i] = W[32T + (lane & 31)][kmap(8s + i, lane >> 5)]
// (rows >= M are zero).  K order matches the lane ownership of the B operand.
inline void pack_frags(const float* W, int M, int K, int Mpad, uint16_t* out, float scale = 1.f) {
    const int nT = Mpad / 32, nS = K / 16;
    for (int T = 0; T < nT; ++T)
        for (int s = 0; s < nS; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                    const int m = 32 * T + (lane & 31), k = kmap_h(8 * s + i, lane >> 5);
                    const float w = (m < M) ? W[(size_t)m * K + k] * scale : 0.f;     // (scale: a power of two)
                    const uint16_t hi = f2x(w);
                    const uint16_t lo = f2x(w - x2f(hi));
                    const size_t base = ((size_t)(T * nS + s) * 2) * 64;
                    out[(base + lane) * 8 + i] = hi;
                    out[(base + 64 + lane) * 8 + i] = lo;
                }
}

// Rows 0..3 = Wq', 4..7 = Wk' (folded, [4][64] each) as MFMA A fragments of an 8-row tile:
//   out[((s * 2 + hi/lo) * 16 + kgrp * 8 + row) * 8 + i], 128 fragments of 16 bytes
inline void pack_qk_frags(const float* wq, const float* wk, uint16_t* qk) {
    for (int s = 0; s < 4; ++s)
        for (int kg = 0; kg < 2; ++kg)
            for (int m = 0; m < 8; ++m)
                for (int i = 0; i < 8; ++i) {
                    const int k = kmap_h(8 * s + i, kg);
                    const float w = (m < 4) ? wq[(size_t)m * E + k] : wk[(size_t)(m - 4) * E + k];
                    const uint16_t hi = f2x(w), lo = f2x(w - x2f(hi));
                    qk[(((size_t)s * 2 + 0) * 16 + kg * 8 + m) * 8 + i] = hi;
                    qk[(((size_t)s * 2 + 1) * 16 + kg * 8 + m) * 8 + i] = lo;
                }
}

// Row-statistics operands of one block's row attention, in the layout of the LDS image tail:
//   [FRAG_WV ..): Wv' hi fragments [2 T][4 s][64];  [FRAG_QK ..): rows 0..7 of [Wq';Wk'] as
//   [4 s][2 hi/lo][2 kgrp][8 rows];  wv_lo: Wv' lo fragments [2 T][4 s][64] (stays in global).
inline void pack_row_stats(const float* wv, const float* wq, const float* wk, uint16_t* img_tail /* from FRAG_WV */,
                    uint16_t* wv_lo) {
    std::vector<uint16_t> full((size_t)2 * 4 * 2 * 64 * 8);
    pack_frags(wv, E, E, E, full.data());
    for (int T = 0; T < 2; ++T)
        for (int s = 0; s < 4; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                    const size_t src = (((size_t)(T * 4 + s) * 2) * 64 + lane) * 8 + i;
                    const size_t dst = ((size_t)(T * 4 + s) * 64 + lane) * 8 + i;
                    img_tail[dst] = full[src];
                    wv_lo[dst] = full[src + 64 * 8];
                }
    pack_qk_frags(wq, wk, img_tail + (size_t)(FRAG_QK - FRAG_WV) * 8);
}

struct AttnHost {
    const float *g, *b, *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo;
};

// fold the LayerNorm affine into a projection: W' = W diag(g), b' = b + W beta  (double accumulate)
inline void fold(const float* W, const float* bias, const float* g, const float* beta, int M, int K,
          std::vector<float>& Wf, std::vector<float>& bf) {
    Wf.resize((size_t)M * K);
    bf.resize(M);
    for (int m = 0; m < M; ++m) {
        double acc = bias ? bias[m] : 0.0;
        for (int k = 0; k < K; ++k) {
            Wf[(size_t)m * K + k] = (float)((double)W[(size_t)m * K + k] * (double)g[k]);
            acc += (double)W[(size_t)m * K + k] * (double)beta[k];
        }
        bf[m] = (float)acc;
    }
}

// Row-attention quantities of block 0 for every residue pair (a, b), in double precision:
//   x0 = T[a] + T[b] (fp32 sum, as the device forms it), xn = LayerNorm(x0) g + beta,
//   q' = elu(Wq xn + bq) + 1, k' likewise, v = Wv' LN(x0) without any bias (the folded bias is added in
//   k_rowfin);  row = [ k'[c >> 4] v[c] (64) | q' (4) | k' (4) ]  -> k_embed sums rows over the sites.
inline void build_pair_table(const float* table, const AttnHost& r, std::vector<float>& out) {
    out.assign((size_t)PAIRTAB_ROWS * PAIRTAB_W, 0.f);
    auto elu1 = [](double z) { return z > 0 ? z + 1.0 : std::exp(z); };
    for (int a = 0; a < NA; ++a)
        for (int b = 0; b < NA; ++b) {
            double x[E], xn[E], mean = 0, var = 0;
            for (int c = 0; c < E; ++c) { x[c] = (double)(table[a * E + c] + table[b * E + c]); mean += x[c]; }
            mean /= E;
            for (int c = 0; c < E; ++c) var += (x[c] - mean) * (x[c] - mean);
            const double rstd = 1.0 / std::sqrt(var / E + (double)LN_EPS);
            for (int c = 0; c < E; ++c) xn[c] = (x[c] - mean) * rstd;
            double q[NH], k[NH];
            for (int hh = 0; hh < NH; ++hh) {
                double zq = r.bq[hh], zk = r.bk[hh];
                for (int c = 0; c < E; ++c) {
                    const double y = xn[c] * (double)r.g[c] + (double)r.b[c];
                    zq += (double)r.wq[hh * E + c] * y;
                    zk += (double)r.wk[hh * E + c] * y;
                }
                q[hh] = elu1(zq);
                k[hh] = elu1(zk);
            }
            float* row = out.data() + (size_t)(a * NA + b) * PAIRTAB_W;
            for (int co = 0; co < E; ++co) {
                double v = 0;
                for (int c = 0; c < E; ++c) v += (double)r.wv[co * E + c] * (double)r.g[c] * xn[c];
                row[co] = (float)(k[co >> 4] * v);
            }
            for (int hh = 0; hh < NH; ++hh) { row[64 + hh] = (float)q[hh]; row[68 + hh] = (float)k[hh]; }
        }
}

// How k_main cuts an alignment's P x Lloc tokens into 32-token tiles (pf_device.hip.h, tile_pos): a function of
// the shape only.  Flat tiling - tiles of 32 consecutive tokens that may cover the end of one pair row and the
// start of the next - whenever rows are at least one tile long and not a whole number of tiles; otherwise every
// row has its own ceil(Lloc / 32) tiles.
struct TilePlan { int flat, nt_aln, slots_aln; };
inline TilePlan tile_plan_core(int P, int Lloc, int tile_force) {
    TilePlan t;
    const int ntiles = (Lloc + 31) / 32;
    // Flat tiling saves the ragged last tile of every row (a fraction `waste` of all MFMA columns) and pays for a
    // tile over two rows - one tile per row - about a tenth of a tile (6 MFMAs, a fragment reload, a second masked
    // reduction): worth it at L = 500 (2.3 % against 0.6 %) or 200 (10.7 % against 1.6 %), not at L = 63 (1.6 %
    // against 5 %: a site-sharded rank at world = 8 - measured 4.33 against 4.20 ms per k_main launch) and a wash
    // at L = 125 (4.16 against 4.15 ms).
    const double waste = (double)(ntiles * 32 - Lloc) / (ntiles * 32), straddle = 0.10 * 32.0 / std::max(Lloc, 1);
    t.flat = (Lloc >= 32 && waste > straddle && tile_force != 0) ? 1 : 0;
    if (tile_force == 1 && Lloc >= 32 && Lloc % 32 != 0) t.flat = 1;      // A/B runs (PF_FLAT_TILES)
    t.nt_aln = t.flat ? (int)(((long)P * Lloc + 31) / 32) : P * ntiles;
    t.slots_aln = t.flat ? t.nt_aln + P : t.nt_aln;
    return t;
}

// Pair groups of k_colstats.  Chosen from the alignment's shape only - never from the batch size - so that
// the association of the pair sums, and with it every output bit, is the same whatever batch an alignment
// travels in (a lone 60 x 500 alignment still yields 128 blocks).
inline int colstats_groups(int /*B*/, int P, int Lloc) {
    // A wave walks its group's pairs one after the other (0.7 us each): the walk length is the kernel's
    // latency for a lone alignment, but every extra group costs a 33 KB partial per 32-site chunk (HBM write
    // + read) and a pipeline fill.  Measured at 60 x 500 (tools/colstats_runs_compare.py), 8 / 16 / 32 groups:
    // 0.91 / 0.98 / 0.99 ms per launch at batch 16, 0.165 / 0.099 / 0.067 ms at batch 1.  The headline is the
    // batched rate: >= 128 blocks per alignment, <= 640 and >= 32 pairs per group.
    const int chunks = (Lloc + 31) / 32;
    int G = std::max((128 + chunks - 1) / chunks, (P + 639) / 640);
    G = std::min(G, std::max(1, P / 32));
    G = std::max(1, std::min(G, 32));
    while (G > 1 && (long)((P + G - 1) / G) * (G - 1) >= P) --G;   // no empty group
    return G;
}

// The whole plan: groups, the runs inside a group (shape only, like the groups: they fix the association of
// the pair sums) and - the one batch-dependent choice, which does not change a bit of the result - whether a
// block walks a group or a single run.  A lone alignment cannot fill the chip with groups (60 x 500: 128
// blocks of 222 pairs = 0.165 ms per launch); by runs it can (1,184 blocks of <= 32 pairs).
struct ColPlan { int G, sub, S, fine; };
inline ColPlan colstats_plan_core(int B, int P, int Lloc, int colstats_fine) {
    ColPlan plan{};
    ColPlan* w = &plan;
    w->G = colstats_groups(B, P, Lloc);
    const int per = (P + w->G - 1) / w->G;
    // (measured at 60 x 500: folding every 32 pairs costs the batched walk 1 %, every 64 nothing; a lone
    // alignment's launch takes 68 us either way, and k_colfin 18 / 11 us)
    // (runs of 8 for groups of 16 .. 47 pairs - round 4: a lone small alignment walks a run in 4 iterations instead
    // of 8, 20 x 200 batch 1 +6 %; groups of >= 48 pairs - every 60-sequence shape, sharded or not - keep 16, and a
    // group shorter than two runs of 8 stays one run)
    w->sub = per >= 128 ? 64 : per >= 64 ? 32 : (per >= 48 || per < 16) ? 16 : 8;
    w->S = (per + w->sub - 1) / w->sub;
    // by groups once they fill the chip's 512 resident blocks (2 per CU), by runs below that
    const long group_blocks = (long)B * ((Lloc + 31) / 32) * w->G;
    const bool auto_fine = w->S > 1 && group_blocks < 512;
    w->fine = colstats_fine < 0 ? (int)auto_fine : (colstats_fine && w->S > 1);
    return plan;
}

}  // namespace pfhost
