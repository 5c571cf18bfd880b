// Shapes and memory-layout constants shared by the device kernels (pf_device.hip.h) and the host-side weight
// preparation (pf_host_prep.h).  Plain C++: no HIP types, so that the host code builds with g++ under
// AddressSanitizer / UBSan for the fuzz tests (tests/test_native_sanitizers.py).
#pragma once

// 16-bit format of the split MFMA operands: 1 = IEEE half (default), 0 = bfloat16 (rounds 1-5; A/B builds only).
// pf_device.hip.h explains the choice; pf_host_prep.h packs the weights accordingly.
#ifndef PF_F16
#define PF_F16 1
#endif

namespace pfk {

constexpr int E = 64;        // embed_dim
constexpr int NH = 4;        // heads
constexpr int HD = 16;       // head_dim
constexpr int FF = 256;      // FFN hidden
constexpr int NA = 22;       // alphabet
constexpr int SROW = 72;     // row statistics per pair
constexpr int MROW = 4 * 64; // folded row mix per pair: M[h][c], 4 heads (the bias row is the same for every pair:
                             // k_colstats keeps it in registers, the MFMA fragments carry it in K slot 4)
constexpr float LN_EPS = 1e-5f;

// LDS image of one block's MFMA A operands (fragments of 8 16-bit values, 16 B per lane):
//   W1' : [8 T][4 s][2 hi/lo][64 lanes]      64 KB   (FFN 64->256, LN affine folded)
//   W2  : [2 To][16 s][2][64]                64 KB   (FFN 256->64)
//   Woc : [2 To][4 s][2][64]                 16 KB   (column out_proj)
//   Wv' : [2 T][4 s][64] hi only              8 KB   (next block's row V projection; lo: below / L2)
//   Wqk : [4 s][2][16]                         2 KB   (next block's row q/k rows, 8 of 32 rows)
//   Wv' lo : [WVLO_LDS of 2 T x 4 s][64]       4 KB   (the first four of the eight lo fragments; the rest from L2)
//   consts (floats): b1'[256] | b2[64] | bqk[8] | head_w[64] | head_b[1] | pad | bo_col[64]
constexpr int FRAG_W1 = 0;                                   // [8 T][4 s][2 hi/lo][64]
constexpr int FRAG_W2 = FRAG_W1 + 8 * 4 * 2 * 64;            // [2 To][16 s][2][64]
constexpr int FRAG_WO = FRAG_W2 + 2 * 16 * 2 * 64;           // [2 To][4 s][2][64]
constexpr int FRAG_WV = FRAG_WO + 2 * 4 * 2 * 64;            // next row attn Wv' hi only: [2 T][4 s][64]
constexpr int FRAG_QK = FRAG_WV + 2 * 4 * 64;                // next row attn [Wq';Wk'] rows 0..7 only:
                                                             //   [4 s][2 hi/lo][2 kgrp][8 rows]
#ifndef PF_WVLO_LDS
#define PF_WVLO_LDS 4        // of the 8 Wv' lo fragments per lane, how many live in LDS (what fits: 4 KB of the 4,288 B left)
#endif
constexpr int WVLO_LDS = PF_WVLO_LDS;
constexpr int FRAG_WVLO = FRAG_QK + 4 * 2 * 16;              // the first WVLO_LDS of Wv' lo's [2 T][4 s] fragments: [..][64]
constexpr int FRAG_END = FRAG_WVLO + WVLO_LDS * 64;          // in fragment (16-byte) units
constexpr int WVLO_FRAGS = 2 * 4 * 64;                       // lo part of Wv', read from global
constexpr int CONST_B1 = 0, CONST_B2 = 256, CONST_BQK = 320, CONST_HW = 328, CONST_HB = 392,
              CONST_BOC = 400;
constexpr int CONST_LEN = 464;                                // floats
constexpr int MAIN_LDS_BYTES = FRAG_END * 16 + CONST_LEN * 4; // 163,648 B of the 163,840 B LDS (159,552 without Wv' lo)
constexpr int MFRAG_PER_PAIR = 2 * 2 * 32;                    // row-mix fragments per pair (lanes h=0)

constexpr int MAIN_THREADS = 512;  // 8 waves: two per SIMD so MFMA and VALU phases of different waves overlap
constexpr int MAIN_WAVES = MAIN_THREADS / 64;

constexpr int PAIRTAB_ROWS = 22 * 22, PAIRTAB_W = 72;

}  // namespace pfk
