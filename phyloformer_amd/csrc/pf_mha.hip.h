// Softmax multi-head self-attention (the reference's MultiHeadAttention, phyloformer/attention.py:53-91)
// for gfx950: split-fp16 MFMA for every contraction (two fp16 limbs = 22 significant bits, 3 passes - the Q-K path
// that feeds the exponent included: rounds 1-5 needed three bf16 terms / 6 passes there, because two bf16 limbs carry
// 16 bits), fp32 online softmax, K/V staged through LDS.
//
// the class is dead code in the reference (nothing instantiates it and no
// checkpoint fits it), so this op is NOT on the graded distance path; it exists because the north star
// names softmax QK^T / PV attention.  Oracle: oracle/mha_oracle.py pinned to outputs of the reference
// class (tests/golden/mha.npz).
//
// x [B][R][C][64] fp32, attention along C for every (b, r) and head (H = 4, D = 16).  Three kernels:
//
//   k_mha_qkv    token tiles of 32 -> Q (pre-scaled by log2(e)/sqrt(D)), K as MFMA operand fragments
//                [row][head][token][kgrp] (hi plane + lo plane), and V *transposed* as A-operand
//                fragments [row][head][key tile][d][kstep][kgrp]: V is computed with the token tile as
//                the A operand (D[m = token][n = channel]), so a lane ends up holding 16 keys of one
//                channel in exactly the K order the P fragments of k_mha_attn have — no transpose pass.
//   k_mha_attn   one wave per (row, head, 32-query tile), four query tiles per workgroup; key/value
//                fragments of 128 keys per stage are staged through LDS (double-buffered) and shared by
//                the four waves.  S^T = K Q^T (keys on M, queries on N: a lane owns 16 scores of ONE
//                query, so the row max / row sum are in-lane reductions plus one permlane32 swap),
//                p = 2^(s - m) online, P split hi/lo in registers -> B operand of O^T += V^T P^T.
//   k_mha_out    out_proj on 32-token tiles, fp32 result.
//
// MFMA shapes: v_mfma_f32_32x32x16_f16 for the projections and QK^T (D = 16 fills its K exactly);
// v_mfma_f32_16x16x32_f16 for PV (M = the head's 16 channels, K = the 32 keys of a tile, two N = 16 query
// halves): the P fragments reach its B layout with one v_permlane16_swap per register pair.
#pragma once
#include <hip/hip_runtime.h>

namespace pfk {

constexpr int MHA_H = 4;
constexpr int MHA_D = 16;
constexpr int MHA_WFRAGS = 2 * 4 * 2 * 64;   // fragments of one packed 64x64 matrix, hi + lo (16 KB)
constexpr int MHA_OFF_WK = MHA_WFRAGS, MHA_OFF_WV = 2 * MHA_WFRAGS, MHA_OFF_WO = 3 * MHA_WFRAGS;
constexpr int MHA_WTOTAL = 4 * MHA_WFRAGS;
constexpr int MHA_KB = 4;                    // key tiles per LDS stage

struct MhaArgs {
    const float* x;        // [rows][C][64]
    float* y;              // [rows][C][64]
    float* att;            // [rows][C][64]  softmax(QK^T)V, heads concatenated
    const frag_t* wfrag;   // Wq, Wk, Wv, Wo as hi / lo fragments: MHA_WTOTAL fragments
    const float* bias;     // [4][64]
    frag_t* qp;            // [2 planes: hi, lo][rows][H][Cpad][2]
    frag_t* kp;            // same
    frag_t* vp;            // [2 planes][rows][H][ntiles][16 d][2 kstep][2 kgrp]
    int rows, C, ntiles;   // rows = B*R, ntiles = ceil(C/32), Cpad = 32*ntiles
    float qscale;          // log2(e) / sqrt(D)
};

// Launchers (pf_mha.hip is its own translation unit, compiled with hipcc's default scheduling strategy: the
// iterative-ILP strategy pf_lib.hip is built with for k_main's sake crashes the register allocator on k_mha_qkv).
// Asynchronous on `s`; grids as the kernels' headers say.
void launch_mha_qkv(hipStream_t s, unsigned grid, const MhaArgs& a);
void launch_mha_attn(hipStream_t s, unsigned grid, const MhaArgs& a);
void launch_mha_out(hipStream_t s, unsigned grid, const MhaArgs& a);

}  // namespace pfk
