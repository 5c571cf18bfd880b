// The PRECISE path: the same forward (phyloformer/model.py:166-187) evaluated in float64, for the shapes on which
// the split-bf16 MFMA path cannot hold the north star's 1e-4.
//
// Why it exists.  Alignments of a handful of sites (or of 2-4 sequences) are nothing the model was trained on: the
// residual stream reaches |x| ~ 500, the distances 5-40, and the forward is ill-conditioned in fp32 itself - the
// fp32 reference is 3e-5 ... 7e-4 away from its own float64 evaluation there.  The default path's
// 2^-17-per-operand products add 1e-4 ... 2e-3 on top.  No fp32 formulation can promise to sit within a fixed bound
// of another fp32 formulation on such input (both are a rounding cloud around the exact value); float64 sits at
// the cloud's centre, so its distance to the reference is the reference's own rounding error and nothing else.
// The host selects this path from the alignment's SHAPE only (pf_lib.hip::use_precise), never from the batch, so
// an alignment gets the same bits wherever it travels.  Fixed-order sums, no atomics.
//
// Arithmetic follows the reference's own op order (un-collapsed LayerNorm affine, separate q / k / v / out
// projections: attention.py:163-195), on raw weights widened to double.  The dense contractions (fused V / q / k
// projection, FFN) run on v_mfma_f64_16x16x4_f64 - 78.6 TFLOP/s on MI355X, the fp64 VALU's own peak: the matrix cores
// do not add flops here, they take the operand traffic away (kp_attn_stats_mfma, kp_ffn_mfma in pf_precise.hip; the
// VALU kernels they replaced stay as cross-check, option "precise_ffn_valu").  32 TFLOP/s algorithmic, 3-9 x the
// default kernels' time (profiles/r05j_precise_bench.txt).
//
// Layout: xd [B][P][Lloc][64] double, token-major like the default path.  MFMA kernels: one wave = 16 tokens, lane
// (g = lane >> 4, j = lane & 15) holds channels 16 g ... 16 g + 15 of token j; VALU kernels: one wave = one token,
// lane = channel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pfp {

constexpr int E = 64, NH = 4, FF = 256, NA = 22, SROW = 72;
constexpr int PT = 256;            // threads per block (4 waves)
constexpr int CHUNK = 64;          // elements of the reduce axis per block (VALU statistics kernel, apply kernel)
constexpr int CHUNK_MFMA = 256;    // the same for the MFMA statistics kernel: four 16-token tiles per wave
constexpr int FFN_NT = 8;          // tokens per FFN block

// weights of one attention sub-block (device, double)
struct AttnW {
    const double *g, *b;       // LayerNorm affine [64]
    const double *wqk;         // [8][64]: rows 0..3 Wq, 4..7 Wk
    const double *bqk;         // [8]
    const double *wvT;         // [64 k][64 c] = Wv[c][k]
    const double *bv;          // [64]
    const double *woT;         // [64 hd][64 c] = Wo[c][hd]
    const double *bo;          // [64]
    // v_mfma_f64_16x16x4_f64 A fragments of the fused V / q / k projection (kp_attn_stats_mfma), lane = (i, kq):
    const double *a72;         // [5 T][16 s][64]: T < 4: Wv[16 T + i][16 kq + s];  T = 4: rows 0..3 Wq, 4..7 Wk, 8..15 zero
};
struct FfnW {
    const double *g, *b;       // [64]
    const double *w1T;         // [64 k][256 j] = W1[j][k]                 (kp_ffn, the VALU cross-check kernel)
    const double *b1;          // [256]
    const double *w2T;         // [256 j][64 c] = W2[c][j]
    const double *b2;          // [64]
    // v_mfma_f64_16x16x4_f64 A fragments (kp_ffn_mfma; one double per lane, lane = (i = lane & 15, kq = lane >> 4)):
    const double *a1;          // [16 T][16 s][64]: W1[16 T + i][16 kq + s]
    const double *a2;          // [16 T][4 r][4 Tc][64]: W2[16 (i & 3) + 4 Tc + (i >> 2)][16 T + kq + 4 r]
};

struct EmbedArgs {
    const uint8_t* idx; const int16_t *pi, *pj; const double* table; double* x;
    int B, N, P, L; unsigned* bad;
};

struct StatsArgs {
    const double* x; double* q; double* part; AttnW w;
    int col;            // 0: line = (b, p), elements = sites;  1: line = (b, l), elements = pairs
    int P, L, nchunk;
};

struct ApplyArgs {
    double* x; const double* q; const double* stats; AttnW w;
    int col, P, L, nchunk;
    double count;       // L_total (row attention) or P (column attention): q / q.mean(dim = -2)
};

struct FfnArgs { double* x; FfnW w; size_t ntok; };

struct HeadArgs { const double* x; const double* hw; const double* hb; double* osum; int nlines, L; };

// Launchers (pf_precise.hip is its own translation unit: hipcc's iterative-ILP scheduling strategy, which
// pf_lib.hip is built with for k_main's sake, crashes the register allocator on these kernels).
// grid = number of 256-thread blocks; asynchronous on `s`.
void launch_embed(hipStream_t s, size_t grid, const EmbedArgs& a);
void launch_attn_stats(hipStream_t s, size_t grid, const StatsArgs& a, bool valu);    // a.nchunk chunks of CHUNK (valu) / CHUNK_MFMA
void launch_stats_fin(hipStream_t s, const double* part, double* stats, int nlines, int nchunk);
void launch_attn_apply(hipStream_t s, size_t grid, const ApplyArgs& a);
void launch_ffn(hipStream_t s, const FfnArgs& a, bool valu);      // valu: the cross-check kernel instead of the MFMA one
void launch_head(hipStream_t s, const HeadArgs& a);
void launch_out(hipStream_t s, const double* osum, float* out, int n, double l_total);
void launch_accumulate(hipStream_t s, double* dst, const double* src, size_t n);
void launch_to_float(hipStream_t s, const double* src, float* dst, size_t n);

}  // namespace pfp
