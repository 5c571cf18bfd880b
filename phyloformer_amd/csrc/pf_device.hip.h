// Device kernels of the Phyloformer forward pass for gfx950 (MI355X, CDNA4).
//
// Data layout in HBM (all fp32 unless noted), Lloc = sites held by this rank:
//   x     [B][P][Lloc][64]   residual stream, token-major (pair, site, channel)
//   qrow  [B][P][Lloc][4]    q' = elu(q)+1 of the row attention of the next block to run
//   qcol  [B][P][Lloc][4]    q' of the column attention of the current block
//   srow  [B][P][72]         row statistics: S_kv[64] (no v-bias) | S_q[4] | S_k[4]
//   mrow  [B][P][4][64]      folded row mix  M[h][c], 4 heads (the bias row, equal for every pair, is not stored)
//   ctx   [B][Lloc][64]      column context ctx[h*16+d], normalised
//
// Algebra (reference: phyloformer/attention.py:160-197, model.py:87-106).  With
// x~ = LayerNorm without affine, and gamma/beta folded into the projections:
//   row block:  y[l,c] = bo[c] + sum_h q'[l,h] * M[h][c],
//               M[h][c] = sum_d Wo[c][16h+d] * (S_kv[16h+d] + bv'[16h+d] S_k[h]) / S_k[h] * L / S_q[h]
//   col block:  y[p,c] = bo[c] + sum_hd Wo[c][hd] * q'[p,h] * ctx[l][hd]
//               ctx[l][hd] = (sum_c Wv'[hd][c] Z[l][h][c] + bv'[hd] S_k[l][h]) / S_k[l][h] * P / S_q[l][h]
//               Z[l][h][c] = sum_p k'[p,l,h] x~[p,l,c]          (V projection pulled out of the sum)
// so the only per-token dense contractions left are the FFN (64->256->64), the
// column out-projection and the next block's row V/q/k projection: those run
// on MFMA as split-fp16 (hi*hi + lo*hi + hi*lo, fp32 accumulate; bf16 in rounds 1-5, PF_F16 below).
//
// MFMA tiling ("token tile" = 32 consecutive sites of one pair, one wave):
//   v_mfma_f32_32x32x16_f16, tokens on the N side.  Lane l = (t = l & 31, h = l >> 5)
//   owns, for token t, the 32 channels kmap(j, h) = 8*(j>>2) + 4*h + (j&3), j = 0..31.
//   This is at once the B-operand layout (8 consecutive j per K-step), the C/D
//   layout of a 32-row output tile (row = (r&3) + 8*(r>>2) + 4*h) and a 16-byte
//   granular global access pattern, so activations never move between lanes:
//   GEMM1's accumulators feed GEMM2's B operand directly (K order is permuted
//   identically in the pre-packed A fragments, see pack_frags() on the host).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pf_layout.h"

namespace pfk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// The 16-bit format of the split MFMA operands (x ~= hi + lo, products hi*hi + hi*lo + lo*hi in fp32):
//   PF_F16 = 1 (round 6, default): IEEE half.  Two limbs carry 22 significant bits (2^-22 relative; below 2^-3 the lo
//              limb is subnormal and the error an absolute 2^-25 - the matrix cores, v_cvt_pk_f16_f32 and
//              v_dot2c_f32_f16 all keep fp16 subnormals, tools/f16_probe.hip), i.e. fp32's own rounding level, at the
//              bf16 MFMA rate.  Range: |operand| < 65504 - guaranteed by construction, see "fp16 operand ranges" below.
//   PF_F16 = 0 (rounds 1-5, kept for A/B builds: tools/build_variant.py lib.so -DPF_F16=0): bfloat16, 16 bits in two
//              limbs (2^-17 relative), fp32's exponent range.
// (the switch itself lives in pf_layout.h: the host-side packing must agree)
#if PF_F16
typedef _Float16 h16_t;
#define PF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define PF_DOT2C "v_dot2c_f32_f16"
#define PF_CVT_PK "v_cvt_pk_f16_f32"
#define PF_NEG1_LO 0x0000bc00u      // (-1, 0) and (0, -1) as packed pairs of the format
#define PF_NEG1_HI 0xbc000000u
#else
typedef __bf16 h16_t;
#define PF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define PF_DOT2C "v_dot2c_f32_bf16"
#define PF_CVT_PK "v_cvt_pk_bf16_f32"
#define PF_NEG1_LO 0x0000bf80u
#define PF_NEG1_HI 0xbf800000u
#endif
typedef h16_t frag_t __attribute__((ext_vector_type(8)));      // one MFMA operand fragment: 8 K-values of a lane
typedef h16_t h16x2 __attribute__((ext_vector_type(2)));
// fp16 operand ranges (PF_F16): an operand that overflows to inf poisons a whole tile, so every split operand is
// bounded by construction - x~ (LayerNorm output) by sqrt(63); the FFN hidden activation and the row-mix base
// matrix by the checkpoint's weights (checked at pf_create, pf_lib.hip::check_f16_ranges: a checkpoint outside the
// range runs on the float64 kernels); the two data-dependent factors 1 / mean(q') are applied on the side where
// they meet q' itself, so that only q' / mean(q') <= (number of elements) is ever converted:
//   row mix:     A = M_base[h][c] * ROWMIX_A_SCALE (k_rowfin),  B = q'[l][h] * L / S_q[h] * ROWMIX_B_SCALE (k_main);
//   column apply: A = Wo * 2^4 (host),  B = q'_c[p][h] * ctx[l][hd] * 2^-4  (<= P * max|v| / 16 < 65504 for P <= 19,900).
constexpr float COLAPPLY_B_SCALE = PF_F16 ? 0.0625f : 1.f;     // 2^-4 on q' (x) ctx; Wo carries 2^4 (pf_lib.hip)
constexpr float COLAPPLY_A_SCALE = PF_F16 ? 16.f : 1.f;
// explicit LDS pointers: with an opaque per-lane base (PF_OPAQUE) every fragment / constant read
// becomes `ds_read_b128 v, base offset:imm` instead of a v_add_u32 with a > 16-bit literal per read
typedef const frag_t __attribute__((address_space(3)))* lds_frag_t;
typedef const float __attribute__((address_space(3)))* lds_f32_t;
#define PF_OPAQUE(p) asm volatile("" : "+v"(p))

__device__ __forceinline__ int kmap(int j, int h) { return 8 * (j >> 2) + 4 * h + (j & 3); }
// Residue bytes index 22-row tables.  The host entry points refuse a byte > 21 (PF_EINVAL, data.py:25-26 raises
// KeyError there); the device entry points take buffers the library has never seen, so every table lookup clamps:
// an out-of-alphabet byte reads row 21 ('-') instead of up to 60 KB past a 5.6 KB table, and k_embed raises the
// handle's sticky flag so that the next synchronising call reports PF_EINVAL (include/phyloformer_amd.h).
__device__ __forceinline__ int residue(int r) { return min(r, NA - 1); }

// ---- cross-lane helpers -----------------------------------------------------------------
// v_permlane32_swap(vdst=v, src=v): r[0] = value of lane (l & 31), r[1] = value of lane 32 + (l & 31).
__device__ __forceinline__ float pair_sum(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float pair_other(float v, int h) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return h ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// all-reduce sum over each row of 16 lanes
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x124>(v);  // row_ror:4
    v += dpp_f<0x128>(v);  // row_ror:8
    return v;
}
// all-reduce sum over each half-wave (32 lanes): rows {0,1} and rows {2,3}
__device__ __forceinline__ float half32_sum(float v) {
    v = row16_sum(v);
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// xor-shuffle inside each 32-lane half through the LDS crossbar (no LDS memory, no VALU slot)
template <int MASK>
__device__ __forceinline__ float swz_xor(float v) {
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (MASK << 10) | 0x1f));
}
// Transposing reduction: every lane holds 32 partial values v[j]; afterwards lane t (of each half)
// holds sum over the half's 32 lanes of v[t].  31 shuffles instead of 160 for 32 all-reduces, and
// the running statistics need one register per lane instead of 32.
template <int M, int DPP_CTRL>
__device__ __forceinline__ void treduce_step(float* v, int t) {
    // exchange partner: lane ^ 16 through the LDS crossbar (DPP_CTRL == 0), otherwise a DPP row
    // pairing that flips bit log2(M) of the lane: row_mirror (8), row_half_mirror (4), quad reverse
    // (2), quad xor 1 (1) - VALU-only, no LDS round trip
    const bool up = (t & M) != 0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const float send = up ? v[i] : v[i + M];
        const float keep = up ? v[i + M] : v[i];
        v[i] = keep + (DPP_CTRL ? dpp_f<DPP_CTRL>(send) : swz_xor<16>(send));
    }
}
// The same step without the two selects per value pair (they were half of the reduction's VALU instructions).
// Lanes whose bit log2(M) is clear want v[i] + partner's v[i], the others v[i + M] + partner's v[i + M]:
//   M = 16: v_permlane16_swap exchanges the odd 16-lane rows of v[i] with the even rows of v[i + M]; the sum of
//           the two registers is then exactly that (one swap + one add instead of two selects, a swizzle and an add);
//   M = 8, 4: two DPP adds, each enabled (bank_mask) only for the lanes it is meant for - banks are 4-lane groups,
//           so bit 3 / bit 2 of the lane selects {0, 1} vs {2, 3} / {0, 2} vs {1, 3}; disabled lanes keep v[i].
// Operands and their pairing are those of treduce_step (own value + partner's value: the same fp32 sums).
__device__ __forceinline__ void treduce_step16_swap(float* v) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 16]), false, false);
        v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
}
#define PF_DPP_PAIR(D, S, CTRL, M0, M1) \
    "v_add_f32_dpp " D ", " D ", " D " " CTRL " row_mask:0xf bank_mask:" M0 "\n\tv_add_f32_dpp " D ", " S ", " S " " CTRL " row_mask:0xf bank_mask:" M1 "\n\t"
// One asm block per step.  Hazard (not interlocked on gfx9): a DPP read of a VGPR needs two wait states after the
// VALU write of that VGPR, and hipcc pads nothing in front of or inside an asm statement - hence the s_nop 1 at
// both ends (inside the block no DPP source is written by an earlier instruction of the block).
__device__ __forceinline__ void treduce_step8_masked(float* v) {
    asm("s_nop 1\n\t"
        PF_DPP_PAIR("%0", "%8", "row_mirror", "0x3", "0xc") PF_DPP_PAIR("%1", "%9", "row_mirror", "0x3", "0xc")
        PF_DPP_PAIR("%2", "%10", "row_mirror", "0x3", "0xc") PF_DPP_PAIR("%3", "%11", "row_mirror", "0x3", "0xc")
        PF_DPP_PAIR("%4", "%12", "row_mirror", "0x3", "0xc") PF_DPP_PAIR("%5", "%13", "row_mirror", "0x3", "0xc")
        PF_DPP_PAIR("%6", "%14", "row_mirror", "0x3", "0xc") PF_DPP_PAIR("%7", "%15", "row_mirror", "0x3", "0xc")
        "s_nop 1"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
        : "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
}
__device__ __forceinline__ void treduce_step4_masked(float* v) {
    asm("s_nop 1\n\t"
        PF_DPP_PAIR("%0", "%4", "row_half_mirror", "0x5", "0xa") PF_DPP_PAIR("%1", "%5", "row_half_mirror", "0x5", "0xa")
        PF_DPP_PAIR("%2", "%6", "row_half_mirror", "0x5", "0xa") PF_DPP_PAIR("%3", "%7", "row_half_mirror", "0x5", "0xa")
        "s_nop 1"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])
        : "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
}
__device__ __forceinline__ float treduce32(float (&v)[32], int t) {
    treduce_step16_swap(v);
    treduce_step8_masked(v);
    treduce_step4_masked(v);
    treduce_step<2, 0x1B>(v, t);
    treduce_step<1, 0xB1>(v, t);
    return v[0];
}

// Same for 8 values: afterwards every lane j (mod 8) holds the sum over its whole 32-lane half of v[j].
// Three transposing steps inside each 8-lane group (DPP only), then two plain all-reduce steps across the
// four groups: 9 shuffle-adds instead of 8 x 5.
__device__ __forceinline__ float treduce8(float (&v)[8], int t) {
    treduce_step4_masked(v);
    treduce_step<2, 0x1B>(v, t);
    treduce_step<1, 0xB1>(v, t);
    float r = v[0];
    r += dpp_f<0x128>(r);      // row_ror:8  (lane ^ 8 inside the 16-lane row)
    r += swz_xor<16>(r);       // lane ^ 16
    return r;
}

// ---- numerics ---------------------------------------------------------------------------
__device__ __forceinline__ float elu1_fast(float v) {
    // elu(v) + 1 with the hardware exp2 (about 1 ulp): v > 0 ? v + 1 : exp(v)
    return v > 0.f ? v + 1.f : __builtin_amdgcn_exp2f(v * 1.44269504088896340736f);
}
template <int PAT>
__device__ __forceinline__ float swz(float v) {
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), PAT));
}

__device__ __forceinline__ float elu1_acc(float v) { return v > 0.f ? v + 1.f : expf(v); }
__device__ __forceinline__ float gelu_erf(float v) {
    return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
}
// Branch-free erf-GELU: gelu(x) = max(x,0) - 0.5|x| * poly(t) * exp(-x^2/2), t = 1/(1 + p|x|/sqrt2)
// (Abramowitz & Stegun 7.1.26, |erf error| <= 1.5e-7 -> |gelu error| <= 5.3e-7 over all x, measured).
// 12 VALU + v_rcp_f32 + v_exp_f32 instead of ocml erff's two divergent branches (~45 instructions).
__device__ __forceinline__ float gelu_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
    const float e = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170368f);  // exp(-x^2/2)
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    return fmaf(-0.5f * ax, p * e, fmaxf(x, 0.f));
}
__device__ __forceinline__ float softplus20(float v) { return v > 20.f ? v : log1pf(expf(v)); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// FFN hidden activation for two accumulator values at once, fused with the 16-bit hi/lo split.
// The FFN weights are pre-scaled on the host so that the accumulator holds a*h with
// a^2 = log2(e)/2 (the variable the activation polynomial below is fitted in) and W2 carries 1/a;
// the function returns 2*a*gelu(h) as packed 16-bit pairs (W2 carries 1/(2a)):  hi = h16(g), lo = h16(g - hi).
// Scalar VOP3 forms with free |x| / -x modifiers; built with -fno-slp-vectorize so that hipcc does
// not re-pack them into v_pk_* (see gelu_scaled).
constexpr float GELU_ALPHA = 0.84932180028801904272f;   // sqrt(log2(e) / 2)
__device__ __forceinline__ float gelu_scaled(float x) {
    // a*gelu(h) for x = a*h:  gelu(h) = max(h, 0) - |h| Q(|h|),  Q(u) = erfc(u / sqrt2) / 2 = 2^P(a u),
    // P a degree-5 fit of log2 Q weighted by u Q (|gelu error| <= 8.3e-7, measured over |h| <= 60; the
    // negative leading coefficient makes the tail vanish for any |h|).  5 FMA + v_exp_f32 + max + FMA
    // = ~30 VALU cycles, against 45 for the Abramowitz-Stegun form (rcp + exp + 10 ops) it replaces.
    // Plain (non-packed) fp32 ops only: on gfx950 v_pk_fma_f32 / v_pk_mul_f32 do not overlap with
    // another wave's MFMAs (MFMA || v_pk_fma = sum of both, MFMA || v_fma = max) - tools/valu_bench.hip.
    // Returned value is 2*a*gelu(h) = x + |x| (1 - 2 Q): the factor 2 (folded into W2 on the host, exact)
    // turns max(x, 0) - |x| Q into one subtraction and one FMA; v_max_f32 costs 4.4 cycles per wave
    // instruction on gfx950 against 2.5 for add / mul / fma (tools/overlap3_bench.hip).
    const float u = fabsf(x);
    float p = fmaf(-0.00107098569f, u, 0.0136151873f);
    p = fmaf(p, u, -0.084594565f);
    p = fmaf(p, u, -0.637684925f);
    p = fmaf(p, u, -1.35494915f);
    p = fmaf(p, u, -0.00003762f);                       // log2(2 Q): constant term of log2 Q plus one
    return fmaf(u, 1.f - __builtin_amdgcn_exp2f(p), x);
}
// (g0, g1) -> packed 16-bit pairs hi = h16(g), lo = h16(g - hi).  The residual g - hi comes from v_fma_mix_f32 (fp16)
// or from v_dot2c_f32_bf16 (bf16 builds: g += hi.lo * -1 + hi.hi * 0, one 4.5-cycle op instead of unpack (4.2) +
// subtract (2.9), bit-identical to the fp32 subtraction, tools/dot2c_test.hip).
__device__ __forceinline__ void split_pair(float g0, float g1, unsigned& hi_out, unsigned& lo_out) {
#if PF_F16
    // (opaque copies: hipcc otherwise fuses the conversion of the hi limb into the producer of g - fpround(fma) ->
    // v_fma_mixlo_f16 + v_fma_mixhi_f16, two instructions per pair that repeat the GELU's last FMA - where one
    // v_cvt_pk_f16_f32 per pair does: 8 instructions of 231 in the hidden loop, tools/isa_histogram.py)
    asm("" : "+v"(g0));
    asm("" : "+v"(g1));
#endif
    const h16x2 h2 = {(h16_t)g0, (h16_t)g1};
    const unsigned hb = __builtin_bit_cast(unsigned, h2);
    float r0 = g0, r1 = g1;
    // HAZARD (measured on gfx950, not interlocked): a VALU that reads the result of v_dot2c_f32_f16 needs >= 3 wait
    // states after it (tools/f16_probe.hip: stale data after two; the bf16 form needs two); hipcc pads nothing
    // inside an asm statement, so the pad lives in the string, with one spare state.
#ifndef PF_DOT2C_PRE
#define PF_DOT2C_PRE ""
#define PF_DOT2C_POST "\n\ts_nop 3"
#endif
#if PF_F16 && !defined(PF_SPLIT_DOT2C) && !defined(PF_SPLIT_NODOT)
    // fp16 (the default): v_fma_mix_f32 reads a half straight out of the packed register (op_sel picks it) and returns
    // g - hi = hi * -1 + g in ONE ordinary VALU instruction - bit-identical to unpack + subtract on 2^20 values,
    // subnormal hi included, interlocked like any VALU result (tools/f16_probe.hip) - where the bf16 kernels of rounds
    // 1-5 needed v_dot2c_f32_bf16, a matrix-side op with a software-managed hazard: k_main -3 %, the forward +1.9 %
    // against v_dot2c_f32_f16 (-DPF_SPLIT_DOT2C; profiles/r06m_ab_fmamix.txt).  Separate non-volatile statements, so
    // the scheduler places them like its own instructions.  (Writing the lo limb straight from the FMA -
    // v_fma_mixlo_f16 / v_fma_mixhi_f16, one instruction less per pair, equally exact - was slower: the read-modify-
    // write of the packed destination costs k_main<MID, flat> its last registers, profiles/r06n_ab_mixlo.txt.)
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hb), "v"(g0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hb), "v"(g1));
#elif defined(PF_SPLIT_NODOT)
    // same residual through plain VALU (unpack / subtract): two more instructions per pair, but
    // unlike v_dot2c (matrix-side datapath) they can issue in the shadow of another wave's MFMA
    r0 = fmaf((float)h2[0], -1.f, g0);
    r1 = fmaf((float)h2[1], -1.f, g1);
#else
    asm(PF_DOT2C_PRE PF_DOT2C " %0, %2, %4\n\t" PF_DOT2C " %1, %3, %4" PF_DOT2C_POST
                 : "+v"(r0), "+v"(r1) : "s"(PF_NEG1_LO), "s"(PF_NEG1_HI), "v"(hb));
#endif
    const h16x2 l2 = {(h16_t)r0, (h16_t)r1};
    hi_out = hb;
    lo_out = __builtin_bit_cast(unsigned, l2);
}
__device__ __forceinline__ void gelu_split_pair(float x0, float x1, unsigned& hi_out, unsigned& lo_out) {   // tools/
    split_pair(gelu_scaled(x0), gelu_scaled(x1), hi_out, lo_out);
}
// split 8 floats into hi + lo fragments (x ~= hi + lo to 2^-22 relative in fp16, 2^-17 in bf16).  fp16: four
// split_pair()s.  The v_dot2c variants issue their eight residuals back to back in one asm block: each then sits
// >= 2 instructions ahead of the first reader of its result, and a single s_nop covers the last one - instead of
// one pad per pair as in split_pair (the pads alone were ~3 % of k_main's issue slots).
__device__ __forceinline__ void split8(const float* v, frag_t& hi, frag_t& lo) {
#if defined(PF_SPLIT_NODOT) || (PF_F16 && !defined(PF_SPLIT_DOT2C))
    u32x4 h, l;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned a, b;
        split_pair(v[2 * k], v[2 * k + 1], a, b);
        h[k] = a;
        l[k] = b;
    }
    hi = __builtin_bit_cast(frag_t, h);
    lo = __builtin_bit_cast(frag_t, l);
#else
    u32x4 h, l;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const h16x2 h2 = {(h16_t)v[2 * k], (h16_t)v[2 * k + 1]};
        h[k] = __builtin_bit_cast(unsigned, h2);
    }
    float r0 = v[0], r1 = v[1], r2 = v[2], r3 = v[3], r4 = v[4], r5 = v[5], r6 = v[6], r7 = v[7];
    unsigned l0, l1, l2, l3;
    // not volatile: a pure function of its operands, so the scheduler may interleave independent splits instead of
    // keeping every asm statement in program order.  The conversions of the residuals sit INSIDE the block, in issue
    // order: v_dot2c_f32_f16 needs >= 3 wait states before a VALU reads its result (tools/f16_probe.hip: two - enough
    // for the bf16 form - return stale data), and outside the block hipcc is free to read the last residual first.
    // Here the closest reader (of r7) is 2 + 3 = 5 wait states behind its dot2c.
    asm(
        PF_DOT2C " %4, %12, %14\n\t" PF_DOT2C " %5, %13, %14\n\t"
        PF_DOT2C " %6, %12, %15\n\t" PF_DOT2C " %7, %13, %15\n\t"
        PF_DOT2C " %8, %12, %16\n\t" PF_DOT2C " %9, %13, %16\n\t"
        PF_DOT2C " %10, %12, %17\n\t" PF_DOT2C " %11, %13, %17\n\ts_nop 1\n\t"
        PF_CVT_PK " %0, %4, %5\n\t" PF_CVT_PK " %1, %6, %7\n\t" PF_CVT_PK " %2, %8, %9\n\t" PF_CVT_PK " %3, %10, %11"
        : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3),
          "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
        : "s"(PF_NEG1_LO), "s"(PF_NEG1_HI), "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]));
    l[0] = l0; l[1] = l1; l[2] = l2; l[3] = l3;
    hi = __builtin_bit_cast(frag_t, h);
    lo = __builtin_bit_cast(frag_t, l);
#endif
}
// eight accumulator values acc[base .. base+7] -> GELU -> one B-operand fragment pair
__device__ __forceinline__ void gelu_split8(const f32x16& acc, int base, frag_t& hi, frag_t& lo) {
    float g[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = gelu_scaled(acc[base + k]);
    split8(g, hi, lo);
}


// acc += (a_hi + a_lo) * (b_hi + b_lo) without the lo*lo term.
// Pass order: consecutive MFMAs share one operand - (hi,lo) (hi,hi) (lo,hi): A changes once, B changes once -
// and `flip` reverses it, so that two back-to-back calls with the same B operand (two output tiles of one K step)
// also share it across the seam: ... (lo,hi) | (lo',hi) (hi',hi) (hi',lo).  Fewer operand switches between
// consecutive MFMAs measurably lower the power the matrix pipe draws: -1.0 % launch time against the
// (lo,hi) (hi,lo) (hi,hi) order of round 1 (A/B in one GPU call, tools/flag_compare.py).
__device__ __forceinline__ void mfma3(f32x16& acc, const frag_t& a_hi, const frag_t& a_lo,
                                      const frag_t& b_hi, const frag_t& b_lo, const bool flip = false) {
    if (!flip) {
        acc = PF_MFMA(a_hi, b_lo, acc);
        acc = PF_MFMA(a_hi, b_hi, acc);
        acc = PF_MFMA(a_lo, b_hi, acc);
    } else {
        acc = PF_MFMA(a_lo, b_hi, acc);
        acc = PF_MFMA(a_hi, b_hi, acc);
        acc = PF_MFMA(a_hi, b_lo, acc);
    }
}

// first product of a chain: C is the inline constant 0 of the MFMA encoding, so the accumulator needs no
// sixteen v_mov to be cleared
__device__ __forceinline__ void mfma3_zero(f32x16& acc, const frag_t& a_hi, const frag_t& a_lo,
                                           const frag_t& b_hi, const frag_t& b_lo, const bool flip = false) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!flip) {
        acc = PF_MFMA(a_hi, b_lo, z);
        acc = PF_MFMA(a_hi, b_hi, acc);
        acc = PF_MFMA(a_lo, b_hi, acc);
    } else {
        acc = PF_MFMA(a_lo, b_hi, z);
        acc = PF_MFMA(a_hi, b_hi, acc);
        acc = PF_MFMA(a_hi, b_lo, acc);
    }
}

// LayerNorm without affine over the 64 channels of a token held by lanes (t,0) and (t,1)
__device__ __forceinline__ void ln_pair(const float (&x)[32], float (&xn)[32]) {
    // four independent partial sums: a single 32-long dependent chain is ~250 cycles of latency that the
    // one other wave on the SIMD cannot hide
    float s4[4] = {x[0], x[1], x[2], x[3]};
#pragma unroll
    for (int j = 4; j < 32; ++j) s4[j & 3] += x[j];
    const float mean = pair_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) * (1.f / 64.f);
    float v4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        xn[j] = x[j] - mean;
        v4[j & 3] = fmaf(xn[j], xn[j], v4[j & 3]);
    }
    const float v = (v4[0] + v4[1]) + (v4[2] + v4[3]);
    // v_rsq_f32 (1 ulp) instead of sqrt + IEEE division (~25 VALU instructions with the fix-up sequence)
    const float rstd = __builtin_amdgcn_rsqf(pair_sum(v) * (1.f / 64.f) + LN_EPS);
#pragma unroll
    for (int j = 0; j < 32; ++j) xn[j] *= rstd;
}

typedef const f32x4 __attribute__((address_space(3)))* lds_f32x4_t;
__device__ __forceinline__ void load_acc_bias(f32x16& acc, lds_f32_t lds_bias_h) {
    // acc[r] = bias[row(r, h)], row = 8*(r>>2) + 4*h + (r&3): four 16-byte LDS reads; the caller
    // passes bias + 4*h
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 bb = *(lds_f32x4_t)(lds_bias_h + 8 * q4);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * q4 + i] = bb[i];
    }
}

__device__ __forceinline__ void load_acc_bias(f32x16& acc, const float* lds_bias, int h) {   // tools/
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(lds_bias + 8 * q4 + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * q4 + i] = bb[i];
    }
}

__device__ __forceinline__ frag_t zero_frag() {
    frag_t z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (h16_t)0.f;
    return z;
}

#ifndef PF_DEVICE_HELPERS_ONLY      // (pf_mha.hip takes the helpers above and none of the kernels below)
struct MainArgs {
    float* x;               // [B][P][Lloc][64] in/out
    float* qrow;            // [B][P][Lloc][4]  in: q' of this block's row attn; out: next block's
    const float* qcol;      // [B][P][Lloc][4]
    const frag_t* mfrag;    // [B][P][2 To][2 hi/lo][32] row mix base M_base^T (+bias rows) as MFMA A fragments
    const float* rq;        // [B][P][4] L / S_q[h] (* RowFinArgs::b_scale): the factor that turns q' into q' / mean(q')
    const float* ctx;       // [B][Lloc][64]
    float* spart;           // [B][P][ntiles][72]  out: per-tile statistics of the next block's row attn
    float* outpart;         // [B][P][ntiles]      out (last block): per-tile sum_l softplus
    const frag_t* wimg;     // LDS image (FRAG_END fragments) in global memory
    const float* consts;    // CONST_LEN floats
    const frag_t* wv_lo;    // [2 T][4 s][64] lo fragments of the next row attn's Wv' (global, L1/L2)
    const float* table;     // [22][64] relu(W_emb + b_emb)          (MODE_FIRST)
    const uint8_t* idx;     // [B][N][Lloc]                           (MODE_FIRST)
    const int16_t* pair_i;  // [P]                                    (MODE_FIRST)
    const int16_t* pair_j;  // [P]
    int B, N, P, Lloc;
    int flat;               // tiling (tile_plan() on the host): 0 = every pair row has its own ceil(L / 32) tiles,
                            // 1 = the P * L tokens of an alignment are cut into tiles of 32 consecutive tokens
    int nt_aln;             // tiles per alignment
    int slots_aln;          // per-tile partial slots per alignment (spart / outpart): nt_aln, + P when flat
    size_t trash_tok;       // token index of a 32-token scratch area behind x and qrow (masked lanes)
    int store_x_last;       // debug: MODE_LAST also writes x back
    unsigned long long* prof;   // optional: per-phase cycle totals [8] (s_memtime), perf experiments
    int ablate;             // perf experiments only (results invalid): 1 no x load, 2 no stores,
                            // 4 no next-row phase, 8 no apply phase, 16 no FFN (2 is unused now)
};

enum { MODE_FIRST = 0, MODE_MID = 1, MODE_LAST = 2, MODE_MID0 = 3 };

// Two tilings of an alignment's P x L tokens into work items of 32 tokens share the kernel:
//   row tiling  (flat = 0): every pair row is cut into ceil(L / 32) tiles of its own; the last tile of a row is
//               ragged (L = 500: 20 of 32 lanes valid, 2.3 % of all MFMA columns wasted; L = 200: 10.7 %);
//   flat tiling (flat = 1; the host picks it when L >= 32 and L % 32 != 0): the alignment's P * L tokens - they
//               are contiguous in x, qrow and qcol - are cut into tiles of 32 CONSECUTIVE tokens, so a tile
//               may cover the last sites of row r0 and the first of row r0 + 1 (L >= 32: never more than two
//               rows); only the alignment's last tile is ragged.  What is per row - the row-mix fragments, the
//               statistics partial, the head sum - is then done once per row PART of such a tile.
// Partials: a tile part owns slot (kt + its row) in flat tiling, kt in row tiling; the parts of row p are the
// consecutive slots part_range() names, summed in slot order by k_rowfin / k_rowsum / k_outsum.  The tiling is
// per alignment, a function of (P, L) only: an alignment gets the same bits wherever it sits in a batch.
struct TilePos { int b, kt, r0, l0; };   // alignment, tile inside it, row of its first token, site of that token
template <bool FLAT>
__device__ __forceinline__ TilePos tile_pos(const MainArgs& a, int task, int ntiles) {
    TilePos q;
    q.b = task / a.nt_aln;
    q.kt = task - q.b * a.nt_aln;
    if (FLAT) { const int tok0 = q.kt * 32; q.r0 = tok0 / a.Lloc; q.l0 = tok0 - q.r0 * a.Lloc; }
    else { q.r0 = q.kt / ntiles; q.l0 = (q.kt - q.r0 * ntiles) * 32; }
    return q;
}
template <bool FLAT>
__device__ __forceinline__ TilePos tile_next(const MainArgs& a, TilePos q) {
    if (++q.kt == a.nt_aln) { q.b++; q.kt = 0; q.r0 = 0; q.l0 = 0; return q; }
    q.l0 += 32;
    if (q.l0 >= a.Lloc) { q.l0 = FLAT ? q.l0 - a.Lloc : 0; q.r0++; }
    return q;
}
// lane t of a tile: its row (inside the alignment), its site, whether it holds a token at all and whether that
// token belongs to the tile's second row.  Lanes without a token are clamped onto a real one (finite values,
// their statistics are masked and their stores go to the trash area).
// Everything that can be is wave-uniform: tok0 (the tile's first token, counted over the batch: tokens of a tile
// are consecutive in both tilings), nvalid (lanes holding a token; < 32 only in a row's / an alignment's last
// tile) and wrap (flat tiling: lanes from `wrap` on belong to row r0 + 1; 32 = none).  Per lane that leaves
// one min for the clamped token offset and a compare + select for the site.
struct LanePos { size_t tok0; int toff, l, nvalid, wrap; bool valid, in_r1; };
template <bool FLAT>
__device__ __forceinline__ LanePos lane_pos(const MainArgs& a, const TilePos& q, int t) {
    LanePos o;
    o.tok0 = ((size_t)q.b * a.P + q.r0) * a.Lloc + q.l0;
    const int left = FLAT ? a.P * a.Lloc - q.kt * 32 : a.Lloc - q.l0;      // tokens from the tile's first one on
    o.nvalid = min(32, left);
    o.wrap = (FLAT && q.r0 + 1 < a.P) ? min(32, a.Lloc - q.l0) : 32;
    o.valid = t < o.nvalid;
    o.toff = min(t, o.nvalid - 1);
    o.in_r1 = o.toff >= o.wrap;
    o.l = q.l0 + o.toff - (o.in_r1 ? a.Lloc : 0);
    return o;
}
// Slots [first, first + count) of the partial buffers that hold row p's (pr = b * P + p) parts.
__device__ __forceinline__ void part_range(int flat, int pr, int nparts, int P, int L, int slots_aln, long* first, int* count) {
    if (!flat) { *first = (long)pr * nparts; *count = nparts; return; }
    const int b = pr / P, p = pr - b * P;
    const int k0 = (int)(((long)p * L) >> 5), k1 = (int)((((long)p + 1) * L - 1) >> 5);
    *first = (long)b * slots_aln + k0 + p;
    *count = k1 - k0 + 1;
}

// Work item = one tile of 32 tokens (see above).  The tiles
// are dealt to the waves in short runs of consecutive tiles, round-robin; waves never synchronise with
// each other after the LDS image is loaded.  Row statistics leave the kernel as one
// 72-float partial per TILE (summed in fixed order by k_rowfin / k_rowsum), never as per-wave running sums:
// the result bits do not depend on how the tiles were dealt to waves, i.e. on batch size or grid, and a
// lone small alignment still spreads over the whole chip (190 rows x 7 tiles of a 20 x 200 alignment are
// 1,330 work items instead of 190).
//   MODE_FIRST: x = embedding pair sum;                      -> row stats of block 0
//   MODE_MID  : row-apply + col-apply + FFN of block k;      -> row stats of block k+1
//   MODE_MID0 : the same for block 0, whose input x0 = T[a_i] + T[a_j] (model.py:173-175) is formed on the
//               fly from the 5.6 KB embedding table (L1-resident) instead of being read from HBM: x0 is
//               never materialised (3.6 GB less written and 2 x 3.6 GB less read per batch of 16)
//   MODE_LAST : row-apply + col-apply + FFN of the last block -> softplus head, site mean
//   FLAT      : the tiling (a.flat), a template parameter so that the row tiling carries none of the flat
//               tiling's bookkeeping (shapes with L % 32 == 0, rows shorter than a tile)
template <int MODE, bool FLAT>
__global__ void __launch_bounds__(MAIN_THREADS, MAIN_WAVES / 4) k_main(MainArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_frag_t lw = (lds_frag_t)smem;
    lds_f32_t lc = (lds_f32_t)(smem + FRAG_END * 16);

    {
        // The LDS image (157 KB per workgroup, L2-resident after the first one) is requested in ONE go:
        // global_load_lds_dwordx4 moves 16 bytes per lane straight into LDS (destination = wave-uniform base +
        // lane * 16: the image is linear, so wave w of step k lands fragments 512 k + 64 w ..), no staging
        // registers, all ~20 requests of a wave in flight before the single wait.  Until round 4 this was a
        // load - wait - ds_write loop, one L2 round trip per step: ~10 us per launch - nothing in a 4 ms launch,
        // a third of k_main's 28 us when a lone 20 x 200 alignment gives every wave one tile.
        // MODE_FIRST only needs the row-statistics operands; the last block only the FFN / out_proj
        constexpr int lo = (MODE == MODE_FIRST) ? FRAG_WV : 0;
        constexpr int hi = (MODE == MODE_LAST) ? FRAG_WV : FRAG_END;
        static_assert(lo % 64 == 0 && hi % 64 == 0 && FRAG_END % 64 == 0, "the image is copied in whole waves");
        const frag_t* src = a.wimg;
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        const float cv = a.consts[min((int)threadIdx.x, CONST_LEN - 1)];
#pragma unroll
        for (int k = 0; k < (hi - lo + MAIN_THREADS - 1) / MAIN_THREADS; ++k) {
            const int f0 = lo + k * MAIN_THREADS + wv * 64;            // wave-uniform first fragment of this request
            if (f0 < hi) __builtin_amdgcn_global_load_lds(src + f0 + ln, smem + (size_t)f0 * 16, 16, 0, 0);
        }
        static_assert(CONST_LEN <= MAIN_THREADS, "one constant per thread");
        if (threadIdx.x < CONST_LEN) reinterpret_cast<float*>(smem + FRAG_END * 16)[threadIdx.x] = cv;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = lane & 31;
    const int h = lane >> 5;
    const int ntiles = (a.Lloc + 31) >> 5;
    const int ntasks = a.B * a.nt_aln;           // < 2^31: the host cuts larger batches into chunks
    // static priority for the second-dispatched half of the workgroup (MI355X_MICROARCH.md, two waves per
    // SIMD, item 4: the younger wave otherwise loses every VALU arbitration): -0.3 % launch time, A/B measured
    if (wave >= MAIN_WAVES / 2) __builtin_amdgcn_s_setprio(1);
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
#define PF_TICK(k) do { if (a.prof) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tacc[k] += tn_ - tprev; tprev = tn_; } } while (0)
    if (a.prof) tprev = __builtin_amdgcn_s_memtime();
    lds_frag_t w1p = lw + FRAG_W1 + lane;   // per-lane bases: all later offsets are immediates
    lds_frag_t w2p = lw + FRAG_W2 + lane;
    lds_frag_t wop = lw + FRAG_WO + lane;
    lds_frag_t qkp = lw + FRAG_QK + h * 8 + (t & 7);
    lds_f32_t lch = lc + 4 * h;
    PF_OPAQUE(wop); PF_OPAQUE(qkp); PF_OPAQUE(lch);
    // Wv' hi sits 16 KB behind Wo in the image: the same per-lane base, the distance in the 16-bit offset field of
    // ds_read_b128 (one base register less to keep alive across the tile loop)
    lds_frag_t wvp = wop + (FRAG_WV - FRAG_WO);

    {
        // Tiles are dealt in short RUNS of consecutive tiles, round-robin over the waves: at any moment the
        // 2,048 waves work inside a window of nwaves * run tiles (a few hundred rows of one or two
        // alignments), so ctx and the row-mix fragments stay hot in L2 - one contiguous chunk per wave had
        // every wave in a different alignment and cost 0.7 GiB of extra L2 misses per launch (FETCH_SIZE 2.67
        // -> 1.99 GiB, tools/fetch_ab.sh; the launch itself is 1 % slower this way, run lengths 2..16 alike)
        // - while a wave still streams `run` consecutive 8 KB tiles and the load stays balanced to one run.
        const int nwaves = gridDim.x * MAIN_WAVES;
#ifndef PF_RUN_MAX
#define PF_RUN_MAX 8
#endif
        const int run = max(1, min(PF_RUN_MAX, ntasks / (nwaves * 8)));
        int run0 = (blockIdx.x * MAIN_WAVES + wave) * run;             // first tile of this wave's current run
        const int task0 = run0, task1 = ntasks;
        TilePos cur = tile_pos<FLAT>(a, min(task0, ntasks - 1), ntiles);
        int frag_row = -1;                          // the row (b * P + p) whose row-mix fragments are in mfr
        frag_t mfr[4];
        f32x4 rqv = {0.f, 0.f, 0.f, 0.f};          // L / S_q[h] of that row (wave-uniform)
        // The next tile's residual rows and q' are requested when the FFN of the current tile starts
        // (the residual lives in the GEMM2 accumulators from then on, see below) and land during
        // its ~6 us of matrix work, so a tile never starts by waiting on HBM.
        f32x4 px[8], pctx[8], pqr, pqc;
        int pri = 0, prj = 0;                       // MODE_MID0: residues of the next tile's site in both sequences
        auto prefetch = [&](const TilePos& np) {
            const LanePos nl = lane_pos<FLAT>(a, np, t);
            const int ll = nl.l, pb = np.b;
            const size_t tk = nl.tok0 + nl.toff;
            if (MODE == MODE_MID0) {
                const uint8_t* ib = a.idx + (size_t)pb * a.N * a.Lloc + ll;
                const int pp = np.r0 + (nl.in_r1 ? 1 : 0);
                pri = residue(ib[(size_t)a.pair_i[pp] * a.Lloc]);
                prj = residue(ib[(size_t)a.pair_j[pp] * a.Lloc]);
            } else {
                const f32x4* xp = reinterpret_cast<const f32x4*>(a.x + tk * 64 + 4 * h);
#pragma unroll
                for (int g = 0; g < 8; ++g) px[g] = xp[2 * g];
            }
            pqr = *reinterpret_cast<const f32x4*>(a.qrow + tk * 4);
            pqc = *reinterpret_cast<const f32x4*>(a.qcol + tk * 4);
            if (MODE != MODE_FIRST) {
                // (h through an opaque copy: hipcc otherwise hoists the per-lane 64-bit base a.ctx + 4 h out of the tile
                // loop, where it does not find a register pair for it - round 3's build kept it in scratch and
                // reloaded it once per tile; recomputing costs one 64-bit add)
                int hq = h;
                asm volatile("" : "+v"(hq));
                const f32x4* cp = reinterpret_cast<const f32x4*>(a.ctx + ((size_t)pb * a.Lloc + ll) * 64 + 4 * hq);
#pragma unroll
                for (int g = 0; g < 8; ++g) pctx[g] = cp[2 * g];
            }
        };
        if (MODE != MODE_FIRST && task0 < task1) prefetch(cur);

        for (int task = task0; task < task1;) {
            const int b = cur.b;
            const int row = b * a.P + cur.r0;          // the tile's (first) row, counted over the batch
            // the tile that follows: the next one of this run, or the first one of the wave's next run
            int ntask = task + 1;
            TilePos nxt = tile_next<FLAT>(a, cur);
            if (ntask == run0 + run) {
                run0 += nwaves * run;
                ntask = run0;
                if (ntask < task1) nxt = tile_pos<FLAT>(a, ntask, ntiles);
            }
            const LanePos lp = lane_pos<FLAT>(a, cur, t);
            // flat tiling: does this tile end row r0 and start row r0 + 1 (wave-uniform)?
            const bool straddle = lp.wrap < 32;
            int ai = 0, aj = 0;
            if (MODE == MODE_FIRST) { const int pp = cur.r0 + (lp.in_r1 ? 1 : 0); ai = a.pair_i[pp]; aj = a.pair_j[pp]; }
            auto load_mfr = [&](int r) {
                const frag_t* mf = a.mfrag + (size_t)r * 128 + t;
#pragma unroll
                for (int q = 0; q < 4; ++q) mfr[q] = mf[q * 32];
                // (the row is wave-uniform: read through the constant address space, i.e. one s_load_dwordx4 into SGPRs -
                // as a vector load the four values cost four VGPRs across the whole tile loop, which k_main does not have;
                // rq was written by the previous launch, so the scalar cache cannot hold a stale line)
                typedef const f32x4 __attribute__((address_space(4)))* const_f32x4_t;
                rqv = *(const_f32x4_t)(unsigned long long)(a.rq + (size_t)__builtin_amdgcn_readfirstlane(r) * 4);
                frag_row = r;
            };
            // the pair's row-mix fragments are loaded once per row, not once per tile
            if (MODE != MODE_FIRST && row != frag_row) load_mfr(row);
            const bool valid = lp.valid;
            const int lc_ = lp.l;                      // (clamped) site for gathers
            const size_t tok = lp.tok0 + lp.toff;
            float x[32];

            if (MODE == MODE_FIRST) {
                // embedding lookup + pair expansion (model.py:173-175): x = T[a_i] + T[a_j]
                const int ri = residue(a.idx[((size_t)b * a.N + ai) * a.Lloc + lc_]);
                const int rj = residue(a.idx[((size_t)b * a.N + aj) * a.Lloc + lc_]);
                const f32x4* ti = reinterpret_cast<const f32x4*>(a.table + ri * 64 + 4 * h);
                const f32x4* tj = reinterpret_cast<const f32x4*>(a.table + rj * 64 + 4 * h);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    f32x4 u = ti[2 * g], w = tj[2 * g];
#pragma unroll
                    for (int i = 0; i < 4; ++i) x[4 * g + i] = u[i] + w[i];   // clamped site: finite
                }
            } else {
                // lanes past the end of the row hold a copy of the row's last site (clamped address):
                // finite values, no masking needed - their statistics are multiplied by 0 and their
                // stores go to the trash area
                if (MODE == MODE_MID0) {
                    // embedding lookup + pair expansion: the same fp32 sums k_embed / MODE_FIRST form
                    const f32x4* ti = reinterpret_cast<const f32x4*>(a.table + pri * 64 + 4 * h);
                    const f32x4* tj = reinterpret_cast<const f32x4*>(a.table + prj * 64 + 4 * h);
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const f32x4 u = ti[2 * g], w = tj[2 * g];
#pragma unroll
                        for (int i = 0; i < 4; ++i) x[4 * g + i] = u[i] + w[i];
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 8; ++g)
#pragma unroll
                        for (int i = 0; i < 4; ++i) x[4 * g + i] = px[g][i];
                }
                PF_TICK(0);
                if (!(a.ablate & 8)) {
                    // the accumulators start from the residual itself; both out_proj biases ride in the
                    // row-mix fragments (K slots 4 and 5, k_rowfin)
                    f32x16 ya[2];
#pragma unroll
                    for (int j = 0; j < 32; ++j) ya[j >> 4][j & 15] = x[j];
                    // ---- row attention apply (block k) incl. the out_proj biases: K = {q'[0..3], 1}
                    {
                        const f32x4 qr = pqr;
                        // a tile over two rows applies each row's mix to its own tokens: the other row's
                        // tokens get an all-zero B column (q' and both bias slots)
                        auto mix = [&](const bool mine) {
                            float v[8];
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = mine ? qr[i] * rqv[i] : 0.f;     // q' / mean(q') <= L
                            v[4] = v[5] = mine ? 1.f : 0.f;   // K slot 4: row out_proj bias, slot 5: column's
                            v[6] = v[7] = 0.f;
                            frag_t qb_hi, qb_lo;
                            split8(v, qb_hi, qb_lo);
                            // lanes h = 1 carry K = 8..15, which the B operand zeroes: any finite A will do, so
                            // they hold their partner's fragment instead of a masked load
#pragma unroll
                            for (int To = 0; To < 2; ++To) mfma3(ya[To], mfr[To * 2], mfr[To * 2 + 1], qb_hi, qb_lo, To == 1);
                        };
                        mix(h == 0 && !lp.in_r1);
                        if (straddle) {                 // the next tile starts in row r0 + 1: its fragments stay
                            load_mfr(row + 1);
                            mix(h == 0 && lp.in_r1);
                        }
                    }
                    // ---- column attention apply: o[hd] = q'_c[h] * ctx[site][hd];  y += Wo_c o
                    {
                        const f32x4 qc = pqc * COLAPPLY_B_SCALE;     // (Wo carries the inverse, see "fp16 operand ranges")
                        float o[32];
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            const f32x4 u = pctx[g];
#pragma unroll
                            for (int i = 0; i < 4; ++i) o[4 * g + i] = u[i] * qc[g >> 1];
                        }
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            frag_t ob_hi, ob_lo;
                            split8(&o[8 * s], ob_hi, ob_lo);
#pragma unroll
                            for (int To = 0; To < 2; ++To) {
                                lds_frag_t f = wop + ((To * 4 + s) * 2) * 64;
                                const frag_t a_hi = f[0], a_lo = f[64];
                                mfma3(ya[To], a_hi, a_lo, ob_hi, ob_lo, To == 1);
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 32; ++j) x[j] = ya[j >> 4][j & 15];
                }

                PF_TICK(1);
                // ---- feed-forward (model.py:101-104): x += W2 gelu(W1' x~ + b1') + b2
                if (!(a.ablate & 16)) {
                    frag_t xb_hi[4], xb_lo[4];
                    {
                        float xn[32];
                        ln_pair(x, xn);
#pragma unroll
                        for (int s = 0; s < 4; ++s) split8(&xn[8 * s], xb_hi[s], xb_lo[s]);
                    }
                    // GEMM2 accumulates straight onto the residual: oa = x + b2 + W2 g.  x's registers
                    // are dead for the whole hidden loop and hold the next tile's prefetch instead.
                    f32x16 oa[2];
                    load_acc_bias(oa[0], lch + CONST_B2);
                    load_acc_bias(oa[1], lch + CONST_B2 + 32);
#pragma unroll
                    for (int j = 0; j < 32; ++j) oa[j >> 4][j & 15] += x[j];
                    if (ntask < task1) prefetch(nxt);
                    PF_TICK(2);
#pragma unroll 1
                    for (int T = 0; T < 8; ++T) {
                        f32x16 ha;
                        lds_frag_t f1 = w1p + T * 512;
                        lds_frag_t f2 = w2p + T * 256;
                        lds_f32_t bp = lch + CONST_B1 + 32 * T;
                        PF_OPAQUE(f1); PF_OPAQUE(f2); PF_OPAQUE(bp);
                        load_acc_bias(ha, bp);
                        // software-pipelined fragment reads: the next step's A operands are in
                        // flight while the current step's three MFMAs issue
                        frag_t fh = f1[0], fl = f1[64];
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            frag_t nh, nl;
                            if (s < 3) { nh = f1[(s + 1) * 128]; nl = f1[(s + 1) * 128 + 64]; }
                            else { nh = f2[0]; nl = f2[64]; }
                            mfma3(ha, fh, fl, xb_hi[s], xb_lo[s]);
                            fh = nh; fl = nl;
                        }
                        frag_t g_hi[2], g_lo[2];
                        gelu_split8(ha, 0, g_hi[0], g_lo[0]);
                        gelu_split8(ha, 8, g_hi[1], g_lo[1]);
                        // GEMM2 steps in (u, To) order; fragment (To, 2T+u) lives at f2[(To*32 + u*2)*64]
#pragma unroll
                        for (int st = 0; st < 4; ++st) {
                            const int u = st >> 1, To = st & 1;
                            frag_t nh = fh, nl = fl;
                            if (st < 3) {
                                const int nu = (st + 1) >> 1, nTo = (st + 1) & 1;
                                nh = f2[(nTo * 32 + nu * 2) * 64];
                                nl = f2[(nTo * 32 + nu * 2) * 64 + 64];
                            }
                            mfma3(oa[To], fh, fl, g_hi[u], g_lo[u], To == 1);
                            fh = nh; fl = nl;
                        }
                    }
                    PF_TICK(3);
#pragma unroll
                    for (int j = 0; j < 32; ++j) x[j] = oa[j >> 4][j & 15];
                }
            }

            if (MODE != MODE_LAST && !(a.ablate & 4)) {
                // ---- statistics of the next block's row attention (attention.py:163-190)
                // Wv' lo fragments (L2) are requested before the residual store for the same reason
                // (the first WVLO_LDS of the eight live in LDS - all the image has room for; each fragment taken out of
                // this stream is 1 KB of L2 reads per tile and wave less)
                frag_t wl[8];
#pragma unroll
                for (int i = WVLO_LDS; i < 8; ++i) wl[i] = a.wv_lo[i * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
                {
                    // Branch-free store: a divergent `if (valid)` makes hipcc merge the vmcnt state of
                    // both paths and wait for these stores (HBM write acks) at the first Wv' lo use.
                    // Lanes past the end of the row write to a 32-token trash area behind x / qrow.
                    const size_t stok = valid ? tok : a.trash_tok + t;
                    f32x4* xo = reinterpret_cast<f32x4*>(a.x + stok * 64 + 4 * h);
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        f32x4 u = {x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]};
                        xo[2 * g] = u;
                    }
                }
                if (a.ablate & 32) { cur = nxt; task = ntask; continue; }   // perf experiment: copy-only
                f32x16 va[3];
                {
                    float xn[32];
                    ln_pair(x, xn);
                    frag_t xb_hi[4], xb_lo[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) split8(&xn[8 * s], xb_hi[s], xb_lo[s]);
                    // q/k rows first (operands in LDS): the Wv' lo fragments requested above get the
                    // LayerNorm, the split and these 12 MFMAs to arrive from L2
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        // output rows >= 8 are never read: lanes t >= 8 reuse rows t & 7 (finite) unmasked
                        const frag_t q_hi = qkp[(s * 2) * 16], q_lo = qkp[(s * 2 + 1) * 16];
                        if (s == 0) mfma3_zero(va[2], q_hi, q_lo, xb_hi[s], xb_lo[s]);
                        else mfma3(va[2], q_hi, q_lo, xb_hi[s], xb_lo[s]);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int T = 0; T < 2; ++T) {
                            const frag_t f_hi = wvp[(T * 4 + s) * 64];
                            const frag_t f_lo = T * 4 + s < WVLO_LDS ? wvp[(FRAG_WVLO - FRAG_WV) + (T * 4 + s) * 64] : wl[T * 4 + s];
                            if (s == 0) mfma3_zero(va[T], f_hi, f_lo, xb_hi[s], xb_lo[s], T == 1);
                            else mfma3(va[T], f_hi, f_lo, xb_hi[s], xb_lo[s], T == 1);
                        }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) va[2][r] += lc[CONST_BQK + 4 * h + r];   // rows 0-3 q, 4-7 k
                float qk[4], ot[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) qk[i] = elu1_fast(va[2][i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) ot[i] = pair_other(qk[i], h);
                float qn[4], kn[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    qn[i] = h ? ot[i] : qk[i];
                    kn[i] = h ? qk[i] : ot[i];
                }
                {
                    // both half-waves hold the same q'; all lanes store (no branch, see above)
                    const size_t stok = valid ? tok : a.trash_tok + t;
                    f32x4 qs = {qn[0], qn[1], qn[2], qn[3]};
                    *reinterpret_cast<f32x4*>(a.qrow + stok * 4) = qs;
                }
                // one partial per row part of the tile: slot kt (+ row in flat tiling)
                const long slot0 = (long)b * a.slots_aln + cur.kt + (FLAT ? cur.r0 : 0);
                for (int part = 0; part < (straddle ? 2 : 1); ++part) {
                    float* sp = a.spart + (size_t)(slot0 + part) * SROW;
                    const float vm = (valid && lp.in_r1 == (part == 1)) ? 1.f : 0.f;
                    float km[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) km[i] = kn[i] * vm;
                    {
                        // S_q | S_k of the part: lane j (mod 8) ends up with the half-wave sum of value j
                        float qk8[8];
#pragma unroll
                        for (int i = 0; i < 4; ++i) { qk8[i] = vm * qn[i]; qk8[4 + i] = km[i]; }
                        const float sqk = treduce8(qk8, t);
                        if (lane < 8) sp[64 + lane] = sqk;
                    }
                    {
                        float kv[32];
#pragma unroll
                        for (int j = 0; j < 32; ++j) kv[j] = km[j >> 3] * va[j >> 4][j & 15];
                        sp[kmap(t, h)] = treduce32(kv, t);      // lane (t, h) owns S_kv[kmap(t, h)]
                    }
                }
            } else if (MODE == MODE_LAST) {
                // ---- head: softplus(w.x + b) summed over sites (model.py:182-185)
                float z = 0.f;
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const f32x4 w4 = *(lds_f32x4_t)(lch + CONST_HW + 8 * g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) z = fmaf(w4[i], x[4 * g + i], z);
                }
                z = pair_sum(z) + lc[CONST_HB];
                const float spz = softplus20(z);
                const long slot0 = (long)b * a.slots_aln + cur.kt + (FLAT ? cur.r0 : 0);
                for (int part = 0; part < (straddle ? 2 : 1); ++part) {
                    const float so = half32_sum((valid && lp.in_r1 == (part == 1)) ? spz : 0.f);
                    if (lane == 0) a.outpart[slot0 + part] = so;
                }
                if (a.store_x_last && valid) {
                    f32x4* xo = reinterpret_cast<f32x4*>(a.x + tok * 64 + 4 * h);
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        f32x4 u = {x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]};
                        xo[2 * g] = u;
                    }
                }
            }
            PF_TICK(4);
            cur = nxt;
            task = ntask;
        }
    }
    // (the lane number from the hardware, not from a register that would have to live across the tile loop)
    if (a.prof && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) atomicAdd(a.prof + k, tacc[k]);
    }
#undef PF_TICK
}

// ---- embedding + pair expansion + row statistics of block 0 by table lookup ----------------------
// x0[p = (i, j)][l] = T[a_i[l]] + T[a_j[l]] (model.py:173-175) takes only 22 x 22 values per site, so
// everything block 0's row attention needs from a token - q' (4), k' (4) and k' (x) v (64), i.e. its
// 72-float contribution to the row statistics - is a function of the residue pair and is tabulated on
// the host in double precision (pf_lib.hip::build_pair_table).  The kernel sums table rows over the
// sites of a pair (LDS-resident table, 139 KB), writes q' per token and materialises x0 for the two
// consumers of block 0.  No LayerNorm, no MFMA: it replaces k_main<MODE_FIRST> (1.41 -> ~0.6 ms per
// batch of 16 alignments, now bound by the 3.6 GB write of x0).
constexpr int EMBED_THREADS = 512;
constexpr int EMBED_LDS_BYTES = (PAIRTAB_ROWS * PAIRTAB_W + 22 * 64) * 4;
struct EmbedArgs {
    const uint8_t* idx;      // [B][N][Lloc]
    const int16_t* pair_i;   // [P]
    const int16_t* pair_j;
    const float* ptab;       // [484][72]  S_kv contribution (64) | q' (4) | k' (4)
    const float* table;      // [22][64]
    float* x;                // [B*P][Lloc][64]; NULL = do not materialise x0 (block 0's consumers form it themselves)
    float* qrow;             // [B*P][Lloc][4]
    float* srow;             // [B*P][72]
    int B, N, P, Lloc;
    unsigned* bad_idx;       // host-mapped sticky flag: set when a residue byte > 21 is seen (then clamped to 21)
};

__global__ void __launch_bounds__(EMBED_THREADS) k_embed(EmbedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float esm[];
    float* tab = esm;
    float* emb = esm + PAIRTAB_ROWS * PAIRTAB_W;
    {
        // the 139 KB pair table in one go (see k_main): 17 direct-to-LDS requests per wave in flight at once, the
        // last 8 vectors and the 5.6 KB embedding table through registers
        constexpr int NV = PAIRTAB_ROWS * PAIRTAB_W / 4, FULL = NV / EMBED_THREADS;     // 8,712 vectors: 17 whole steps
        const f32x4* src = reinterpret_cast<const f32x4*>(a.ptab);
        f32x4* dst = reinterpret_cast<f32x4*>(tab);
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < FULL; ++k) {
            const int v0 = k * EMBED_THREADS + wv * 64;
            __builtin_amdgcn_global_load_lds(src + v0 + ln, tab + (size_t)v0 * 4, 16, 0, 0);
        }
        const int it = FULL * EMBED_THREADS + threadIdx.x;
        const f32x4 tail = src[min(it, NV - 1)];
        const f32x4* s2 = reinterpret_cast<const f32x4*>(a.table);
        f32x4* d2 = reinterpret_cast<f32x4*>(emb);
        static_assert(22 * 64 / 4 <= EMBED_THREADS, "one embedding vector per thread");
        const f32x4 ev = s2[min((int)threadIdx.x, 22 * 64 / 4 - 1)];
        if (it < NV) dst[it] = tail;
        if (threadIdx.x < 22 * 64 / 4) d2[threadIdx.x] = ev;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sg = lane >> 4, cl = lane & 15;     // 4 sites per wave instruction, 4 channels per lane
    const int ntasks = a.B * a.P;
    for (int task = blockIdx.x * (EMBED_THREADS / 64) + wave; task < ntasks; task += gridDim.x * (EMBED_THREADS / 64)) {
        const int b = task / a.P, p = task - b * a.P;
        const uint8_t* ri = a.idx + ((size_t)b * a.N + a.pair_i[p]) * a.Lloc;
        const uint8_t* rj = a.idx + ((size_t)b * a.N + a.pair_j[p]) * a.Lloc;
        const size_t row0 = (size_t)task * a.Lloc;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acce = {0.f, 0.f, 0.f, 0.f};
        // Residues: the wave fetches 64 consecutive sites of both sequences with one byte pair per lane, one
        // block ahead, and an iteration takes its residues from a lane of that register (ds_bpermute) - a
        // dependent byte load per iteration (global -> LDS address -> table row) made the walk latency-bound
        // at ~480 cycles per four sites.  Same sites per lane in the same order: same sums.
        // (what travels is the row number of the residue pair in the table: one multiply-add per site instead
        // of one per lane and iteration)
        bool bad = false;
        auto fetch = [&](int blk) {
            const int l = min(blk * 64 + lane, a.Lloc - 1);
            const int ra = ri[l], rb = rj[l];
            bad |= max(ra, rb) >= NA;
            return residue(ra) * 22 + residue(rb);
        };
        const int nblk = (a.Lloc + 63) >> 6;
        int cur = fetch(0);
        for (int blk = 0; blk < nblk; ++blk) {
            const int nxt = fetch(min(blk + 1, nblk - 1));
            const int lbase = blk * 64;
            const int nit = min(16, (a.Lloc - lbase + 3) >> 2);
            auto site = [&](int it, bool valid) {
                const int l = lbase + 4 * it + sg;
                const int rr = __shfl(cur, 4 * it + sg);
                const float* tr = tab + rr * PAIRTAB_W;
                const f32x4 s = *reinterpret_cast<const f32x4*>(tr + 4 * cl);
                const f32x4 e = *reinterpret_cast<const f32x4*>(tr + 64 + 4 * (cl & 1));   // even lanes q', odd k'
                if (valid) {
                    acc += s;
                    acce += e;
                    const size_t tok = row0 + l;
                    if (a.x) {       // wave-uniform: only the round-1 path / the debug tap materialise x0
                        const int ra = rr / 22, rb = rr - 22 * ra;
                        const f32x4 xa = *reinterpret_cast<const f32x4*>(emb + ra * 64 + 4 * cl);
                        const f32x4 xb = *reinterpret_cast<const f32x4*>(emb + rb * 64 + 4 * cl);
                        *reinterpret_cast<f32x4*>(a.x + tok * 64 + 4 * cl) = xa + xb;
                    }
                    if (cl == 0) *reinterpret_cast<f32x4*>(a.qrow + tok * 4) = e;
                }
            };
            if (lbase + 64 <= a.Lloc) {             // a full block: no per-lane validity
#pragma unroll 4
                for (int it = 0; it < 16; ++it) site(it, true);
            } else {
                for (int it = 0; it < nit; ++it) site(it, lbase + 4 * it + sg < a.Lloc);   // (a lane past the end reads the clamped last site, unused)
            }
            cur = nxt;
        }
        // sum the four site sub-groups (lanes 16 and 32 apart)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] += __shfl_xor(acc[i], 16);
            acc[i] += __shfl_xor(acc[i], 32);
            acce[i] += __shfl_xor(acce[i], 16);
            acce[i] += __shfl_xor(acce[i], 32);
        }
        if (sg == 0) {
            float* sr = a.srow + (size_t)task * SROW;
            *reinterpret_cast<f32x4*>(sr + 4 * cl) = acc;
            if (cl < 2) *reinterpret_cast<f32x4*>(sr + 64 + 4 * cl) = acce;
        }
        if (bad && a.bad_idx) *a.bad_idx = 1u;      // (never taken on valid input; plain store, any writer wins)
    }
}

// ---- row finalisation: srow -> mrow ------------------------------------------------------
struct RowFinArgs {
    const float* srow;   // [B*P][nparts][72]  statistics, as nparts partial sums per pair (k_main: one per
                         //                    tile; k_embed / all-reduced: nparts = 1)
    float* mrow;         // [B*P][4][64]   fp32 M[h][c] = M_base[h][c] * L / S_q[h] (k_colstats, debug tap)
    frag_t* mfrag;       // [B*P][2 To][2 hi/lo][32]  MFMA A fragments of M_base^T * a_scale: K slots 0-3 heads (WITHOUT
                         //                    the factor L / S_q[h], which k_main applies to q' - rq below - so that no
                         //                    1 / mean(q') ever has to fit the 16-bit format), 4 the row out_proj bias,
                         //                    5 the column out_proj bias (k_main sets both B slots to 1)
    const float* woT;    // [64 hd][64 c]  row out_proj, transposed
    const float* bv;     // [64] folded row v bias
    const float* bias;   // [64] row out_proj bias
    const float* bias_col;  // [64] column out_proj bias of the same block
    int npairs, nparts;
    float L_total;
    int flat, P, Lloc, slots_aln;   // where a pair's partials are (part_range): nparts each, or k_main's flat tiling
    int iters;           // groups of four pairs per block
    float* rq;           // [B*P][4]  L / S_q[h] * b_scale
    float a_scale, b_scale;   // powers of two, a_scale * b_scale = 1: b_scale < 1 only when L_total > 16,384 (q' / mean(q')
                              // can reach L_total; M_base is bounded by the weights, pf_lib.hip::check_f16_ranges)
};

// A block handles `iters` groups of four pairs one after the other: the 64 out_proj weights a thread holds are
// loaded once per block, not once per four pairs - with many pairs and one partial each (a site-sharded rank:
// 8 x the pairs, all-reduced statistics) that load was most of the kernel (1.7 ms per step at world = 8).
__global__ void __launch_bounds__(256) k_rowfin(RowFinArgs a) {
    __shared__ float ctx[4][64];      // 4 pairs at a time
    __shared__ float mm[4][6][64];
    __shared__ float st[4][SROW];
    __shared__ float rqs[4][4];
    const int sub = threadIdx.x >> 6, c = threadIdx.x & 63;
    // the out_proj column of this thread first: its latency hides behind the partial sums (a lone
    // alignment's forward is a chain of 26 such latencies)
    float wo[4][16];
#pragma unroll
    for (int hh = 0; hh < 4; ++hh)
#pragma unroll
        for (int d = 0; d < 16; ++d) wo[hh][d] = a.woT[(16 * hh + d) * 64 + c];
    const float bvc = a.bv[c], biasc = a.bias[c], biascol = a.bias_col[c];
    for (int it = 0; it < a.iters; ++it) {
    const int pr = (blockIdx.x * a.iters + it) * 4 + sub;
    const bool ok = pr < a.npairs;
    if (ok) {
        // partial statistics are summed in index order: the association is a function of the shape only
        long first;
        int nparts;
        part_range(a.flat, pr, a.nparts, a.P, a.Lloc, a.slots_aln, &first, &nparts);
        const float* sp = a.srow + (size_t)first * SROW;
        // lanes 0..63 take S_kv[c], lanes 0..7 also S_q | S_k; four partials in flight per lane
        const int c2 = 64 + (c & 7);
        // eight partials in flight per lane and round, the tail masked (uniform conditions; x + 0 leaves every bit
        // of x): a row of <= 8 tiles - every shape up to 256 sites - costs ONE round trip, where a 4-wide loop with
        // a scalar remainder took four for the 7 tiles of a 200-site row
        float acc = 0.f, acc2 = 0.f;
        for (int i = 0; i < nparts; i += 8) {
            const float* q0 = sp + i * SROW;
            float v[8], w8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool on = i + q < nparts;
                v[q] = on ? q0[q * SROW + c] : 0.f;
                w8[q] = on ? q0[q * SROW + c2] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { acc += v[q]; acc2 += w8[q]; }
        }
        st[sub][c] = acc;
        if (c < 8) st[sub][64 + c] = acc2;
    }
    __syncthreads();
    if (ok) {
        const float* s = st[sub];
        const int hh = c >> 4;
        const float sk = s[68 + hh];
        // (k / sum k)^T v: attention.py:186-190 (a k'-weighted mean of v: bounded by the weights)
        ctx[sub][c] = (s[c] + bvc * sk) / sk;
        // q / mean(q), attention.py:183: the factor k_main multiplies q' with
        if (c < 4) {
            const float r = a.L_total / s[64 + c];
            rqs[sub][c] = r;
            a.rq[(size_t)pr * 4 + c] = r * a.b_scale;
        }
    }
    __syncthreads();
    if (ok) {
        float* m = a.mrow ? a.mrow + (size_t)pr * MROW : nullptr;
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < 16; ++d) acc = fmaf(wo[hh][d], ctx[sub][16 * hh + d], acc);
            if (m) m[hh * 64 + c] = acc * rqs[sub][hh];
            mm[sub][hh][c] = acc * a.a_scale;
        }
        mm[sub][4][c] = biasc;
        mm[sub][5][c] = biascol;
    }
    __syncthreads();
    if (ok) {
        // A fragments of M^T for the row-apply MFMA: lane t of half 0 holds K slots 0..5
        const int To = c >> 5, t = c & 31;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (i < 6) ? mm[sub][i][32 * To + t] : 0.f;
        frag_t hi, lo;
        split8(v, hi, lo);
        frag_t* mf = a.mfrag + (size_t)pr * MFRAG_PER_PAIR + To * 64 + t;
        mf[0] = hi;
        mf[32] = lo;
    }
    // (the next group's st / ctx / rqs / mm writes are ordered behind this group's reads by the barriers above:
    // st is rewritten before the first barrier, last read before the second; ctx, rqs and mm likewise one phase on)
    }
}

// per-tile partial statistics -> one [72] row per pair (site-sharded runs all-reduce this; debug taps)
__global__ void k_rowsum(const float* spart, float* srow, int npairs, int nparts_, int flat, int P, int Lloc, int slots_aln) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npairs * SROW) return;
    const int pr = i / SROW, c = i - pr * SROW;
    long first;
    int nparts;
    part_range(flat, pr, nparts_, P, Lloc, slots_aln, &first, &nparts);
    const float* sp = spart + (size_t)first * SROW + c;
    float acc = 0.f;
    for (int k = 0; k < nparts; k += 8) {          // eight loads in flight, summed in index order (see k_rowfin)
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = k + q < nparts ? sp[(size_t)(k + q) * SROW] : 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q];
    }
    srow[i] = acc;
}

// per-tile softplus sums of the last block -> distances (site mean, model.py:185)
__global__ void k_outsum(const float* outpart, float* out, int npairs, int nparts_, float inv_L_total, int flat, int P,
                         int Lloc, int slots_aln) {
    const int pr = blockIdx.x * blockDim.x + threadIdx.x;
    if (pr >= npairs) return;
    long first;
    int nparts;
    part_range(flat, pr, nparts_, P, Lloc, slots_aln, &first, &nparts);
    const float* sp = outpart + first;
    float acc = 0.f;
    for (int k = 0; k < nparts; k += 8) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = k + q < nparts ? sp[k + q] : 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q];
    }
    out[pr] = acc * inv_L_total;
}

// ---- column statistics -------------------------------------------------------------------
struct ColStatsArgs {
    const float* x;      // [B][P][Lloc][64]   (unused by the block-0 variant, which forms x0 from `table`)
    const float* qrow;   // [B][P][Lloc][4]
    const float* mrow;   // [B][P][4][64]
    float* qcol;         // [B][P][Lloc][4]  out
    float* part;         // [B][G][Lloc][4*64 + 8]  out: Z~[h][c] | S_q[4] | S_k[4]
    const float* wqk;    // [8][64] folded col q (rows 0-3) and k (rows 4-7)
    const float* bqk;    // [8]
    const float* brow;   // [64] row out_proj bias of this block (row 4 of every pair's mrow: read once, not per pair)
    int B, P, Lloc, G, nchunks;
    // Fixed two-level association of the sum over pairs: every group is cut into runs of `sub` pairs (8, or a multiple
    // of 16; S = runs per group); a run is summed pair by pair from zero, a group is the in-order sum of its
    // runs, the total (k_colfin) the in-order sum of the groups.  fine = 0: one block walks a whole group and
    // folds run after run in registers (part[b][g][l]); fine = 1: one block per run (part[b][g][s][l]) and
    // k_colfin folds - the same tree, so the same bits, whichever way the host picks for the batch at hand.
    int sub, S, fine;
    // block 0 only (EMBED): x0[p = (i, j)][l] = table[idx[i][l]] + table[idx[j][l]]
    const float* table;      // [22][64]
    const uint8_t* idx;      // [B][N][Lloc]
    const int16_t* pair_i;   // [P]
    const int16_t* pair_j;
    int N;
};
constexpr int CPART = 4 * 64 + 8;

// Block = (alignment b, chunk of 32 sites, pair group g); wave w owns sites 8w .. 8w+7 of the chunk and
// all four waves walk the group's pairs together, so the block reads 8 KB of contiguous x per pair and
// shares the per-pair row-attention matrix.  Lanes: site = lane >> 3, channels 8*(lane & 7) .. +7 (two
// 16-byte loads per lane).  Applies the row attention on the fly (it is not materialised in HBM), then
// LayerNorm -> q', k' -> Z~ += k' x~.  Eight lanes per token: three steps per cross-lane reduction and,
// after the transposing butterfly, exactly one projection per lane - 24 VALU instructions per token.
// RING > 0 (the x-reading variant only): the token rows and q' of the next RING pairs travel global -> LDS without
// passing through registers (global_load_lds into a ring of RING slots per wave).  The register prefetch it replaces
// reaches one iteration (two pairs) ahead - all that 244 VGPRs leave room for - and every iteration then started by
// waiting for the loads issued one iteration earlier: with ~0.6 us of work per iteration against > 1 us of loaded HBM
// latency the kernel was latency-bound (4.4 TB/s; cutting its VALU work by 37 % with packed math bought 7 %).  The ring
// holds RING pairs in flight per wave whatever the register budget; same operations on the same values: same bits.
constexpr int CS_SLOT = 2 * 256 + 64;       // floats per ring slot: 8 sites x 64 channels as two 16-byte halves per lane, q' [8][4] (+ 32 spare)
#ifndef PF_CS_RING
#define PF_CS_RING 4
#endif
#ifndef PF_CS_MT
#define PF_CS_MT 16          // pair matrices per staging buffer of the ring variant
#endif
template <bool EMBED, int RING, int MT>
#ifndef PF_CS_WAVES
#define PF_CS_WAVES 2
#endif
__global__ void __launch_bounds__(256, PF_CS_WAVES) k_colstats(ColStatsArgs a) {
    static_assert(!(EMBED && RING), "block 0 reads residue bytes, not x");
    static_assert(RING % 2 == 0, "pairs are consumed two at a time");
    static_assert(MT == 16 || MT == 8, "pair matrices are staged 16 or 8 at a time (runs are multiples of 8)");
    // ONE LDS object (hipcc, ROCm 7.2: with a second __shared__ object it orders every ds_read behind every outstanding
    // global_load_lds - s_waitcnt vmcnt(0) - and the ring below would drain once per iteration; cdna_hip_programming.md):
    //   two staging buffers of MT pair matrices (4 x 64 floats each) | the ring | block 0: embedding table, pair indices
    constexpr int LDS_MST = 2 * MT * MROW, LDS_RING = RING ? 4 * RING * CS_SLOT : 0, LDS_EMB = EMBED ? 22 * 64 : 0;
    // EMBED: the group's (i, j) sequence indices, so that the residue fetch of the next pair is one dependent
    // load (idx byte) instead of two (pair table, then idx byte) - the second level did not fit one iteration
    constexpr int PIJ_CAP = 640;
    __shared__ __attribute__((aligned(16))) float lds_all[LDS_MST + LDS_RING + LDS_EMB + (EMBED ? PIJ_CAP : 0)];
    float* const mst = lds_all;
    float* const xring = lds_all + LDS_MST;
    float* const emb = lds_all + LDS_MST + LDS_RING;
    int16_t* const pij = reinterpret_cast<int16_t*>(lds_all + LDS_MST + LDS_RING + LDS_EMB);
    if (EMBED) {
        // both requests in flight before the first store (a load - wait - store loop costs one L2 round trip per step)
        constexpr int NV = 22 * 64 / 4;
        const f32x4* tsrc = reinterpret_cast<const f32x4*>(a.table);
        const int i0 = threadIdx.x, i1 = threadIdx.x + 256;
        static_assert(NV <= 2 * 256, "two vectors per thread cover the table");
        const f32x4 v0 = tsrc[min(i0, NV - 1)], v1 = tsrc[min(i1, NV - 1)];
        if (i0 < NV) reinterpret_cast<f32x4*>(emb)[i0] = v0;
        if (i1 < NV) reinterpret_cast<f32x4*>(emb)[i1] = v1;
        // visible to all waves after the __syncthreads() that follows the first staging request
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ts = lane >> 3, cl = lane & 7;
    const int per = (a.P + a.G - 1) / a.G;

    // Packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes' worth of fp32 per issue slot).  This kernel has
    // no MFMA beside which packed math would hurt (k_main's reason for -fno-slp-vectorize) and its VALU pipe was 80 %
    // busy (PMC): every per-channel operation below works on pairs.
    // Projection weights as pairs over the OUTPUTS (o, o + 1): pv(o, o + 1) += w2[o / 2][i] * d[i] keeps the
    // per-output summation order over i, i.e. the bits of the scalar loop it replaces.
    f32x2 w2[4][8];
#pragma unroll
    for (int o = 0; o < 8; o += 2)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(a.wqk + o * 64 + 8 * cl + 4 * q);
            const f32x4 u1 = *reinterpret_cast<const f32x4*>(a.wqk + (o + 1) * 64 + 8 * cl + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) w2[o >> 1][4 * q + i] = f32x2{u0[i], u1[i]};
        }
    // after the transposing reduction lane cl holds projection cl (0-3: q', 4-7: k')
    const float bj = a.bqk[cl];
    // the row out_proj bias is the same for every pair: eight registers instead of two of the ten 16-byte LDS
    // reads per lane and pair (the LDS pipe is what this kernel is closest to)
    const f32x4 br0 = *reinterpret_cast<const f32x4*>(a.brow + 8 * cl), br1 = *reinterpret_cast<const f32x4*>(a.brow + 8 * cl + 4);
    const bool up1 = (cl & 2) != 0, up0 = (cl & 1) != 0;
    int bid = blockIdx.x;
    int run = 0;
    if (a.fine) { run = bid % a.S; bid /= a.S; }
    const int g = bid % a.G; bid /= a.G;
    const int chunk = bid % a.nchunks;
    const int b = bid / a.nchunks;
    const int l = chunk * 32 + wave * 8 + ts;
    const bool lvalid = l < a.Lloc;
    const int lcl = lvalid ? l : a.Lloc - 1;
    const int gp0 = g * per, gp1 = min(a.P, gp0 + per);
    const int p0 = a.fine ? gp0 + run * a.sub : gp0, p1 = a.fine ? min(gp1, p0 + a.sub) : gp1;
    if (p0 >= p1) return;       // a run past the end of the last group (k_colfin does not read its slot)
    const float vmask = lvalid ? 1.f : 0.f;
    f32x2 z[4][4], zt[4][4];          // Z~[h][8 channels of this lane] as channel pairs: the current run, the group so far
    float s_acc = 0.f, s_tot = 0.f;
#pragma unroll
    for (int hh = 0; hh < 4; ++hh)
#pragma unroll
        for (int i = 0; i < 4; ++i) { z[hh][i] = f32x2{0.f, 0.f}; zt[hh][i] = f32x2{0.f, 0.f}; }
    auto fold = [&]() {         // group += run; run = 0
#pragma unroll
        for (int hh = 0; hh < 4; ++hh)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                zt[hh][i] += z[hh][i];
                z[hh][i] = f32x2{0.f, 0.f};
            }
        s_tot += s_acc; s_acc = 0.f;
    };

    // the next pair's token row and q' are requested one iteration ahead (two ahead measured no better)
    // Two pairs are in flight per iteration: their chains (row mix -> LayerNorm -> projection -> butterfly ->
    // activation) are independent, so the compiler interleaves them and the 0.7 us latency of one chain is
    // shared by two pairs; Z~ is still updated pair by pair, in order (same bits as one pair at a time).
    f32x4 nx0[2], nx1[2], nqr[2];
    int nri[2] = {0, 0}, nrj[2] = {0, 0};
    const bool pij_lds = EMBED && (p1 - p0) <= PIJ_CAP;
    if (pij_lds) {
        for (int i = threadIdx.x; i < p1 - p0; i += 256) { pij[i] = a.pair_i[p0 + i]; pij[PIJ_CAP + i] = a.pair_j[p0 + i]; }
        __syncthreads();
    }
    auto fetch = [&](int p, int u) {
        const int pc = min(p, a.P - 1);
        const size_t tk = ((size_t)b * a.P + pc) * a.Lloc + lcl;
        if (!EMBED) {
            nx0[u] = *reinterpret_cast<const f32x4*>(a.x + tk * 64 + 8 * cl);
            nx1[u] = *reinterpret_cast<const f32x4*>(a.x + tk * 64 + 8 * cl + 4);
        }
        nqr[u] = *reinterpret_cast<const f32x4*>(a.qrow + tk * 4);
    };
    // ---- the ring (RING > 0) ----
    // slot of pair p: (p - p0) % RING.  Three requests per pair and wave: the lane's two 16-byte halves of its token row
    // (destination = slot base + lane * 16, the layout each lane reads back), and q' as one dword per lane - lane j < 32
    // fetches q'[site j >> 2][j & 3] (lanes 32..63 repeat them into the spare words) -> q'[8 sites][4] contiguous.
    float* const wring = xring + (RING ? wave * RING * CS_SLOT : 0);
    const int lq = min(chunk * 32 + wave * 8 + ((lane & 31) >> 2), a.Lloc - 1);      // the site whose q' this lane fetches
    auto issue = [&](int p, size_t lane_off) {
        const int pc = min(p, a.P - 1);                       // (past the end: a valid pair nobody uses)
        const size_t row = ((size_t)b * a.P + pc) * a.Lloc;
        float* dst = wring + ((p - p0) % RING) * CS_SLOT;
        const float* xs = a.x + row * 64 + lane_off;          // lane_off = (site of the lane) * 64 + 8 * (lane & 7)
        __builtin_amdgcn_global_load_lds(xs, dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(xs + 4, dst + 256, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(a.qrow + (row + lq) * 4 + (lane & 3), dst + 512, 4, 0, 0);
    };
    const size_t lane_off0 = (size_t)lcl * 64 + 8 * cl;
    // Reading the ring.  With all LDS in one object hipcc does not order ordinary LDS reads behind outstanding
    // global_load_lds requests, so these are plain loads the scheduler may place and wait for like any other (an asm
    // block with its own lgkmcnt(0) exposed the LDS latency twice per iteration); what orders them behind the DATA is the
    // counted s_waitcnt vmcnt in front of them (asm volatile with a memory clobber: nothing moves across it).  Loads
    // return in issue order, so "at most 3 (RING - 2) requests outstanding" means the two oldest pairs have landed
    // (later stores and staging requests only make the wait conservative).  The two slots are refilled in the same
    // iteration, and a refill must not land before the reads above have been SERVED (eight waves share the CU's LDS
    // queue; a request that hits in L2 is back in a few hundred cycles): the address of the refill is made to depend on
    // the six vectors just read (an empty asm that consumes them), so hipcc waits for exactly those reads - the pair
    // matrices it requested meanwhile stay in flight - before it can issue the refill.
    auto take = [&](int p, int u) {
        const float* slot = wring + ((p - p0) % RING) * CS_SLOT;
        nx0[u] = *reinterpret_cast<const f32x4*>(slot + lane * 4);
        nx1[u] = *reinterpret_cast<const f32x4*>(slot + 256 + lane * 4);
        nqr[u] = *reinterpret_cast<const f32x4*>(slot + 512 + ts * 4);
    };
    // EMBED: x0 of a pair is two table rows; its residues are requested two iterations ahead and the rows read
    // from LDS one iteration ahead (residue load -> LDS read -> use is a dependent chain: inside one iteration
    // it sat on the critical path whenever the compiler's schedule changed)
    auto fetch_idx = [&](int p, int u) {
        const int pc = min(p, a.P - 1);
        const uint8_t* ib = a.idx + (size_t)b * a.N * a.Lloc + lcl;
        const int pl = min(pc, p1 - 1) - p0;                  // the prefetch past the group's end is never used
        const int si = pij_lds ? (int)pij[pl] : (int)a.pair_i[pc];
        const int sj = pij_lds ? (int)pij[PIJ_CAP + pl] : (int)a.pair_j[pc];
        nri[u] = residue(ib[(size_t)si * a.Lloc]);
        nrj[u] = residue(ib[(size_t)sj * a.Lloc]);
    };
    auto lookup = [&](int u) {
        const float* ei = emb + nri[u] * 64 + 8 * cl;
        const float* ej = emb + nrj[u] * 64 + 8 * cl;
        nx0[u] = *reinterpret_cast<const f32x4*>(ei) + *reinterpret_cast<const f32x4*>(ej);
        nx1[u] = *reinterpret_cast<const f32x4*>(ei + 4) + *reinterpret_cast<const f32x4*>(ej + 4);
    };
    // The per-pair row-attention matrices are staged through LDS 16 pairs at a time (double-buffered):
    // read straight from L2, every 8-lane group of every wave would fetch them again - 10 KB of L1
    // traffic per 8 tokens against 2 KB of x.
    // global_load_lds_dwordx4: 16 bytes per lane go straight from global memory to LDS (destination =
    // wave-uniform base + lane * 16), no staging registers
    auto stage = [&](int pt, int buf) {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.mrow + ((size_t)b * a.P + pt) * MROW);
        const int n4 = min(MT, p1 - pt) * (MROW / 4);
        float* dst = mst + (size_t)buf * MT * MROW;
#pragma unroll
        for (int k = 0; k < MT * MROW / 4 / 256; ++k) {
            const int i = (int)threadIdx.x + 256 * k;
            __builtin_amdgcn_global_load_lds(src + min(i, n4 - 1), dst + (256 * k + 64 * wave) * 4, 16, 0, 0);
        }
    };
    stage(p0, 0);
    if (RING) {
#pragma unroll
        for (int i = 0; i < RING; ++i) issue(p0 + i, lane_off0);
    } else {
        fetch(p0, 0); fetch(p0 + 1, 1);
    }
    if (EMBED) { fetch_idx(p0, 0); fetch_idx(p0 + 1, 1); }
    // (RING: only the staged matrices have to be there - they were requested first, the ring's requests stay in flight)
    if (RING) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (RING ? RING : 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (RING) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } else __syncthreads();
    if (EMBED) { lookup(0); lookup(1); fetch_idx(p0 + 2, 0); fetch_idx(p0 + 3, 1); }
    int buf = 0;
    for (int pt = p0; pt < p1; pt += MT, buf ^= 1) {
      const bool more = pt + MT < p1;
      if (more) stage(pt + MT, buf ^ 1);                   // lands during this tile's compute
      const float* mt = mst + (size_t)buf * MT * MROW;
      const int pe = min(pt + MT, p1);
      for (int p = pt; p < pe; p += 2) {
        const bool two = p + 1 < pe;                       // wave-uniform (an odd group end leaves one pair)
        f32x4 xv0[2], xv1[2], qr[2];
        f32x4 mpre[8];              // RING: the first pair's matrix rows, requested before the refill (see below)
        if (RING) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (RING > 2 ? RING - 2 : 0)) : "memory");
            take(p, 0); take(p + 1, 1);
            // (the refill writes LDS, so hipcc keeps every later LDS read behind it: the first pair's matrix rows are
            // requested HERE, and fly while the refill waits for the ring reads in front of them)
            const float* m = mt + (p - pt) * MROW + 8 * cl;
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) {
                mpre[2 * hh] = *reinterpret_cast<const f32x4*>(m + hh * 64);
                mpre[2 * hh + 1] = *reinterpret_cast<const f32x4*>(m + hh * 64 + 4);
            }
            size_t lo = lane_off0;
            // (the matrix rows are operands too: one wait covers all fourteen reads, and the arithmetic starts right
            // behind the refill's six issue slots instead of behind a second LDS round trip)
            asm("" : "+v"(lo) : "v"(nx0[0]), "v"(nx1[0]), "v"(nqr[0]), "v"(nx0[1]), "v"(nx1[1]), "v"(nqr[1]),
                                "v"(mpre[0]), "v"(mpre[1]), "v"(mpre[2]), "v"(mpre[3]), "v"(mpre[4]), "v"(mpre[5]), "v"(mpre[6]), "v"(mpre[7]));
            issue(p + RING, lo); issue(p + RING + 1, lo);   // into the two slots just read
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) { qr[u] = nqr[u]; xv0[u] = nx0[u]; xv1[u] = nx1[u]; }
        if (EMBED) { lookup(0); lookup(1); fetch_idx(p + 4, 0); fetch_idx(p + 5, 1); }
        if (!RING) {
            fetch(p + 2, 0);
            fetch(p + 3, 1);
        }
        f32x2 d[2][4];
        float act[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            // (for a lone last pair the second chain recomputes the first pair's matrix row: finite, unused)
            const float* m = mt + (two ? (p + u - pt) : (p - pt)) * MROW + 8 * cl;
            f32x2 y[4] = {f32x2{br0[0], br0[1]}, f32x2{br0[2], br0[3]}, f32x2{br1[0], br1[1]}, f32x2{br1[2], br1[3]}};
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) {
                const f32x4 m0 = (RING && u == 0) ? mpre[2 * hh] : *reinterpret_cast<const f32x4*>(m + hh * 64);
                const f32x4 m1 = (RING && u == 0) ? mpre[2 * hh + 1] : *reinterpret_cast<const f32x4*>(m + hh * 64 + 4);
                const f32x2 qh = f32x2{qr[u][hh], qr[u][hh]};
                y[0] = pk_fma(qh, f32x2{m0[0], m0[1]}, y[0]);
                y[1] = pk_fma(qh, f32x2{m0[2], m0[3]}, y[1]);
                y[2] = pk_fma(qh, f32x2{m1[0], m1[1]}, y[2]);
                y[3] = pk_fma(qh, f32x2{m1[2], m1[3]}, y[3]);
            }
            // x' = x + row attention of this block (bias row included)
            d[u][0] = f32x2{xv0[u][0], xv0[u][1]} + y[0];
            d[u][1] = f32x2{xv0[u][2], xv0[u][3]} + y[1];
            d[u][2] = f32x2{xv1[u][0], xv1[u][1]} + y[2];
            d[u][3] = f32x2{xv1[u][2], xv1[u][3]} + y[3];
            const f32x2 s2 = (d[u][0] + d[u][1]) + (d[u][2] + d[u][3]);
            float sm = s2[0] + s2[1];
            sm += dpp_f<0x141>(sm);     // row_half_mirror: lane i <-> 7 - i
            sm += dpp_f<0x1B>(sm);      // quad reverse
            sm += dpp_f<0xB1>(sm);      // xor 1
            const float mean = sm * (1.f / 64.f);
            const f32x2 mean2 = f32x2{mean, mean};
#pragma unroll
            for (int i = 0; i < 4; ++i) d[u][i] -= mean2;
            f32x2 v2 = d[u][0] * d[u][0];
#pragma unroll
            for (int i = 1; i < 4; ++i) v2 = pk_fma(d[u][i], d[u][i], v2);
            float v = v2[0] + v2[1];
            v += dpp_f<0x141>(v);
            v += dpp_f<0x1B>(v);
            v += dpp_f<0xB1>(v);
            const float rstd = __builtin_amdgcn_rsqf(v * (1.f / 64.f) + LN_EPS);
            const f32x2 rstd2 = f32x2{rstd, rstd};
#pragma unroll
            for (int i = 0; i < 4; ++i) d[u][i] *= rstd2;
            // eight 64-long dot products: 8 channels per lane (two outputs per packed FMA, channel after channel), then
            // a transposing butterfly over the 8 lanes of the token (half mirror, quad reverse, xor 1)
            float pv[8];
#pragma unroll
            for (int o2 = 0; o2 < 4; ++o2) {
                f32x2 acc = w2[o2][0] * f32x2{d[u][0][0], d[u][0][0]};
#pragma unroll
                for (int i = 1; i < 8; ++i) acc = pk_fma(w2[o2][i], f32x2{d[u][i >> 1][i & 1], d[u][i >> 1][i & 1]}, acc);
                pv[2 * o2] = acc[0];
                pv[2 * o2 + 1] = acc[1];
            }
            // lane-bit-2 step without selects: two masked DPP adds per value pair (bank_mask picks the 4-lane groups
            // each is meant for; same operands and pairing as keep + dpp(send), see treduce_step4_masked)
            treduce_step4_masked(pv);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float send = up1 ? pv[i] : pv[i + 2], keep = up1 ? pv[i + 2] : pv[i];
                pv[i] = keep + dpp_f<0x1B>(send);
            }
            {
                const float send = up0 ? pv[0] : pv[1], keep = up0 ? pv[1] : pv[0];
                pv[0] = keep + dpp_f<0xB1>(send);
            }
            act[u] = elu1_fast(pv[0] + bj) * vmask;   // q'[cl] or k'[cl - 4]
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            const size_t tok = ((size_t)b * a.P + p + u) * a.Lloc + lcl;
            s_acc += act[u];
            if (lvalid && cl < 4) a.qcol[tok * 4 + cl] = act[u];
            // k'[hh] sits in lane 4 + hh of this token's 8-lane group
            const float k0 = swz<(4 << 5) | 0x18>(act[u]), k1 = swz<(5 << 5) | 0x18>(act[u]),
                        k2 = swz<(6 << 5) | 0x18>(act[u]), k3 = swz<(7 << 5) | 0x18>(act[u]);
            const f32x2 kk[4] = {f32x2{k0, k0}, f32x2{k1, k1}, f32x2{k2, k2}, f32x2{k3, k3}};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int hh = 0; hh < 4; ++hh) z[hh][i] = pk_fma(kk[hh], d[u][i], z[hh][i]);
        }
        // runs of 8 pairs (short groups): the run boundary in the middle of a staged tile (fine = 0 only; wave-uniform)
        if (a.sub == 8 && !a.fine && p + 2 - pt == 8 && p + 2 < pe) fold();
      }
      if (RING) {
          // the next tile's matrices were requested before this tile's ring traffic: once at most the ring's own
          // requests are outstanding they have landed (in-order return) - the ring itself keeps flowing
          asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (RING ? RING : 1)) : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
      } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the staged tile (and the last prefetches) have landed
          __syncthreads();
      }
      if (more && (pt + MT - p0) % a.sub == 0) fold();    // a run boundary inside the group (fine = 0 only)
    }
    fold();
    // (the ring's requests past the end of the walk are still in flight, bound for this block's LDS: they must land
    // before the block gives its LDS back)
    if (RING) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // every wave owns its sites: one partial per (b, g[, run], site), no cross-wave reduction
    if (lvalid) {
        const size_t slot = a.fine ? ((size_t)b * a.G + g) * a.S + run : (size_t)b * a.G + g;
        float* out = a.part + (slot * a.Lloc + l) * CPART;
#pragma unroll
        for (int hh = 0; hh < 4; ++hh)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x4 u = {zt[hh][2 * q][0], zt[hh][2 * q][1], zt[hh][2 * q + 1][0], zt[hh][2 * q + 1][1]};
                *reinterpret_cast<f32x4*>(out + hh * 64 + 8 * cl + 4 * q) = u;
            }
        out[256 + cl] = s_tot;      // S_q[0..3] | S_k[0..3]
    }
}


// ---- column finalisation: partials -> ctx --------------------------------------------------
struct ColFinArgs {
    const float* part;   // [B][G][Lloc][CPART]
    float* ctx;          // [B][Lloc][64]
    const float* wvT;    // [64 c][64 hd] folded col v_proj, transposed
    const float* bv;     // [64] folded col v bias
    int B, Lloc, G;
    float P;
    int npairs, sub, S, fine;   // the association tree of ColStatsArgs (fine = 1: part[b][g][s][l], folded here)
};

constexpr int COLFIN_THREADS = 320;      // >= CPART: one value per thread
__global__ void __launch_bounds__(COLFIN_THREADS) k_colfin(ColFinArgs a) {
    __shared__ float zs[CPART];
    __shared__ float dp[4][64];
    const int site = blockIdx.x;  // b * Lloc + l
    const int b = site / a.Lloc, l = site - b * a.Lloc;
    const size_t gs = (size_t)a.Lloc * CPART;
    const int per = (a.npairs + a.G - 1) / a.G;
    const int i = threadIdx.x;
    // the projection weights first (thread = output hd, a quarter of the 64 inputs): their latency hides
    // behind the partial sums
    const int hd = i & 63, quarter = (i >> 6) & 3;
    float w[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = a.wvT[(16 * quarter + k) * 64 + hd];
    if (i < CPART) {
        // partials are summed in run, then group order (fixed association); up to eight loads in flight
        float acc = 0.f;
        if (a.fine) {
            // four groups a round, four runs of each in flight: up to sixteen loads per thread before the first add
            // (a lone alignment's forward is a chain of such latencies); the sums associate as before - a group's
            // runs in run order from zero, then the groups in group order
            constexpr int GU = 4;
            const float* pp = a.part + ((size_t)b * a.G * a.S * a.Lloc + l) * CPART + i;
            for (int g = 0; g < a.G; g += GU) {
                int n[GU], nmax = 0;
                const float* pg[GU];
                float t[GU];
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int gg = g + u;
                    n[u] = gg < a.G ? (min(a.npairs, (gg + 1) * per) - gg * per + a.sub - 1) / a.sub : 0;
                    nmax = max(nmax, n[u]);
                    pg[u] = pp + (size_t)gg * a.S * gs;
                    t[u] = 0.f;
                }
                for (int s = 0; s < nmax; s += 4) {
                    float v[GU][4];
#pragma unroll
                    for (int u = 0; u < GU; ++u)
#pragma unroll
                        for (int q = 0; q < 4; ++q)      // (uniform conditions; x + 0 leaves every bit of x)
                            v[u][q] = s + q < n[u] ? pg[u][(size_t)(s + q) * gs] : 0.f;
#pragma unroll
                    for (int u = 0; u < GU; ++u) t[u] = (((t[u] + v[u][0]) + v[u][1]) + v[u][2]) + v[u][3];
                }
#pragma unroll
                for (int u = 0; u < GU; ++u)
                    if (g + u < a.G) acc += t[u];
            }
        } else {
            const float* pp = a.part + ((size_t)b * a.G * a.Lloc + l) * CPART + i;
            int g = 0;
            for (; g + 4 <= a.G; g += 4) {
                const float v0 = pp[(size_t)g * gs], v1 = pp[(size_t)(g + 1) * gs], v2 = pp[(size_t)(g + 2) * gs],
                            v3 = pp[(size_t)(g + 3) * gs];
                acc = (((acc + v0) + v1) + v2) + v3;
            }
            for (; g < a.G; ++g) acc += pp[(size_t)g * gs];
        }
        zs[i] = acc;
    }
    __syncthreads();
    if (i < 256) {
        const float* z = zs + (hd >> 4) * 64 + 16 * quarter;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fmaf(w[k], z[k], acc);
        dp[quarter][hd] = acc;
    }
    __syncthreads();
    if (i < 64) {
        const int hh = hd >> 4;
        const float acc = ((dp[0][hd] + dp[1][hd]) + dp[2][hd]) + dp[3][hd];
        const float sq = zs[256 + hh], sk = zs[260 + hh];
        a.ctx[(size_t)site * 64 + hd] = (acc + a.bv[hd] * sk) / sk * (a.P / sq);
    }
}

// dst += src (stands in for the all-reduce in the single-GPU shard emulation)
__global__ void k_accumulate(float* dst, const float* src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// ---- self test of the hardware-layout assumptions ------------------------------------------
// out[0..63]: pair_sum(lane)        expected lane%32 + lane%32+32
// out[64..127]: pair_other(lane)    expected lane ^ 32
// out[128..191]: row16_sum(lane)    expected sum of the 16-lane row
// out[192..255]: half32_sum(lane)   expected sum of the 32-lane half
// out[256..1279]:  D1 = A1*B, A1[m][k] = m + 32*(k%8), B[k][n] = (k == n%16)  -> m + 32*((n%16)%8)
// out[1280..2303]: D2 = A2*B, A2[m][k] = k                                    -> n%16
// both in C/D register order: out[.. + lane*16 + r] = D[(r&3) + 8*(r>>2) + 4*(lane>>5)][lane&31]
__global__ void k_selftest(float* out) {
    const int lane = threadIdx.x;
    out[lane] = pair_sum((float)lane);
    out[64 + lane] = pair_other((float)lane, lane >> 5);
    out[128 + lane] = row16_sum((float)lane);
    out[192 + lane] = half32_sum((float)lane);
    frag_t A1, A2, Bf;
    const int m = lane & 31, kg = lane >> 5;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = 8 * kg + i;
        A1[i] = (h16_t)(float)(m + 32 * (k % 8));      // exact in bf16 and fp16 (< 256)
        A2[i] = (h16_t)(float)k;
        Bf[i] = (h16_t)((k == (m & 15)) ? 1.f : 0.f);  // B[k][n = m] selects k == n % 16
    }
    f32x16 acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc2[r] = 0.f; }
    acc1 = PF_MFMA(A1, Bf, acc1);
    acc2 = PF_MFMA(A2, Bf, acc2);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        out[256 + lane * 16 + r] = acc1[r];
        out[1280 + lane * 16 + r] = acc2[r];
    }
}

#endif  // PF_DEVICE_HELPERS_ONLY

}  // namespace pfk
