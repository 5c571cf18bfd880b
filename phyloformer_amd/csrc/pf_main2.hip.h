// k_main2: the dominant kernel in the one-wave-per-SIMD regime (gfx950, 512 registers per lane).
//
// Same arithmetic, buffers and per-tile statistics as k_main (pf_device.hip.h); what changes is how the
// work is fed to the SIMD.  A 256-thread workgroup puts ONE wave on each SIMD; that wave owns TWO 32-token
// tiles (consecutive work items of its chunk) at a time:
//   * the FFN hidden loop - 72 % of the MFMAs and half of the vector work - is the hand-placed instruction
//     stream of pf_hidden_asm.inc (generator: tools/gen_hidden_asm.py): the two tiles are skewed by half a
//     hidden-tile step, so the GEMM2(T) + GEMM1(T+1) MFMAs of one tile issue between the GELU + bf16-split
//     instructions of the other, consecutive MFMAs always on different accumulators, GEMM1 accumulators in
//     VGPRs (no v_accvgpr_read per hidden value), one MFMA every 7th issue slot;
//   * the phases around it (attention apply, LayerNorm, split, next-row statistics / head) are the C++ of
//     k_main, instantiated for both tiles in one basic block so that the compiler interleaves two
//     independent streams where a second wave used to hide latencies.
// Results are bit-identical to k_main: the hidden loop performs the same operations in the same order per
// accumulator (tools/ffn3_bench.hip checks the stream against the C++ loop bit for bit).
#pragma once
#include "pf_device.hip.h"
#include "pf_hidden_asm.inc"

namespace pfk {

constexpr int MAIN2_THREADS = 256;   // 4 waves: one per SIMD
constexpr int MAIN2_WAVES = MAIN2_THREADS / 64;
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ u32x16 pack_frags4(const bf16x8 (&f)[4]) {
    u32x16 r;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const u32x4 q = __builtin_bit_cast(u32x4, f[s]);
#pragma unroll
        for (int i = 0; i < 4; ++i) r[4 * s + i] = q[i];
    }
    return r;
}

// per-tile state that lives across the phases of one iteration
struct Tile2 {
    int row, tile, b;     // b * P + p, index of the 32-site tile inside the row, alignment
    long task;            // work item (partial-statistics slot)
    bool live;            // false: ghost tile (odd chunk tail) - computes on clamped data, stores go to the trash area
    bool valid;           // this lane's site exists
    size_t tok, stok;     // token index for loads (clamped) and for stores (trash for lanes past the row end)
    float x[32];
};

template <int MODE>
__global__ void __launch_bounds__(MAIN2_THREADS, 1) k_main2(MainArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_frag_t lw = (lds_frag_t)smem;
    lds_f32_t lc = (lds_f32_t)(smem + FRAG_END * 16);
    {
        uint4* dst = reinterpret_cast<uint4*>(smem);
        const uint4* src = reinterpret_cast<const uint4*>(a.wimg);
        const int hi = (MODE == MODE_LAST) ? FRAG_WV : FRAG_END;
        for (int i = threadIdx.x; i < hi; i += MAIN2_THREADS) dst[i] = src[i];
        float* dc = reinterpret_cast<float*>(smem + FRAG_END * 16);
        for (int i = threadIdx.x; i < CONST_LEN; i += MAIN2_THREADS) dc[i] = a.consts[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = lane & 31;
    const int h = lane >> 5;
    const int ntiles = (a.Lloc + 31) >> 5;
    const long ntasks = (long)a.B * a.P * ntiles;
    lds_frag_t wop = lw + FRAG_WO + lane;
    lds_frag_t wvp = lw + FRAG_WV + lane;
    lds_frag_t qkp = lw + FRAG_QK + h * 8 + (t & 7);
    lds_f32_t lch = lc + 4 * h;
    PF_OPAQUE(wop); PF_OPAQUE(wvp); PF_OPAQUE(qkp); PF_OPAQUE(lch);

    const long nwaves = (long)gridDim.x * MAIN2_WAVES;
    const long chunk = (ntasks + nwaves - 1) / nwaves;
    const long task0 = (long)(blockIdx.x * MAIN2_WAVES + wave) * chunk;
    const long task1 = min(ntasks, task0 + chunk);

    // position of a work item, advanced incrementally (one division per chunk, none per tile)
    struct Pos { int row, tile, b, p; };
    auto advance = [&](Pos& q) {
        if (++q.tile == ntiles) {
            q.tile = 0; ++q.row;
            if (++q.p == a.P) { q.p = 0; ++q.b; }
        }
    };
    auto locate = [&](Tile2& T, const Pos& q, long task, bool live) {
        T.task = task;
        T.live = live;
        T.row = q.row; T.tile = q.tile; T.b = q.b;
        const int l = q.tile * 32 + t;
        T.valid = l < a.Lloc;
        const int lc_ = T.valid ? l : a.Lloc - 1;
        T.tok = (size_t)q.row * a.Lloc + lc_;
        T.stok = (T.valid && live) ? T.tok : a.trash_tok + t;
    };
    // everything one iteration reads from global memory for its two tiles: requested one iteration ahead
    // (in front of the hidden loop of the previous pair), so a pair never starts by waiting on HBM
    // (the residual rows, q' and the row-mix fragments; ctx is L2-resident and is read where it is used - 64 registers less to carry across the hidden loop)
    struct Loads {
        f32x4 px[2][8], pqr[2], pqc[2];
        int ri[2], rj[2];
    };
    auto request = [&](Loads& L, const Tile2 (&T)[2], const Pos (&q)[2]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const Tile2& Tu = T[u];
            const int ll = (int)(Tu.tok - (size_t)Tu.row * a.Lloc);
            if (MODE == MODE_MID0) {
                const uint8_t* ib = a.idx + (size_t)Tu.b * a.N * a.Lloc + ll;
                L.ri[u] = ib[(size_t)a.pair_i[q[u].p] * a.Lloc];
                L.rj[u] = ib[(size_t)a.pair_j[q[u].p] * a.Lloc];
            } else {
                const f32x4* xp = reinterpret_cast<const f32x4*>(a.x + Tu.tok * 64 + 4 * h);
#pragma unroll
                for (int g = 0; g < 8; ++g) L.px[u][g] = xp[2 * g];
            }
            L.pqr[u] = *reinterpret_cast<const f32x4*>(a.qrow + Tu.tok * 4);
            L.pqc[u] = *reinterpret_cast<const f32x4*>(a.qcol + Tu.tok * 4);
        }
    };

    Pos pos[2];
    Tile2 T[2];
    Loads L;
    if (task0 < task1) {
        pos[0].row = (int)(task0 / ntiles);
        pos[0].tile = (int)(task0 - (long)pos[0].row * ntiles);
        pos[0].b = pos[0].row / a.P;
        pos[0].p = pos[0].row - pos[0].b * a.P;
        pos[1] = pos[0];
        if (task0 + 1 < task1) advance(pos[1]);
        locate(T[0], pos[0], task0, true);
        locate(T[1], pos[1], (task0 + 1 < task1) ? task0 + 1 : task0, task0 + 1 < task1);
        request(L, T, pos);
    }

    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
#define PF_TICK(k) do { if (a.prof) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tacc[k] += tn_ - tprev; tprev = tn_; } } while (0)
    if (a.prof) tprev = __builtin_amdgcn_s_memtime();
    for (long task = task0; task < task1; task += 2) {
        // ---- this pair's data (requested during the previous pair's hidden loop)
        f32x4 pctx[2][8], pqr[2], pqc[2];
        bf16x8 mfr[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (MODE == MODE_MID0) {
                // embedding lookup + pair expansion: the same fp32 sums k_embed forms (table is L1-resident)
                const f32x4* ti = reinterpret_cast<const f32x4*>(a.table + L.ri[u] * 64 + 4 * h);
                const f32x4* tj = reinterpret_cast<const f32x4*>(a.table + L.rj[u] * 64 + 4 * h);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const f32x4 v0 = ti[2 * g], v1 = tj[2 * g];
#pragma unroll
                    for (int i = 0; i < 4; ++i) T[u].x[4 * g + i] = v0[i] + v1[i];
                }
            } else {
#pragma unroll
                for (int g = 0; g < 8; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) T[u].x[4 * g + i] = L.px[u][g][i];
            }
            pqr[u] = L.pqr[u]; pqc[u] = L.pqc[u];
            const int ll = (int)(T[u].tok - (size_t)T[u].row * a.Lloc);
            const f32x4* cp = reinterpret_cast<const f32x4*>(a.ctx + ((size_t)T[u].b * a.Lloc + ll) * 64 + 4 * h);
#pragma unroll
            for (int g = 0; g < 8; ++g) pctx[u][g] = cp[2 * g];
            const bf16x8* mf = a.mfrag + (size_t)T[u].row * 128 + t;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) mfr[u][q4] = mf[q4 * 32];
        }
        // the pair after this one
        Pos npos[2] = {pos[1], pos[1]};
        Tile2 NT[2];
        const bool more = task + 2 < task1;
        if (more) {
            advance(npos[0]);
            npos[1] = npos[0];
            if (task + 3 < task1) advance(npos[1]);
            locate(NT[0], npos[0], task + 2, true);
            locate(NT[1], npos[1], (task + 3 < task1) ? task + 3 : task + 2, task + 3 < task1);
        }

        PF_TICK(0);
        // ---- attention apply of block k (row mix incl. both out_proj biases + column out_proj), both tiles
        bf16x8 xb_hi[2][4], xb_lo[2][4];
        f32x16 oa[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            Tile2& Tu = T[u];
            f32x16 ya[2];
#pragma unroll
            for (int j = 0; j < 32; ++j) ya[j >> 4][j & 15] = Tu.x[j];
            {
                const f32x4 qr = pqr[u];
                float v[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (h == 0) ? qr[i] : 0.f;
                v[4] = v[5] = (h == 0) ? 1.f : 0.f;
                v[6] = v[7] = 0.f;
                bf16x8 qb_hi, qb_lo;
                split8(v, qb_hi, qb_lo);
#pragma unroll
                for (int To = 0; To < 2; ++To) mfma3(ya[To], mfr[u][To * 2], mfr[u][To * 2 + 1], qb_hi, qb_lo, To == 1);
            }
            {
                const f32x4 qc = pqc[u];
                float o[32];
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const f32x4 w4 = pctx[u][g];
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[4 * g + i] = w4[i] * qc[g >> 1];
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    bf16x8 ob_hi, ob_lo;
                    split8(&o[8 * s], ob_hi, ob_lo);
#pragma unroll
                    for (int To = 0; To < 2; ++To) {
                        lds_frag_t f = wop + ((To * 4 + s) * 2) * 64;
                        const bf16x8 a_hi = f[0], a_lo = f[64];
                        mfma3(ya[To], a_hi, a_lo, ob_hi, ob_lo, To == 1);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 32; ++j) Tu.x[j] = ya[j >> 4][j & 15];
            // ---- feed-forward input: LayerNorm + split; GEMM2 accumulators start from residual + b2
            {
                float xn[32];
                ln_pair(Tu.x, xn);
#pragma unroll
                for (int s = 0; s < 4; ++s) split8(&xn[8 * s], xb_hi[u][s], xb_lo[u][s]);
            }
            load_acc_bias(oa[u][0], lch + CONST_B2);
            load_acc_bias(oa[u][1], lch + CONST_B2 + 32);
#pragma unroll
            for (int j = 0; j < 32; ++j) oa[u][j >> 4][j & 15] += Tu.x[j];
        }

        if (more) request(L, NT, npos);      // lands during the hidden loop
        PF_TICK(1);

        // ---- the hidden loop of both tiles: one hand-placed instruction stream (pf_hidden_asm.inc)
        {
            const u32x16 xAh = pack_frags4(xb_hi[0]), xAl = pack_frags4(xb_lo[0]);
            const u32x16 xBh = pack_frags4(xb_hi[1]), xBl = pack_frags4(xb_lo[1]);
            int aw1 = lane * 16, aw2 = FRAG_W2 * 16 + lane * 16, ab = FRAG_END * 16 + (CONST_B1 + 4 * h) * 4;
            int tcount;
            const float c4 = 0.0136151873f;          // gelu_scaled()'s polynomial: c5 u + c4, then c3 .. c0
            f32x16 o0, o1, o2, o3;
            asm volatile(PF_HID2_ASM
                         : PF_HID2_OUT0_A(o0), PF_HID2_OUT1_A(o1), PF_HID2_OUT0_B(o2), PF_HID2_OUT1_B(o3),
                           PF_HID2_INIT0_A(oa[0][0]), PF_HID2_INIT1_A(oa[0][1]), PF_HID2_INIT0_B(oa[1][0]),
                           PF_HID2_INIT1_B(oa[1][1]), PF_HID2_AW1(aw1), PF_HID2_AW2(aw2), PF_HID2_AB(ab),
                           [t] "=&s"(tcount)
                         : PF_HID2_XH_A(xAh), PF_HID2_XL_A(xAl), PF_HID2_XH_B(xBh), PF_HID2_XL_B(xBl), PF_HID2_C4(c4),
                           [c5] "s"(-0.00107098569f), [c3] "s"(-0.084594565f), [c2] "s"(-0.637684925f),
                           [c1] "s"(-1.35494915f), [c0] "s"(-0.00003762f), [sl] "s"(0x0000bf80u), [sh] "s"(0xbf800000u)
                         : PF_HID2_CLOBBERS, "scc");
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                T[0].x[j] = o0[j]; T[0].x[16 + j] = o1[j];
                T[1].x[j] = o2[j]; T[1].x[16 + j] = o3[j];
            }
        }

        PF_TICK(2);
        // ---- behind the FFN: store + next block's row statistics, or the head
        // (keeping these 32 registers resident for the whole kernel, or carrying ctx / the row-mix fragments of
        // the next pair across the hidden loop, makes hipcc spill around the pinned registers of the asm block:
        // measured 4.45 / 4.70 ms against 4.36 ms for this form)
        bf16x8 wl[8];
        if (MODE != MODE_LAST) {
#pragma unroll
            for (int i = 0; i < 8; ++i) wl[i] = a.wv_lo[i * 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            Tile2& Tu = T[u];
            if (MODE != MODE_LAST) {
                {
                    f32x4* xo = reinterpret_cast<f32x4*>(a.x + Tu.stok * 64 + 4 * h);
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const f32x4 v4 = {Tu.x[4 * g], Tu.x[4 * g + 1], Tu.x[4 * g + 2], Tu.x[4 * g + 3]};
                        xo[2 * g] = v4;
                    }
                }
                f32x16 va[3];
                {
                    float xn[32];
                    ln_pair(Tu.x, xn);
                    bf16x8 nb_hi[4], nb_lo[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) split8(&xn[8 * s], nb_hi[s], nb_lo[s]);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const bf16x8 q_hi = qkp[(s * 2) * 16], q_lo = qkp[(s * 2 + 1) * 16];
                        if (s == 0) mfma3_zero(va[2], q_hi, q_lo, nb_hi[s], nb_lo[s]);
                        else mfma3(va[2], q_hi, q_lo, nb_hi[s], nb_lo[s]);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int Tt = 0; Tt < 2; ++Tt) {
                            const bf16x8 f_hi = wvp[(Tt * 4 + s) * 64];
                            if (s == 0) mfma3_zero(va[Tt], f_hi, wl[Tt * 4 + s], nb_hi[s], nb_lo[s], Tt == 1);
                            else mfma3(va[Tt], f_hi, wl[Tt * 4 + s], nb_hi[s], nb_lo[s], Tt == 1);
                        }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) va[2][r] += lc[CONST_BQK + 4 * h + r];   // rows 0-3 q, 4-7 k
                float qk[4], ot[4], qn[4], kn[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) qk[i] = elu1_fast(va[2][i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) ot[i] = pair_other(qk[i], h);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    qn[i] = h ? ot[i] : qk[i];
                    kn[i] = h ? qk[i] : ot[i];
                }
                const float vm = Tu.valid ? 1.f : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) kn[i] *= vm;
                {
                    const f32x4 qs = {qn[0], qn[1], qn[2], qn[3]};
                    *reinterpret_cast<f32x4*>(a.qrow + Tu.stok * 4) = qs;
                }
                float qk8[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) { qk8[i] = vm * qn[i]; qk8[4 + i] = kn[i]; }
                const float sqk = treduce8(qk8, t);
                float kv[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) kv[j] = kn[j >> 3] * va[j >> 4][j & 15];
                const float skv = treduce32(kv, t);
                if (Tu.live) {                                   // wave-uniform
                    float* sp = a.spart + (size_t)Tu.task * SROW;
                    if (lane < 8) sp[64 + lane] = sqk;
                    sp[kmap(t, h)] = skv;
                }
            } else {
                float z = 0.f;
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const f32x4 w4 = *(lds_f32x4_t)(lch + CONST_HW + 8 * g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) z = fmaf(w4[i], Tu.x[4 * g + i], z);
                }
                z = pair_sum(z) + lc[CONST_HB];
                const float so = half32_sum(Tu.valid ? softplus20(z) : 0.f);
                if (Tu.live && lane == 0) a.outpart[Tu.task] = so;
                if (a.store_x_last && Tu.live && Tu.valid) {
                    f32x4* xo = reinterpret_cast<f32x4*>(a.x + Tu.tok * 64 + 4 * h);
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const f32x4 v4 = {Tu.x[4 * g], Tu.x[4 * g + 1], Tu.x[4 * g + 2], Tu.x[4 * g + 3]};
                        xo[2 * g] = v4;
                    }
                }
            }
        }
        if (more) {
            pos[0] = npos[0]; pos[1] = npos[1];
            T[0] = NT[0]; T[1] = NT[1];
        }
        PF_TICK(3);
    }
    if (a.prof && lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(a.prof + k, tacc[k]);
    }
#undef PF_TICK
}

}  // namespace pfk
