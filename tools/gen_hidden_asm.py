#!/usr/bin/env python3
"""Generator of the hand-scheduled FFN hidden loop of k_main (gfx950 inline asm).

    python tools/gen_hidden_asm.py 6 base two > phyloformer_amd/csrc/pf_hidden_asm.inc     (product include)
    python tools/gen_hidden_asm.py <fillers per MFMA gap> <mode> > /tmp/hid.inc                (tools/ffn3_bench.hip)

The hidden loop (model.py:101-104, 64 -> 256 -> 64 with erf-GELU) is 72 % of k_main's MFMAs and half
of its VALU work.  hipcc schedules it as MFMA clusters followed by VALU clusters and parks the GEMM1
accumulators in AGPRs (one v_accvgpr_read per hidden value); here every instruction is placed by hand:
one `v_mfma_f32_32x32x16_bf16` followed by a fixed number of VALU / LDS fillers, consecutive MFMAs on
different accumulators, all operands in pinned physical registers.

Two register maps share the generator:

  two-tile  (one wave per SIMD, 512 registers): tiles A and B skewed by half a hidden-tile step; while the
            matrix pipe runs GEMM2(X, T) + GEMM1(X, T+1) of one tile the VALU evaluates GELU + bf16 split
            of the other tile's hidden tile T.
  one-tile  (two waves per SIMD, 256 registers): the same stream software-pipelined over T within one
            tile (GEMM2(T-1) + GEMM1(T+1) beside GELU(T)), ha / g ping-pong by parity.

LDS image offsets follow pf_device.hip.h (FRAG_W1 = 0, FRAG_W2 = 4096 fragments, constants behind
FRAG_END); the K order of the fragments is the lane mapping kmap() described there.
"""
import sys

FRAG_END_BYTES = (4096 + 4096 + 1024 + 512 + 128) * 16      # FRAG_END * 16 (pf_device.hip.h)


class Regs:
    """Pinned physical registers of one variant."""

    def __init__(self, two_tile, vtop=255, atop=255):
        self.two_tile = two_tile
        v = vtop

        def take(n):
            nonlocal v
            v -= n
            lo = v + 1
            assert lo % 4 == 0 or n == 1, (lo, n)
            return lo

        tiles = ("A", "B") if two_tile else ("A",)
        self.xh, self.xl, self.ha, self.g = {}, {}, {}, {}
        for t in tiles:
            self.xh[t] = take(16)
            self.xl[t] = take(16)
        # GEMM1 accumulators (VGPR: the VALU reads them) and GELU outputs (hi[2] | lo[2] fragments)
        for t in ("A", "B"):
            self.ha[t] = take(16)
            self.g[t] = take(16)
        self.bias = take(16)
        self.tmp = take(16)
        self.aw1 = take(1)
        self.aw2 = take(1)
        self.ab = take(1)
        self.c4 = take(1)
        self.vlo = v + 1                      # lowest pinned VGPR
        a = atop

        def takea(n):
            nonlocal a
            a -= n
            return a + 1

        self.oa = {}
        for t in tiles:
            self.oa[t] = (takea(16), takea(16))
        self.F = (takea(16), takea(16))       # fragment double buffer: ah, al, wh, wl (4 regs each)
        self.alo = a + 1

    def vt(self, lo, n):
        return f"v[{lo}:{lo + n - 1}]" if n > 1 else f"v{lo}"

    def at(self, lo, n):
        return f"a[{lo}:{lo + n - 1}]"


MODE = "base"      # experiment switch (tools/ffn3_bench.hip): base | nodot | nosplit | poly3 | movonly | fmaonly | mfmaonly
NOLDS = False      # "<mode>+nolds": no fragment reads (the MFMAs reuse whatever the fragment registers hold)


def schedule_gelu(R, y):
    """List-schedule GELU + split of the 16 accumulator values of tile y into one VALU stream.
    Greedy: at every slot issue the ready op with the longest remaining chain; a transcendental is never
    issued right behind another one.  Returns the instruction list."""
    ha, g = R.ha[y], R.g[y]
    ops = []      # (id, text, deps[(id, dist)], trans, prio)
    ids = {}

    def add(key, text, deps, trans, prio):
        ids[key] = len(ops)
        ops.append((text, [(ids[d], dist) for d, dist in deps], trans, prio))

    if MODE == "mfmaonly":
        return []
    if MODE == "movonly":
        return [f"v_mov_b32 v{R.tmp + (i % 16)}, v{ha + (i % 16)}" for i in range(160)]
    if MODE == "fmaonly":      # the same 160 slots as plain fp32 FMAs (three register reads, no transcendental)
        return [f"v_fma_f32 v{R.tmp + (i % 16)}, v{ha + (i % 16)}, v{R.c4}, v{R.tmp + ((i + 5) % 16)}" for i in range(160)]
    for r in range(16):
        x, p = f"v{ha + r}", f"v{R.tmp + r}"
        if MODE == "poly3":
            add(("p1", r), f"v_fma_f32 {p}, |{x}|, %[c5], v{R.c4}", [], False, 12)
            add(("p5", r), f"v_fma_f32 {p}, {p}, |{x}|, %[c0]", [(("p1", r), 2)], False, 8)
            add(("ex", r), f"v_exp_f32 {p}, {p}", [(("p5", r), 2)], True, 7)
            add(("sb", r), f"v_sub_f32 {p}, 1.0, {p}", [(("ex", r), 3)], False, 6)
            add(("gg", r), f"v_fma_f32 {x}, |{x}|, {p}, {x}", [(("sb", r), 2)], False, 5)
            continue
        add(("p1", r), f"v_fma_f32 {p}, |{x}|, %[c5], v{R.c4}", [], False, 12)
        add(("p2", r), f"v_fma_f32 {p}, {p}, |{x}|, %[c3]", [(("p1", r), 2)], False, 11)
        add(("p3", r), f"v_fma_f32 {p}, {p}, |{x}|, %[c2]", [(("p2", r), 2)], False, 10)
        add(("p4", r), f"v_fma_f32 {p}, {p}, |{x}|, %[c1]", [(("p3", r), 2)], False, 9)
        add(("p5", r), f"v_fma_f32 {p}, {p}, |{x}|, %[c0]", [(("p4", r), 2)], False, 8)
        if MODE == "noexp":
            add(("ex", r), f"v_mul_f32 {p}, {p}, {p}", [(("p5", r), 2)], False, 7)
        else:
            add(("ex", r), f"v_exp_f32 {p}, {p}", [(("p5", r), 2)], True, 7)
        add(("sb", r), f"v_sub_f32 {p}, 1.0, {p}", [(("ex", r), 3)], False, 6)
        add(("gg", r), f"v_fma_f32 {x}, |{x}|, {p}, {x}", [(("sb", r), 2)], False, 5)
    for k in range(8):
        r0, r1 = 2 * k, 2 * k + 1
        # fragment u = k // 4 holds values 8u .. 8u+7: hi dword k % 4 of g_hi[u], lo likewise
        u, d = k // 4, k % 4
        hi = f"v{g + 4 * u + d}"
        lo = f"v{g + 8 + 4 * u + d}"
        add(("hi", k), f"v_cvt_pk_bf16_f32 {hi}, v{ha + r0}, v{ha + r1}", [(("gg", r0), 2), (("gg", r1), 2)], False, 4)
        if MODE == "nosplit":
            add(("lo", k), f"v_mov_b32 {lo}, {hi}", [(("hi", k), 2)], False, 2)
            continue
        if MODE == "nodot":
            t0, t1 = f"v{R.tmp + r0}", f"v{R.tmp + r1}"
            add(("u0", k), f"v_lshlrev_b32 {t0}, 16, {hi}", [(("hi", k), 2)], False, 3)
            add(("u1", k), f"v_and_b32 {t1}, %[sh], {hi}", [(("hi", k), 2)], False, 3)      # %[sh] = 0xffff0000 here
            add(("d0", k), f"v_sub_f32 v{ha + r0}, v{ha + r0}, {t0}", [(("u0", k), 2)], False, 3)
            add(("d1", k), f"v_sub_f32 v{ha + r1}, v{ha + r1}, {t1}", [(("u1", k), 2)], False, 3)
            add(("lo", k), f"v_cvt_pk_bf16_f32 {lo}, v{ha + r0}, v{ha + r1}", [(("d0", k), 2), (("d1", k), 2)], False, 2)
            continue
        add(("d0", k), f"v_dot2c_f32_bf16 v{ha + r0}, %[sl], {hi}", [(("hi", k), 2)], False, 3)
        add(("d1", k), f"v_dot2c_f32_bf16 v{ha + r1}, %[sh], {hi}", [(("hi", k), 2)], False, 3)
        add(("lo", k), f"v_cvt_pk_bf16_f32 {lo}, v{ha + r0}, v{ha + r1}", [(("d0", k), 4), (("d1", k), 4)], False, 2)
    n = len(ops)
    done_at = [None] * n
    out = []
    slot = 0
    last_trans = -10
    remaining = set(range(n))
    while remaining:
        best = None
        for i in sorted(remaining):
            text, deps, trans, prio = ops[i]
            if any(done_at[d] is None or slot - done_at[d] < dist for d, dist in deps):
                continue
            if trans and slot - last_trans < 2:
                continue
            if best is None or prio > ops[best][3]:
                best = i
        if best is None:
            out.append(None)          # bubble: the caller fills it with something else or an s_nop
        else:
            out.append(ops[best][0])
            done_at[best] = slot
            if ops[best][2]:
                last_trans = slot
            remaining.discard(best)
        slot += 1
    return out


def half_step(R, X, Y, g2, g1, gelu, w1_off, w2_off, b_off, load_bias_next, first_frags_loaded,
              next_frags, fill, delay_gaps=0, extra=None):
    """Instruction list of one half step.
      X: tile whose GEMMs run (g2: GEMM2 of hidden tile T from g[X]; g1: GEMM1 of T+1 into ha[X]),
      Y: tile whose GELU + split runs (ha[Y] -> g[Y]) or None,
      w1_off / w2_off / b_off: immediate byte offsets relative to the address registers,
      load_bias_next: byte offset of the bias block to load into R.bias during this half step (or None),
      next_frags: (g2n, g1n, w1n, w2n) of the half step that follows (its step-0 fragments are requested
                  during this one's last step), or None,
      fill: VALU fillers per MFMA gap,
      delay_gaps: leading MFMA gaps left without VALU (the GELU source was written by the MFMAs right
                  in front of this half step: their results need ~11 wait states before a VALU read),
      extra: filler instructions used instead of a GELU stream (accumulator moves of prologue / epilogue)."""
    ins = []
    oa0, oa1 = R.oa[X] if R.two_tile else R.oa["A"]
    xh = R.xh[X] if R.two_tile else R.xh["A"]
    xl = R.xl[X] if R.two_tile else R.xl["A"]
    ha, gx = R.ha[X], R.g[X]
    F = R.F

    def frag_reads(buf, i, g2_, g1_, w1o, w2o):
        rd = []
        u, To = i >> 1, i & 1
        if NOLDS:
            return rd
        if g1_:
            rd.append(f"ds_read_b128 {R.at(F[buf] + 0, 4)}, v{R.aw1} offset:{w1o + i * 2048}")
            rd.append(f"ds_read_b128 {R.at(F[buf] + 4, 4)}, v{R.aw1} offset:{w1o + i * 2048 + 1024}")
        if g2_:
            rd.append(f"ds_read_b128 {R.at(F[buf] + 8, 4)}, v{R.aw2} offset:{w2o + To * 32768 + u * 2048}")
            rd.append(f"ds_read_b128 {R.at(F[buf] + 12, 4)}, v{R.aw2} offset:{w2o + To * 32768 + u * 2048 + 1024}")
        return rd

    valu = schedule_gelu(R, Y) if (gelu and Y is not None) else []
    if extra:
        valu = list(extra) + valu
    vpos = 0

    def take_fill(k):
        nonlocal vpos
        got = []
        while len(got) < k and vpos < len(valu):
            v = valu[vpos]
            vpos += 1
            got.append(v if v is not None else "s_nop 0")
        return got

    if not first_frags_loaded:
        ins += frag_reads(0, 0, g2, g1, w1_off, w2_off)
    nm = (3 if g1 else 0) + (3 if g2 else 0)
    for i in range(4):
        buf = i & 1
        u, To = i >> 1, i & 1
        pending = []
        if i < 3:
            pending = frag_reads(buf ^ 1, i + 1, g2, g1, w1_off, w2_off)
        elif next_frags is not None:
            pending = frag_reads(buf ^ 1, 0, *next_frags)
        if load_bias_next is not None and i == 2:
            pending += [f"ds_read_b128 {R.vt(R.bias + 4 * q, 4)}, v{R.ab} offset:{load_bias_next + 32 * q}" for q in range(4)]
        ins.append("s_waitcnt lgkmcnt(0)")
        ah, al = R.at(F[buf] + 0, 4), R.at(F[buf] + 4, 4)
        wh, wl = R.at(F[buf] + 8, 4), R.at(F[buf] + 12, 4)
        hacc = R.vt(ha, 16)
        csrc = R.vt(R.bias, 16) if i == 0 else hacc
        mf = []
        # pass order as mfma3() in pf_device.hip.h: (hi,lo) (hi,hi) (lo,hi); GEMM2's second output tile of a K
        # step reversed, so that consecutive MFMAs on one accumulator chain share an operand (bit-identical sums)
        g1m = [
            f"v_mfma_f32_32x32x16_bf16 {hacc}, {ah}, {R.vt(xl + 4 * i, 4)}, {csrc}",
            f"v_mfma_f32_32x32x16_bf16 {hacc}, {ah}, {R.vt(xh + 4 * i, 4)}, {hacc}",
            f"v_mfma_f32_32x32x16_bf16 {hacc}, {al}, {R.vt(xh + 4 * i, 4)}, {hacc}",
        ]
        oacc = R.at(oa1 if To else oa0, 16)
        ghi, glo = R.vt(gx + 4 * u, 4), R.vt(gx + 8 + 4 * u, 4)
        g2m = [
            f"v_mfma_f32_32x32x16_bf16 {oacc}, {wh}, {glo}, {oacc}",
            f"v_mfma_f32_32x32x16_bf16 {oacc}, {wh}, {ghi}, {oacc}",
            f"v_mfma_f32_32x32x16_bf16 {oacc}, {wl}, {ghi}, {oacc}",
        ]
        if To:
            g2m.reverse()
        for k in range(3):
            if g1:
                mf.append(g1m[k])
            if g2:
                mf.append(g2m[k])
        for m in mf:
            ins.append(m)
            if pending:
                ins.append(pending.pop(0))
            if delay_gaps > 0:
                delay_gaps -= 1
            else:
                ins += take_fill(fill)
        ins += pending
    # whatever VALU is left (fill too small) goes behind the last MFMA
    ins += take_fill(10 ** 6)
    return ins


def emit_block(name, lines, R):
    """One asm volatile statement as a C++ macro body (operands are supplied by the including file)."""
    print(f"#define {name} \\")
    for ln in lines:
        print(f'    "{ln}\\n\\t" \\')
    print("    \"\"")
    print()


def clobbers(R):
    regs = []
    for t in R.ha:
        if t not in R.xh:           # ha / g of the tiles in R.xh are in-out operands (accumulator start values)
            regs += [f"v{R.ha[t] + i}" for i in range(16)] + [f"v{R.g[t] + i}" for i in range(16)]
    regs += [f"v{R.bias + i}" for i in range(16)] + [f"v{R.tmp + i}" for i in range(16)]
    regs += [f"a{R.F[0] + i}" for i in range(16)] + [f"a{R.F[1] + i}" for i in range(16)]
    for t in R.oa:
        regs += [f"a{R.oa[t][0] + i}" for i in range(16)] + [f"a{R.oa[t][1] + i}" for i in range(16)]
    return ", ".join(f'"{r}"' for r in regs)


def gen_two_tile(fill):
    R = Regs(True)
    L = []
    # GEMM2 accumulators start from residual + b2, handed over in the (still unused) ha / g registers
    init = [(R.oa["A"][0], R.ha["A"]), (R.oa["A"][1], R.g["A"]), (R.oa["B"][0], R.ha["B"]), (R.oa["B"][1], R.g["B"])]
    wr = [f"v_accvgpr_write_b32 a{a + i}, v{v + i}" for a, v in init for i in range(16)]
    # prologue 1: bias(0) -> R.bias, GEMM1(A, 0) (its target ha[A] is moved out first)
    L += [f"ds_read_b128 {R.vt(R.bias + 4 * q, 4)}, v{R.ab} offset:{32 * q}" for q in range(4)]
    L += wr[:16]
    L += half_step(R, "A", None, False, True, False, 0, 0, 0, None, False, (False, True, 0, 0), 4, extra=wr[16:])
    # prologue 2: GEMM1(B, 0) beside GELU(A, 0); loads bias(1); requests the loop's first fragments
    L += half_step(R, "B", "A", False, True, True, 0, 0, 0, 128, True, (True, True, 8192, 0), fill, delay_gaps=2)
    L += ["s_mov_b32 %[t], 0", "L_hid_%=:"]
    # loop body, T = %[t]: addresses are relative to aw1(T), aw2(T), ab(T)
    L += half_step(R, "A", "B", True, True, True, 8192, 0, 0, None, True, (True, True, 8192, 0), fill)
    # bias(T+2) for the next iteration is loaded once B's first GEMM1 MFMA has read bias(T+1)
    L += half_step(R, "B", "A", True, True, True, 8192, 0, 0, 256, True, (True, True, 16384, 4096), fill)
    L += [f"v_add_u32 v{R.aw1}, 0x2000, v{R.aw1}", f"v_add_u32 v{R.aw2}, 0x1000, v{R.aw2}",
          f"v_add_u32 v{R.ab}, 0x80, v{R.ab}",
          "s_add_u32 %[t], %[t], 1", "s_cmp_lt_u32 %[t], 6", "s_cbranch_scc1 L_hid_%="]
    # last full iteration (T = 6) without the out-of-range bias / fragment requests, then the epilogue (T = 7)
    L += half_step(R, "A", "B", True, True, True, 8192, 0, 0, None, True, (True, True, 8192, 0), fill)
    L += half_step(R, "B", "A", True, True, True, 8192, 0, 0, None, True, (True, False, 0, 4096), fill)
    L += half_step(R, "A", "B", True, False, True, 0, 4096, 0, None, True, (True, False, 0, 4096), fill)
    # results leave in the xh / xl registers: tile A's during the last (matrix-only) half step
    outs = [(R.xh["A"], R.oa["A"][0]), (R.xl["A"], R.oa["A"][1]), (R.xh["B"], R.oa["B"][0]), (R.xl["B"], R.oa["B"][1])]
    rd = [f"v_accvgpr_read_b32 v{v + i}, a{a + i}" for v, a in outs for i in range(16)]
    L += half_step(R, "B", None, True, False, False, 0, 4096, 0, None, True, None, 3, delay_gaps=1, extra=rd[:32])
    L += ["s_nop 15", "s_nop 1"]
    L += rd[32:]
    return R, L


def gen_one_tile(fill):
    """Software pipeline over T within one tile: ha / g ping-pong between the 'A' and 'B' register sets."""
    R = Regs(False, vtop=175, atop=63)     # 176 VGPRs + 64 AGPRs (+ compiler spills): two waves per SIMD
    L = []
    L += [f"ds_read_b128 {R.vt(R.bias + 4 * q, 4)}, v{R.ab} offset:{32 * q}" for q in range(4)]
    init = [(R.oa["A"][0], R.ha["A"]), (R.oa["A"][1], R.g["A"])]
    wr = [f"v_accvgpr_write_b32 a{a + i}, v{v + i}" for a, v in init for i in range(16)]
    L += wr[:16]
    # prologue 1: GEMM1(0) -> ha[A]
    L += half_step(R, "A", None, False, True, False, 0, 0, 0, 128, False, (False, True, 8192, 0), 2, extra=wr[16:])
    # prologue 2: GEMM1(1) -> ha[B] beside GELU(ha[A]) -> g[A]; bias(2)
    L += step_one(R, "B", "A", None, False, True, 8192, 0, 256, (True, True, 16384, 0), fill, delay_gaps=2)
    L += ["s_mov_b32 %[t], 0", "L_hid1_%=:"]
    # iteration j (T = 2j+1, 2j+2): relative to aw1 = W1 + (2j)*8192 etc.
    #   step T=2j+1: G2(2j) [g A] + G1(2j+2) -> ha[A]  ||  GELU(ha[B]) -> g[B]
    L += step_one(R, "A", "B", "A", True, True, 16384, 0, 384, (True, True, 24576, 4096), fill)
    #   step T=2j+2: G2(2j+1) [g B] + G1(2j+3) -> ha[B] || GELU(ha[A]) -> g[A]
    L += step_one(R, "B", "A", "B", True, True, 24576, 4096, 512, (True, True, 32768, 8192), fill)
    L += [f"v_add_u32 v{R.aw1}, 0x4000, v{R.aw1}", f"v_add_u32 v{R.aw2}, 0x2000, v{R.aw2}",
          f"v_add_u32 v{R.ab}, 0x100, v{R.ab}",
          "s_add_u32 %[t], %[t], 1", "s_cmp_lt_u32 %[t], 2", "s_cbranch_scc1 L_hid1_%="]
    # j = 2 peeled: T = 5 (G1(6) -> ha[A]), T = 6 (G1(7) -> ha[B], no further bias)
    L += step_one(R, "A", "B", "A", True, True, 16384, 0, 384, (True, True, 24576, 4096), fill)
    L += step_one(R, "B", "A", "B", True, True, 24576, 4096, None, (True, False, 0, 8192), fill)
    # T = 7: G2(6) [g A] || GELU(ha[B]) -> g[B];  then G2(7) [g B]
    L += step_one(R, "A", "B", "A", True, False, 0, 8192, None, (True, False, 0, 12288), fill)
    L += step_one(R, None, None, "B", True, False, 0, 12288, None, None, fill)
    L += ["s_nop 15", "s_nop 1"]
    outs = [(R.xh["A"], R.oa["A"][0]), (R.xl["A"], R.oa["A"][1])]
    L += [f"v_accvgpr_read_b32 v{v + i}, a{a + i}" for v, a in outs for i in range(16)]
    return R, L


def step_one(R, hn_set, gelu_set, g2_set, g2, g1, w1_off, w2_off, bias_next, next_frags, fill, delay_gaps=0):
    """One-tile variant of half_step: the GEMM1 target (hn_set), the GELU source (gelu_set) and the GEMM2
    operand set (g2_set) are independent register sets."""
    # half_step() reads ha[X] (GEMM1 target) and g[X] (GEMM2 operand) from the same tile key, so build a
    # view whose 'X' entry mixes the two sets
    class V:
        pass
    v = V()
    v.__dict__.update(R.__dict__)
    v.two_tile = False
    v.vt, v.at = R.vt, R.at
    v.ha = dict(R.ha)
    v.g = dict(R.g)
    v.ha["X"] = R.ha[hn_set] if hn_set else R.ha["A"]
    v.g["X"] = R.g[g2_set] if g2_set else R.g["A"]
    return half_step(v, "X", gelu_set, g2, g1, gelu_set is not None, w1_off, w2_off, 0, bias_next, True,
                     next_frags, fill, delay_gaps)


def main():
    global MODE
    fill = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    if len(sys.argv) > 2:
        MODE = sys.argv[2]
        if MODE.endswith("+nolds"):
            global NOLDS
            NOLDS = True
            MODE = MODE[:-len("+nolds")]
    print("// GENERATED by tools/gen_hidden_asm.py - do not edit.  Hand-scheduled FFN hidden loop (gfx950).")
    print(f"// fillers per MFMA gap: {fill}")
    variants = (("PF_HID2", gen_two_tile), ("PF_HID1", gen_one_tile))
    if len(sys.argv) > 3 and sys.argv[3] == "two":      # the product include only needs the two-tile stream
        variants = variants[:1]
    for nm, gen in variants:
        R, L = gen(fill)
        n_mfma = sum(1 for x in L if x.startswith("v_mfma"))
        print(f"// {nm}: {len(L)} instructions in the text, {n_mfma} MFMAs (static)")
        emit_block(nm + "_ASM", L, R)
        print(f"#define {nm}_CLOBBERS {clobbers(R)}")
        for t in R.xh:
            print(f"#define {nm}_XH_{t} \"{{v[{R.xh[t]}:{R.xh[t] + 15}]}}\"")
            print(f"#define {nm}_XL_{t} \"{{v[{R.xl[t]}:{R.xl[t] + 15}]}}\"")
            print(f"#define {nm}_OUT0_{t} \"={{v[{R.xh[t]}:{R.xh[t] + 15}]}}\"")
            print(f"#define {nm}_OUT1_{t} \"={{v[{R.xl[t]}:{R.xl[t] + 15}]}}\"")
            print(f"#define {nm}_INIT0_{t} \"+{{v[{R.ha[t]}:{R.ha[t] + 15}]}}\"")
            print(f"#define {nm}_INIT1_{t} \"+{{v[{R.g[t]}:{R.g[t] + 15}]}}\"")
        print(f"#define {nm}_AW1 \"+{{v{R.aw1}}}\"")
        print(f"#define {nm}_AW2 \"+{{v{R.aw2}}}\"")
        print(f"#define {nm}_AB \"+{{v{R.ab}}}\"")
        print(f"#define {nm}_C4 \"{{v{R.c4}}}\"")
        print(f"// lowest pinned VGPR v{R.vlo}, lowest pinned AGPR a{R.alo}")
        print()


if __name__ == "__main__":
    main()
