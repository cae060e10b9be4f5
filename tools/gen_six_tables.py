#!/usr/bin/env python3
"""Generates plutus_halo2_verifier_gen_amd/csrc/six_tables.h: the operand tables of the SIX-LANES-PER-PROOF Fp12 engine
of the pairing kernel (h2v_pairing_six.hpp: ten proofs per wave), after checking them twice:
  * at VALUE level against the package's big-integer Fp12 arithmetic (every table, random operands), and
  * at LIMB level against a model of the device code (28-bit limbs, 64-bit column accumulators that wrap, the signed /
    unsigned Montgomery reductions) on operands at their declared bounds - the headroom argument of the header, executed.

Model.  Fp12 = Fp2[w]/(w^6 - xi), element = 6 Fp2 coefficients; lane k < 6 of a group owns coefficient k WHOLE (re, im).
A Karatsuba term is an Fp2 product x * y computed as three Fp products into three sets of column accumulators,
      U += x0 y0        V += x1 y1        W += (x0 + x1)(y0 + y1),
and an engine call (MUL: 6 terms, SQR: 4, LINE: 3) ends with   re = U - V  (signed columns),  im = W - U - V  (column-wise
equal to sum(x0 y1 + x1 y0) because the engine forms the sums LIMB-WISE in registers, so unsigned), one Montgomery reduction each: 3 NT + 2 products
of 196 multiply-adds where the one-coefficient-per-lane engine spends 2 (2 NT + 1) on the same Fp2 coefficient.
Wrapped terms (x xi) take the xi on the A side: XA = xi a = (a0 - a1, a0 + a1).
The cyclotomic squaring is five products into U, U, V, W, W (operands S = re + im - formed by the engine -, M = re - im,
NA = -im, D = 2a): re = U - V, im = W + V; the lane forms h = 3 (reduced) -/+ 2 g itself and folds it below 2p by a quotient
estimate from the top limb (the +-2/3 constant products of the other engines cost two products more).
A group's operand slots are 56 bytes (14 limbs, no padding) and 38 in number: 2.1 KB per proof, 23 KB per wave of ten.
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plutus_halo2_verifier_gen_amd import bls12_381 as bls  # noqa: E402

P = bls.P
R = 1 << 392
MASK = (1 << 28) - 1
M64 = (1 << 64) - 1
N0 = (-pow(P, -1, 1 << 28)) % (1 << 28)
P_L = [(P >> (28 * i)) & MASK for i in range(14)]

# ---- slot map: group-local slots < 64 (38 used, 56 bytes each), wave-shared >= 64
A0 = lambda k: 2 * k
A1 = lambda k: 2 * k + 1
XA0 = lambda k: 12 + 2 * (k - 1)        # k = 1..5 (MUL); the Miller loop stages k = 3..5 only
XA1 = lambda k: 13 + 2 * (k - 1)
B0 = lambda k: 22 + 2 * k               # MUL: b ; SQR / CSQR: D = 2a ; between the squaring and the lines of a Miller round:
B1 = lambda k: 23 + 2 * k               # the line products T1 = (b0, b1) of loop 1 at 22, 23 and T2 at 24, 25
T1_B0, T1_B1, T2_B0, T2_B1 = 22, 23, 24, 25
# cyclotomic squaring: NA, ND2 in the XA area, M = re - im behind them and in the P area; S = re + im is formed by the engine (A0 + A1)
C_NA = lambda k: 12 + k
C_ND2 = 18
C_M = lambda k: 19 + k if k < 3 else 31 + k     # 19..21 behind ND2, 34..36 over the points
PX1, PY1, PX2, PY2 = 34, 35, 36, 37
N_GROUP_SLOTS = 38
SH = 64
LN1, LN2 = SH, SH + 8                   # per line 8 slots [nl0, nl1, nxl0, nxl1, c0, c1, xc0, xc1] (the plan's record)
LN_NL0, LN_NL1, LN_C0, LN_C1 = 0, 1, 4, 5
C23P, C23N, ZERO = SH + 16, SH + 17, SH + 18
N_SHARED_SLOTS = 19
C23 = 2 * pow(3, -1, P) % P
N_MUL, N_SQR, N_LINE, N_CSQR = 6, 4, 3, 5
ZT = (ZERO, ZERO, ZERO, ZERO)


def a_side(i, wrapped):
    return (XA0(i), XA1(i)) if wrapped else (A0(i), A1(i))


def kterm(xs, ys):
    return (xs[0], ys[0], xs[1], ys[1])


def mul_table():
    tab = []
    for k in range(6):
        terms = []
        for i in range(6):
            j = (k - i) % 6
            terms.append(kterm(a_side(i, i > k), (B0(j), B1(j))))
        tab.append(terms)
    return tab


def sqr_table():
    """c_k = sum over unordered pairs {i, j}, i + j = k mod 6, of a_i a_j (x 2 when i != j, x xi when i + j >= 6).
    i < j unwrapped: a_i x D_j ; wrapped: D_i x XA_j (j in {4, 5}) ; squares: a_i x a_i, wrapped a_i x XA_i (i >= 3)."""
    D = lambda k: (B0(k), B1(k))
    tab = []
    for k in range(6):
        terms = []
        for i in range(6):
            for j in range(i, 6):
                if (i + j) % 6 != k:
                    continue
                wrapped = i + j >= 6
                if i < j:
                    terms.append(kterm(D(i), a_side(j, True)) if wrapped else kterm(a_side(i, False), D(j)))
                else:
                    terms.append(kterm(a_side(i, False), a_side(i, wrapped)))
        assert len(terms) <= N_SQR
        while len(terms) < N_SQR:
            terms.append(ZT)
        tab.append(terms)
    return tab


def line_table(loop):
    ln = LN1 if loop == 1 else LN2
    tb = (T1_B0, T1_B1) if loop == 1 else (T2_B0, T2_B1)
    py = PY1 if loop == 1 else PY2
    tab = []
    for k in range(6):
        i2, i3 = (k - 2) % 6, (k - 3) % 6
        tab.append([kterm(a_side(k, False), (ln + LN_C0, ln + LN_C1)),
                    kterm(a_side(i2, i2 > k), tb),
                    kterm(a_side(i3, i3 > k), (py, ZERO))])
    return tab


def csqr_table():
    """Per lane FIVE products (x + x2) * y into the accumulators U, U, V, W, W; the engine ends with re = U - V (signed columns),
    im = W + V, one reduction each (formulas: gen_coop_tables.csqr_table; x2 = ZERO except for S_k = re_k + im_k, which the engine
    forms from A0(k), A1(k)).  Granger-Scott with the pair (a, b) of the lane's kind:
       even    a^2 + xi b^2:  re = S_a M_a + S_b M_b - 2 b0 b1     im = 2 a0 a1 + S_b M_b + 2 b0 b1     (V = b0 D_b1 is SHARED, and so
                                                                                                         is the product S_b M_b: U and W)
       odd     2 a b:         re = a0 D_b0 + (-a1) D_b1            im = a0 D_b1 + a1 D_b0               (V = 0)
       xi-odd  2 xi a b:      re = D_a0 M_b + (-2 a1) S_b          im = S_b D_a0 + D_a1 M_b             (V = 0)
    The lane then forms h = 3 (reduced) -/+ 2 g_k itself and folds it below 2p (h2v_pairing_six.hpp: six_csqr)."""
    kind = {0: ("even", 0, 3), 3: ("odd", 0, 3), 1: ("xi_odd", 2, 5), 4: ("even", 2, 5), 2: ("even", 1, 4), 5: ("odd", 1, 4)}
    Z = (ZERO, ZERO, ZERO)
    one = lambda x, y: (x, ZERO, y)
    SxM = lambda k: (A0(k), A1(k), C_M(k))
    tab = []
    for k in range(6):
        ty, a, b = kind[k]
        if ty == "even":
            sets = [SxM(a), SxM(b), one(A0(b), B1(b)), one(A0(a), B1(a)), SxM(b)]
        elif ty == "odd":
            sets = [one(A0(a), B0(b)), one(C_NA(a), B1(b)), Z, one(A0(a), B1(b)), one(A1(a), B0(b))]
        else:
            assert a == 2
            sets = [one(B0(a), C_M(b)), (A0(b), A1(b), C_ND2), Z, (A0(b), A1(b), B0(a)), one(B1(a), C_M(b))]
        tab.append(sets)
    return tab


# ----------------------------------------------------------------------------- staging (value, limb bound) models
class Slots(dict):
    """slot -> list of 14 limbs (limbs may exceed 28 bits where the device stores them uncarried)"""

    def __init__(self):
        super().__init__()
        self.lam = {}                   # slot -> bound on its limbs (the analytic headroom check uses these, not the data)

    def put(self, s, value):            # carried
        self[s] = [(value >> (28 * i)) & MASK for i in range(13)] + [value >> (28 * 13)]
        assert self[s][13] < (1 << 28)
        self.lam[s] = 1 << 28

    def put_sum(self, s, *src):         # limb-wise sum of staged slots (uncarried)
        self[s] = [sum(self[q][i] for q in src) for i in range(14)]
        self.lam[s] = sum(self.lam[q] for q in src)

    def put_scaled(self, s, src, f):    # limb-wise multiple (uncarried)
        self[s] = [f * v for v in self[src]]
        self.lam[s] = f * self.lam[src]

    def val(self, s):
        return sum(v << (28 * i) for i, v in enumerate(self[s]))


def stage_a(s, f, xa_from):
    """f: list of 6 (re, im) integer representatives (any multiple of p allowed, < 6p)"""
    for k in range(6):
        s.put(A0(k), f[k][0]); s.put(A1(k), f[k][1])
        if k >= xa_from:
            s.put(XA0(k), f[k][0] + 7 * P - f[k][1]); s.put(XA1(k), f[k][0] + f[k][1])


def stage_b(s, f):
    for k in range(6):
        s.put(B0(k), f[k][0]); s.put(B1(k), f[k][1])


def stage_d(s, f):
    for k in range(6):
        s.put(A0(k), f[k][0]); s.put(A1(k), f[k][1])
        s.put_scaled(B0(k), A0(k), 2); s.put_scaled(B1(k), A1(k), 2)


def stage_csqr(s, g):
    s.put(C23P, C23 * R % P); s.put(C23N, (-C23) % P * R % P)     # Montgomery forms of +-2/3
    for k in range(6):
        s.put(A0(k), g[k][0]); s.put(A1(k), g[k][1]); s.put(C_NA(k), 7 * P - g[k][1])
        s.put_scaled(B0(k), A0(k), 2); s.put_scaled(B1(k), A1(k), 2)
        s.put(C_M(k), g[k][0] + 7 * P - g[k][1])
    s.put_scaled(C_ND2, C_NA(2), 2)


# ----------------------------------------------------------------------------- the device engine, limb for limb
def mac(acc, x, y):
    for i in range(14):
        for j in range(14):
            acc[i + j] = (acc[i + j] + x[i] * y[j]) & M64
            assert x[i] < (1 << 32) and y[j] < (1 << 32)


def to_signed(v):
    return v - (1 << 64) if v >> 63 else v


def reduce_cols(acc, signed):
    """Montgomery reduction of 28 columns (mod 2^64 registers); signed: columns are two's complement, p is added at the
    end.  Returns the 14 result limbs (carried) - asserts that no column left its register."""
    a = [to_signed(v) if signed else v for v in acc]
    lo, hi = (-(1 << 63), 1 << 63) if signed else (0, 1 << 64)
    for k in range(14):
        m = ((a[k] & 0xffffffff) * N0) & MASK
        for j in range(14):
            a[k + j] += m * P_L[j]
            assert lo <= a[k + j] < hi, "column overflow in the reduction"
        assert a[k] & MASK == 0
        a[k + 1] += a[k] >> 28
        assert lo <= a[k + 1] < hi
    out, carry = [], 0
    for k in range(13):
        carry += a[14 + k] + (P_L[k] if signed else 0)
        out.append(carry & MASK)
        carry >>= 28
    top = carry + a[27] + (P_L[13] if signed else 0)
    assert 0 <= top < (1 << 32), "negative or oversized result"
    return out + [top]


def limbs_val(l):
    return sum(v << (28 * i) for i, v in enumerate(l))


RED = 15 << 56      # what the reduction adds to a column at most: 14 products m p_j and a carry


def kara_engine(terms, s):
    # analytic headroom from the limb bounds of the staged slots (whatever the data)
    col = lambda x, y: 14 * s.lam[x] * s.lam[y]
    assert sum(col(t[0], t[1]) for t in terms) + RED < (1 << 63) and sum(col(t[2], t[3]) for t in terms) < (1 << 63), "re columns"
    assert sum(col(t[0], t[3]) + col(t[2], t[1]) for t in terms) + RED < (1 << 64), "im columns"
    U, V, W = [0] * 28, [0] * 28, [0] * 28
    for (x0, y0, x1, y1) in terms:
        mac(U, s[x0], s[y0]); mac(V, s[x1], s[y1])
        mac(W, [a + b for a, b in zip(s[x0], s[x1])], [a + b for a, b in zip(s[y0], s[y1])])     # the sums, limb-wise, in registers
    for acc in (U, V):
        assert max(acc) < (1 << 63), "U / V must not wrap (signed difference)"
    im = [(W[i] - U[i] - V[i]) & M64 for i in range(28)]
    re = [(U[i] - V[i]) & M64 for i in range(28)]
    # the column-wise identity: im columns are the true (non-negative) cross sums
    chk = [0] * 28
    for (x0, y0, x1, y1) in terms:
        for i in range(14):
            for j in range(14):
                chk[i + j] += s[x0][i] * s[y1][j] + s[x1][i] * s[y0][j]
    assert chk == im and max(chk) < (1 << 64)
    return limbs_val(reduce_cols(re, True)), limbs_val(reduce_cols(im, False))


FOLD_M = (1 << 32) // ((P >> 364) + 1)
BIAS_13_2 = None


def bias_13_2():
    """13 p written with every limb below the top >= 2 * 2^28 (tools/gen_device_consts.py: bias(13, 2))"""
    c = [(13 * P >> (28 * i)) & MASK for i in range(13)] + [13 * P >> 364]
    sp = 3
    out = [c[0] + (sp << 28)] + [c[i] + (sp << 28) - sp for i in range(1, 13)] + [c[13] - sp]
    assert sum(x << (28 * i) for i, x in enumerate(out)) == 13 * P
    return out


def carry_limbs(l):
    out, c = [], 0
    for i in range(13):
        t = l[i] + c
        assert t < (1 << 32)
        out.append(t & MASK)
        c = t >> 28
    assert l[13] + c < (1 << 32)
    return out + [l[13] + c]


def fold(l):
    """the device's f28_fold: carried limbs of a value below 32 p -> the same residue below 2p (and a hair), carried"""
    assert all(v < (1 << 28) for v in l[:13]) and l[13] < (1 << 22)
    q = (l[13] * FOLD_M) >> 32
    out, t = [], 0
    for i in range(14):
        t += l[i] - q * P_L[i]
        assert -(1 << 63) <= t < (1 << 63)
        out.append(t & MASK if i < 13 else t)
        t >>= 28
    assert t == 0 and 0 <= out[13] < (1 << 28)
    return out


def csqr_engine(sets, s, g, k):
    """sets: the lane's five products; g = (re, im) limbs of the lane's own coefficient as staged in A0(k), A1(k)"""
    lam3 = lambda t: 14 * (s.lam[t[0]] + s.lam[t[1]]) * s.lam[t[2]]
    assert lam3(sets[0]) + lam3(sets[1]) + RED < (1 << 63) and lam3(sets[2]) < (1 << 63), "re columns"
    assert lam3(sets[2]) + lam3(sets[3]) + lam3(sets[4]) + RED < (1 << 64), "im columns"
    U, V, W = [0] * 28, [0] * 28, [0] * 28
    for acc, (x, x2, y) in zip((U, U, V, W, W), sets):
        mac(acc, [a + b for a, b in zip(s[x], s[x2])], s[y])
    assert max(U) < (1 << 63) and max(V) < (1 << 63)
    re = [(U[i] - V[i]) & M64 for i in range(28)]
    im = [(W[i] + V[i]) & M64 for i in range(28)]
    r = [reduce_cols(re, True), reduce_cols(im, False)]
    bias = bias_13_2()
    out = []
    for part in range(2):
        two_g = [2 * v for v in s[(A0 if part == 0 else A1)(k)]]
        h = [3 * r[part][i] + (bias[i] - two_g[i] if k % 2 == 0 else two_g[i]) for i in range(14)]
        assert all(0 <= v < (1 << 32) for v in h)
        f = fold(carry_limbs(h))
        assert limbs_val(f) < 2 * P + (P >> 10)
        out.append(limbs_val(f))
    return out[0], out[1]


def prod_engine(s, x, y):
    acc = [0] * 28
    mac(acc, s[x], s[y])
    return limbs_val(reduce_cols(acc, False))


# ----------------------------------------------------------------------------- checks
RINV = pow(R, -1, P)


def mont(f):        # value -> Montgomery representative
    return [(a * R % P, b * R % P) for a, b in f]


def unmont(f):
    return [(a * RINV % P, b * RINV % P) for a, b in f]


def spread(rng, f, vmax):
    """random representatives: value + t p below vmax p"""
    return [tuple(c + rng.randrange(vmax) * P for c in pair) for pair in f]


def self_check():
    rng = random.Random(7)
    rf2 = lambda: (rng.randrange(P), rng.randrange(P))
    mt, st, ct = mul_table(), sqr_table(), csqr_table()
    for trial in range(6):
        vmax = 6 if trial else 1
        a = [rf2() for _ in range(6)]
        b = [rf2() for _ in range(6)]
        s = Slots(); s.put(ZERO, 0); s.lam[ZERO] = 0
        am, bm = spread(rng, mont(a), vmax), spread(rng, mont(b), vmax)
        stage_a(s, am, 1); stage_b(s, bm)
        got = [kara_engine(mt[k], s) for k in range(6)]
        assert all(v < 3 * P for pair in got for v in pair)
        assert unmont([(x % P, y % P) for x, y in got]) == bls.f12_mul(a, b)
        # squaring
        s = Slots(); s.put(ZERO, 0); s.lam[ZERO] = 0
        stage_d(s, am); stage_a(s, am, 3)
        got = [kara_engine(st[k], s) for k in range(6)]
        assert all(v < 3 * P for pair in got for v in pair)
        assert unmont([(x % P, y % P) for x, y in got]) == bls.f12_sqr(a)
        # lines
        for loop in (1, 2):
            lt = line_table(loop)
            lam, cc = rf2(), rf2()
            xp, yp = rng.randrange(P), rng.randrange(P)
            s = Slots(); s.put(ZERO, 0); s.lam[ZERO] = 0
            stage_a(s, am, 3)
            ln = LN1 if loop == 1 else LN2
            nl = bls.f2_neg(lam)
            s.put(ln + LN_NL0, nl[0] * R % P); s.put(ln + LN_NL1, nl[1] * R % P)
            s.put(ln + LN_C0, cc[0] * R % P); s.put(ln + LN_C1, cc[1] * R % P)
            px, py = (PX1, PY1) if loop == 1 else (PX2, PY2)
            s.put(px, xp * R % P); s.put(py, yp * R % P)
            tb = (T1_B0, T1_B1) if loop == 1 else (T2_B0, T2_B1)
            b0 = prod_engine(s, ln + LN_NL0, px); b1 = prod_engine(s, ln + LN_NL1, px)
            assert b0 < 2 * P and b1 < 2 * P
            s.put(tb[0], b0); s.put(tb[1], b1)
            got = [kara_engine(lt[k], s) for k in range(6)]
            assert all(v < 3 * P for pair in got for v in pair)
            line = [cc, bls.F2_ZERO, bls.f2_scale(nl, xp), (yp, 0), bls.F2_ZERO, bls.F2_ZERO]
            assert unmont([(x % P, y % P) for x, y in got]) == bls.f12_mul(a, line)
    # the fold on its own: every multiple of p up to 32 p and its neighbours, random values, the largest value it is handed
    for v in [k * P + d for k in range(32) for d in (-1, 0, 1) if k * P + d >= 0] + [rng.randrange(32 * P) for _ in range(2000)] + [32 * P - 1]:
        lim = [(v >> (28 * i)) & MASK for i in range(13)] + [v >> 364]
        f = limbs_val(fold(lim))
        assert f % P == v % P and f < 2 * P + (P >> 10)
    # cyclotomic squaring on an element of the cyclotomic subgroup
    f = bls.miller_loop(bls.g1_mul(bls.G1_GEN, 777), bls.g2_mul(bls.G2_GEN, 3))
    t = bls.f12_mul(bls.f12_conj(f), bls.f12_inv(f))
    t = bls.f12_mul(bls.f12_frob(bls.f12_frob(t)), t)
    for trial in range(4):
        s = Slots(); s.put(ZERO, 0); s.lam[ZERO] = 0
        stage_csqr(s, spread(rng, mont(t), 6))
        got = [csqr_engine(ct[k], s, None, k) for k in range(6)]
        assert all(v < 3 * P for pair in got for v in pair)
        assert unmont([(x % P, y % P) for x, y in got]) == bls.f12_sqr(t)
        t = bls.f12_mul(bls.f12_sqr(t), t)
    return True


def emit():
    assert self_check()
    o = ["// GENERATED by tools/gen_six_tables.py (tables checked against big-integer Fp12 arithmetic and a limb-level model of the",
         "// engine) - do not edit.", "#pragma once", "#include <stdint.h>"]
    for name, val in (("A", 0), ("XA", 12), ("T1", T1_B0), ("T2", T2_B0), ("B", 22), ("C_NA", 12), ("C_ND2", C_ND2), ("PX1", PX1),
                      ("PY1", PY1), ("PX2", PX2), ("PY2", PY2), ("ZERO", ZERO), ("LN1", LN1), ("LN2", LN2), ("C23P", C23P), ("C23N", C23N)):
        o.append("#define SIX_SLOT_%s %d" % (name, val))
    for name, val in (("N_GROUP_SLOTS", N_GROUP_SLOTS), ("N_SHARED_SLOTS", N_SHARED_SLOTS), ("SHARED_BASE", SH), ("N_MUL", N_MUL), ("N_SQR", N_SQR),
                      ("N_LINE", N_LINE), ("N_CSQR", N_CSQR)):
        o.append("#define SIX_%s %d" % (name, val))

    def arr(name, tab, width):
        rows = ["{" + ", ".join(str(v) for term in tab[k] for v in term) + "}" for k in range(6)]
        o.append("__device__ alignas(4) static constexpr uint8_t %s[6][%d] = {\n    %s};" % (name, width, ",\n    ".join(rows)))

    arr("SIX_TAB_MUL", mul_table(), 4 * N_MUL)
    arr("SIX_TAB_SQR", sqr_table(), 4 * N_SQR)
    arr("SIX_TAB_LINE1", line_table(1), 4 * N_LINE)
    arr("SIX_TAB_LINE2", line_table(2), 4 * N_LINE)
    arr("SIX_TAB_CSQR", [row + [(ZERO,)] for row in csqr_table()], 16)      # (15 slot bytes per lane, padded to 16)
    path = os.path.join(ROOT, "plutus_halo2_verifier_gen_amd", "csrc", "six_tables.h")
    with open(path, "w") as f:
        f.write("\n".join(o) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    emit()
