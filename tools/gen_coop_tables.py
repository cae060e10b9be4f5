#!/usr/bin/env python3
"""Generates plutus_halo2_verifier_gen_amd/csrc/coop_tables.h: the per-lane operand tables of the cooperative
(16 lanes per proof) Fp12 engine used by the pairing kernel, and checks them against the package's big-integer
Fp12 arithmetic before writing anything.

Model.  Fp12 = Fp2[w]/(w^6 - xi), element = 6 Fp2 coefficients c_k = c_k0 + c_k1 u.  Lane g < 12 of a group owns the
Fp value c_{k,part} with k = g >> 1, part = g & 1; lanes 12..15 are spare and compute the next line's per-proof
products.  An engine call computes, in every lane,   out = sum_t  X[xs_t] * Y[ys_t]  (mod p)
with ONE Montgomery reduction, X/Y being operand slots staged in LDS (28-bit limbs).  This file defines the slot map
and the (xs, ys) lists of the two operations:
   MUL   c = a * b          (12 terms)   slots A, NA (= -a_k1), B, XB (= xi * b)
   LINE  f = f * (c + b w^2 + yP w^3)   (6 terms)   sparse Miller-loop line, c shared per step, b = (-lambda) xP
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plutus_halo2_verifier_gen_amd import bls12_381 as bls  # noqa: E402

P = bls.P

# ---- slot map (group-local slots < 64; wave-shared constant slots >= 64)
SLOT_A = 0        # 12: a_{k,part} at A + 2k + part
SLOT_NA = 12      # 6 : -a_{k,1}
SLOT_B = 18       # 12
SLOT_XB = 30      # 12: (xi * b)_{k,part}
SLOT_T1 = 42      # 4 : loop-1 line products  b0, b1, (xi b)0, (xi b)1   with b = (-lambda) * xP1
SLOT_T2 = 46      # 4 : same for loop 2
SLOT_PX1, SLOT_PY1, SLOT_PX2, SLOT_PY2 = 50, 51, 52, 53
SLOT_ZERO = 54
SLOT_XA = 56      # 6 : (xi * a)_{k,part} for k = 3, 4, 5 (the wrapped squares of SQR) at XA + 2 (k - 3) + part
N_GROUP_SLOTS = 62
SH = 64           # shared slots: per line 8 constants  [nl0, nl1, nxl0, nxl1, c0, c1, xc0, xc1]
SLOT_LN1 = SH + 0
SLOT_LN2 = SH + 8
SLOT_C23P = SH + 16   # +2/3 and -2/3 (wave-shared constants of the cyclotomic squaring)
SLOT_C23N = SH + 17
N_SHARED_SLOTS = 18
C23 = 2 * pow(3, -1, P) % P
LN_NL0, LN_NL1, LN_NXL0, LN_NXL1, LN_C0, LN_C1, LN_XC0, LN_XC1 = range(8)

N_MUL_TERMS = 12
N_LINE_TERMS = 6
N_CSQR_TERMS = 4
N_SQR_TERMS = 8
# general squaring (the Miller loop's f^2) stages D = 2a in the B area and XD = xi * D in the XB area
SLOT_XD = SLOT_XB
# cyclotomic squaring reuses the B area for the doubled operand D = 2*g (12 slots; products with -2*g_k1 are taken as
# (-g_k1) * (2*g)), the XB area for the sums / differences S_k = g_k0 + g_k1, M_k = g_k0 - g_k1 (S_k at SM + 2k, M_k at
# SM + 2k + 1) and one slot of the XA area for ND2 = -2 g_21 (the one doubled NEGATIVE the xi-odd coefficient needs)
SLOT_D = SLOT_B
SLOT_SM = SLOT_XB
SLOT_ND2 = SLOT_XA


def mul_table():
    tab = []
    for g in range(16):
        terms = []
        if g < 12:
            k, part = g >> 1, g & 1
            for i in range(6):
                j = (k - i) % 6
                wrapped = i > k
                bj = SLOT_XB if wrapped else SLOT_B
                if part == 0:
                    terms.append((SLOT_A + 2 * i, bj + 2 * j))          # a_i0 * b_j0
                    terms.append((SLOT_NA + i, bj + 2 * j + 1))         # (-a_i1) * b_j1
                else:
                    terms.append((SLOT_A + 2 * i, bj + 2 * j + 1))      # a_i0 * b_j1
                    terms.append((SLOT_A + 2 * i + 1, bj + 2 * j))      # a_i1 * b_j0
        while len(terms) < N_MUL_TERMS:
            terms.append((SLOT_ZERO, SLOT_ZERO))
        tab.append(terms)
    return tab


def line_table(loop):
    """loop 1: constants LN1, products T1, point PY1 ; spare lanes compute the loop-2 products T2 (from LN2, PX2).
    loop 2: constants LN2, products T2, point PY2 ; spare lanes compute the NEXT loop-1 products T1 (from LN1, PX1)."""
    ln = SLOT_LN1 if loop == 1 else SLOT_LN2
    tt = SLOT_T1 if loop == 1 else SLOT_T2
    py = SLOT_PY1 if loop == 1 else SLOT_PY2
    oln = SLOT_LN2 if loop == 1 else SLOT_LN1
    opx = SLOT_PX2 if loop == 1 else SLOT_PX1
    tab = []
    for g in range(16):
        terms = []
        if g < 12:
            k, part = g >> 1, g & 1
            # j = 0: b_0 = c (never wrapped: i = k)
            i = k
            if part == 0:
                terms += [(SLOT_A + 2 * i, ln + LN_C0), (SLOT_NA + i, ln + LN_C1)]
            else:
                terms += [(SLOT_A + 2 * i, ln + LN_C1), (SLOT_A + 2 * i + 1, ln + LN_C0)]
            # j = 2: b_2 = b = (-lambda) xP  (products T: b0, b1, xb0, xb1)
            i = (k - 2) % 6
            wrapped = i > k
            b0, b1 = (tt + 2, tt + 3) if wrapped else (tt + 0, tt + 1)
            if part == 0:
                terms += [(SLOT_A + 2 * i, b0), (SLOT_NA + i, b1)]
            else:
                terms += [(SLOT_A + 2 * i, b1), (SLOT_A + 2 * i + 1, b0)]
            # j = 3: b_3 = (yP, 0); xi * (yP, 0) = (yP, yP)
            i = (k - 3) % 6
            wrapped = i > k
            if not wrapped:
                terms += [(SLOT_A + 2 * i + part, py)]
            elif part == 0:
                terms += [(SLOT_A + 2 * i, py), (SLOT_NA + i, py)]
            else:
                terms += [(SLOT_A + 2 * i, py), (SLOT_A + 2 * i + 1, py)]
        else:
            # spare lane s computes one product of the other loop's next line: [nl0, nl1, nxl0, nxl1][s] * xP
            terms.append((oln + (g - 12), opx))
        assert len(terms) <= N_LINE_TERMS
        while len(terms) < N_LINE_TERMS:
            terms.append((SLOT_ZERO, SLOT_ZERO))
        tab.append(terms)
    return tab


def sqr_table():
    """c = a^2 for a general Fp12 element, 8 terms per coefficient instead of MUL's 12: the products a_i a_j and
    a_j a_i of c_k = sum_{i+j = k mod 6} a_i a_j (x xi when i + j >= 6) are taken once against the doubled operand
    D = 2a (XD = xi D when wrapped); the two squares a_i^2 (i = k/2 unwrapped, i = k/2 + 3 wrapped, k even) are
    a_i x a_i and a_i x (xi a_i) = a_i x XA_i, their imaginary part 2 a_i0 a_i1 again through D."""
    A = lambda k, part: SLOT_A + 2 * k + part
    NA = lambda k: SLOT_NA + k
    D = lambda k, part: SLOT_D + 2 * k + part
    XD = lambda k, part: SLOT_XD + 2 * k + part
    XA = lambda k, part: SLOT_XA + 2 * (k - 3) + part
    tab = []
    for g in range(16):
        terms = []
        if g < 12:
            k, part = g >> 1, g & 1
            for i in range(6):
                j = (k - i) % 6
                if i > j:
                    continue
                wrapped = i + j >= 6
                if i < j:
                    Y = XD if wrapped else D
                    if part == 0:
                        terms += [(A(i, 0), Y(j, 0)), (NA(i), Y(j, 1))]
                    else:
                        terms += [(A(i, 0), Y(j, 1)), (A(i, 1), Y(j, 0))]
                elif not wrapped:        # a_i^2
                    if part == 0:
                        terms += [(A(i, 0), A(i, 0)), (NA(i), A(i, 1))]
                    else:
                        terms += [(A(i, 0), D(i, 1))]
                else:                    # xi a_i^2 = a_i x XA_i
                    if part == 0:
                        terms += [(A(i, 0), XA(i, 0)), (NA(i), XA(i, 1))]
                    else:
                        terms += [(A(i, 0), XA(i, 1)), (A(i, 1), XA(i, 0))]
        assert len(terms) <= N_SQR_TERMS
        while len(terms) < N_SQR_TERMS:
            terms.append((SLOT_ZERO, SLOT_ZERO))
        tab.append(terms)
    return tab


def stage_sqr(a):
    s = {SLOT_ZERO: 0}
    for k in range(6):
        s[SLOT_A + 2 * k], s[SLOT_A + 2 * k + 1] = a[k]
        s[SLOT_NA + k] = (-a[k][1]) % P
        d = (2 * a[k][0] % P, 2 * a[k][1] % P)
        s[SLOT_D + 2 * k], s[SLOT_D + 2 * k + 1] = d
        s[SLOT_XD + 2 * k], s[SLOT_XD + 2 * k + 1] = bls.f2_mul(bls.XI, d)
        if k >= 3:
            s[SLOT_XA + 2 * (k - 3)], s[SLOT_XA + 2 * (k - 3) + 1] = bls.f2_mul(bls.XI, a[k])
    return s


def csqr_table():
    """Granger-Scott squaring in the cyclotomic subgroup, flat basis, pairs (g0,g3), (g1,g4), (g2,g5):
         h0 = 3(g0^2 + xi g3^2) - 2g0   h3 = 3(2 g0 g3) + 2g3
         h1 = 3 xi (2 g2 g5)    + 2g1   h4 = 3(g2^2 + xi g5^2) - 2g4
         h2 = 3(g1^2 + xi g4^2) - 2g2   h5 = 3(2 g1 g4) + 2g5
    With the staged sums and differences S_k = g_k0 + g_k1, M_k = g_k0 - g_k1 (Re(x^2) = S M: one product instead of
    two) and D = 2g every Fp coefficient is at most THREE products (xi = 1 + u: xi z = (z0 - z1, z0 + z1)):
         even, real:  S_a M_a + S_b M_b - g_b1 D_b0          imaginary:  g_a0 D_a1 + S_b M_b + g_b0 D_b1
         odd,  real:  g_a0 D_b0 - g_a1 D_b1                  imaginary:  g_a0 D_b1 + g_a1 D_b0
         xi-odd, real: D_a0 M_b + (-2 g_a1) S_b              imaginary:  D_a0 S_b + D_a1 M_b
    The engine computes Q'_k = Q_k -/+ (2/3) g_k (minus for even k; the last term of every lane, against the shared
    constants +-2/3) with its column accumulators tripled before the Montgomery reduction, so h_k = 3 Q'_k leaves the
    engine reduced and no additive post-processing is left to the lane: 4 terms per coefficient, 2 per lane."""
    A = lambda k, part: SLOT_A + 2 * k + part
    NA = lambda k: SLOT_NA + k
    D = lambda k, part: SLOT_D + 2 * k + part
    S = lambda k: SLOT_SM + 2 * k
    M = lambda k: SLOT_SM + 2 * k + 1
    kind = {0: ("even", 0, 3), 3: ("odd", 0, 3), 1: ("xi_odd", 2, 5), 4: ("even", 2, 5), 2: ("even", 1, 4), 5: ("odd", 1, 4)}
    tab = []
    for g in range(16):
        terms = []
        if g < 12:
            k, part = g >> 1, g & 1
            ty, a, b = kind[k]
            if ty == "even":    # a^2 + xi b^2
                if part == 0:
                    terms = [(S(a), M(a)), (S(b), M(b)), (NA(b), D(b, 0))]
                else:
                    terms = [(A(a, 0), D(a, 1)), (S(b), M(b)), (A(b, 0), D(b, 1))]
            elif ty == "odd":   # 2 a b
                if part == 0:
                    terms = [(A(a, 0), D(b, 0)), (NA(a), D(b, 1))]
                else:
                    terms = [(A(a, 0), D(b, 1)), (A(a, 1), D(b, 0))]
            else:               # 2 xi a b = (2a) (xi b), a = g2: the doubling rides on a (D, and ND2 = -2 g_21 = 2 NA_2)
                assert a == 2
                if part == 0:
                    terms = [(D(a, 0), M(b)), (SLOT_ND2, S(b))]
                else:
                    terms = [(D(a, 0), S(b)), (D(a, 1), M(b))]
        if g < 12:
            assert len(terms) <= N_CSQR_TERMS - 1
            while len(terms) < N_CSQR_TERMS - 1:
                terms.append((SLOT_ZERO, SLOT_ZERO))
            # the constant term LAST in every row: the one-lane-per-coefficient engines leave it out and form -/+ 2 g themselves
            terms.append((A(g >> 1, g & 1), SLOT_C23N if (g >> 1) % 2 == 0 else SLOT_C23P))
        while len(terms) < N_CSQR_TERMS:
            terms.append((SLOT_ZERO, SLOT_ZERO))
        tab.append(terms)
    return tab


def stage_csqr(g):
    s = {SLOT_ZERO: 0, SLOT_C23P: C23, SLOT_C23N: (-C23) % P}
    for k in range(6):
        s[SLOT_A + 2 * k], s[SLOT_A + 2 * k + 1] = g[k]
        s[SLOT_NA + k] = (-g[k][1]) % P
        s[SLOT_D + 2 * k], s[SLOT_D + 2 * k + 1] = 2 * g[k][0] % P, 2 * g[k][1] % P
        s[SLOT_SM + 2 * k], s[SLOT_SM + 2 * k + 1] = (g[k][0] + g[k][1]) % P, (g[k][0] - g[k][1]) % P
    s[SLOT_ND2] = (-2 * g[2][1]) % P
    return s


def finish_csqr(q, g):
    out = []
    for lane in range(12):
        k, part = lane >> 1, lane & 1
        out.append(3 * q[lane] % P)
    return out


# ----------------------------------------------------------------------------- big-integer simulation
def stage_mul(a, b):
    """a, b: flat Fp12 (6 Fp2).  Returns slot dict."""
    s = {SLOT_ZERO: 0}
    for k in range(6):
        s[SLOT_A + 2 * k], s[SLOT_A + 2 * k + 1] = a[k]
        s[SLOT_NA + k] = (-a[k][1]) % P
        s[SLOT_B + 2 * k], s[SLOT_B + 2 * k + 1] = b[k]
        xb = bls.f2_mul(bls.XI, b[k])
        s[SLOT_XB + 2 * k], s[SLOT_XB + 2 * k + 1] = xb
    return s


def run(tab, slots):
    out = []
    for g in range(16):
        acc = 0
        for xs, ys in tab[g]:
            acc += slots[xs] * slots[ys]
        out.append(acc % P)
    return out


def to_flat(lanes):
    return [(lanes[2 * k], lanes[2 * k + 1]) for k in range(6)]


def line_consts(lam, c):
    nl = bls.f2_neg(lam)
    nxl = bls.f2_mul(bls.XI, nl)
    xc = bls.f2_mul(bls.XI, c)
    return [nl[0], nl[1], nxl[0], nxl[1], c[0], c[1], xc[0], xc[1]]


def self_check():
    rng = random.Random(1)
    rf2 = lambda: (rng.randrange(P), rng.randrange(P))
    mt = mul_table()
    for _ in range(10):
        a = [rf2() for _ in range(6)]
        b = [rf2() for _ in range(6)]
        assert to_flat(run(mt, stage_mul(a, b))) == bls.f12_mul(a, b)
    for loop in (1, 2):
        lt = line_table(loop)
        for _ in range(10):
            f = [rf2() for _ in range(6)]
            lam, c, lam_o = rf2(), rf2(), rf2()
            xp, yp, xo = rng.randrange(P), rng.randrange(P), rng.randrange(P)
            s = stage_mul(f, [bls.F2_ZERO] * 6)
            ln = SLOT_LN1 if loop == 1 else SLOT_LN2
            oln = SLOT_LN2 if loop == 1 else SLOT_LN1
            tt = SLOT_T1 if loop == 1 else SLOT_T2
            for q, v in enumerate(line_consts(lam, c)):
                s[ln + q] = v
            for q, v in enumerate(line_consts(lam_o, (0, 0))):
                s[oln + q] = v
            for q in range(4):
                s[tt + q] = s[ln + q] * xp % P
            s[SLOT_PY1 if loop == 1 else SLOT_PY2] = yp
            s[SLOT_PX2 if loop == 1 else SLOT_PX1] = xo
            out = run(lt, s)
            line = [c, bls.F2_ZERO, bls.f2_scale(bls.f2_neg(lam), xp), (yp, 0), bls.F2_ZERO, bls.F2_ZERO]
            assert to_flat(out) == bls.f12_mul(f, line)
            # spare lanes: products for the other loop's next line
            exp = [v * xo % P for v in line_consts(lam_o, (0, 0))[:4]]
            assert out[12:16] == exp
    st = sqr_table()
    for _ in range(10):
        a = [rf2() for _ in range(6)]
        assert to_flat(run(st, stage_sqr(a))) == bls.f12_sqr(a) == bls.f12_mul(a, a)
    # cyclotomic squaring on an element of the cyclotomic subgroup (easy part of a Miller value)
    f = bls.miller_loop(bls.g1_mul(bls.G1_GEN, 777), bls.g2_mul(bls.G2_GEN, 3))
    t = bls.f12_mul(bls.f12_conj(f), bls.f12_inv(f))
    t = bls.f12_mul(bls.f12_frob(bls.f12_frob(t)), t)
    ct = csqr_table()
    for _ in range(4):
        q = run(ct, stage_csqr(t))
        assert to_flat(finish_csqr(q, t)) == bls.f12_sqr(t)
        t = bls.f12_mul(bls.f12_sqr(t), t)
    return True


def emit():
    assert self_check()
    o = ["// GENERATED by tools/gen_coop_tables.py (tables verified against big-integer Fp12 arithmetic) - do not edit.",
         "#pragma once", "#include <stdint.h>"]
    for name in ("SLOT_A", "SLOT_NA", "SLOT_B", "SLOT_XB", "SLOT_T1", "SLOT_T2", "SLOT_PX1", "SLOT_PY1", "SLOT_PX2",
                 "SLOT_PY2", "SLOT_ZERO", "N_GROUP_SLOTS", "SLOT_LN1", "SLOT_LN2", "N_SHARED_SLOTS", "N_MUL_TERMS",
                 "N_LINE_TERMS", "N_CSQR_TERMS", "SLOT_D", "SLOT_SM", "SLOT_ND2", "SLOT_C23P", "SLOT_C23N", "SLOT_XA", "SLOT_XD",
                 "N_SQR_TERMS"):
        o.append("#define COOP_%s %d" % (name, globals()[name]))
    o.append("#define COOP_SHARED_BASE %d" % SH)

    def arr(name, tab, n):
        rows = []
        for g in range(16):
            rows.append("{" + ", ".join("%d, %d" % t for t in tab[g]) + "}")
        o.append("__device__ alignas(16) static constexpr uint8_t %s[16][%d] = {\n    %s};" % (name, 2 * n, ",\n    ".join(rows)))

    arr("COOP_TAB_MUL", mul_table(), N_MUL_TERMS)
    arr("COOP_TAB_LINE1", line_table(1), N_LINE_TERMS)
    arr("COOP_TAB_LINE2", line_table(2), N_LINE_TERMS)
    arr("COOP_TAB_CSQR", csqr_table(), N_CSQR_TERMS)
    arr("COOP_TAB_SQR", sqr_table(), N_SQR_TERMS)
    path = os.path.join(ROOT, "plutus_halo2_verifier_gen_amd", "csrc", "coop_tables.h")
    with open(path, "w") as f:
        f.write("\n".join(o) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    emit()
