"""Host-side (plan-time) BLS12-381 big-integer arithmetic.

This module is NOT on the verification hot path.  It is the host logic the
plan compiler needs once per VerifyingKey (decompress / validate VK points,
precompute the G2 line table for the fixed pairing arguments, convert constants
to Montgomery limbs) and what the synthetic-workload generator (`synth.py`)
uses to forge accepting proofs with the test-SRS trapdoor.  The reference does
the same work on the host, in Rust, inside `extract_circuit`
(/root/reference/src/plutus_gen/extraction/data/circuit_types/instantiation_data.rs:65-137).

Curve facts (public BLS12-381 standard; constants as cited by the reference):
  p  : /root/reference/aiken-verifier/aiken_halo2/lib/bls_utils.ak:14-15
  r  : /root/reference/plinth-verifier/plutus-halo2/src/Plutus/Crypto/BlsTypes.hs:97
  E  : y^2 = x^3 + 4          (CompressUncompress.hs:98)
  E' : y^2 = x^3 + 4(1+u)     (M-type sextic twist over Fp2 = Fp[u]/(u^2+1))
  zcash compressed encoding flags: bls_utils.ak:17-28
"""
from __future__ import annotations

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
BLS_X = 0xD201000000010000  # |x|; the BLS parameter is -BLS_X
BLS_X_IS_NEG = True

G1_X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
G1_Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
G2_X = (
    0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
    0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E,
)
G2_Y = (
    0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
    0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE,
)

# delta = 7^(2^32) mod r  (/root/reference/aiken-verifier/templates/verification_h2.hbs:24)
DELTA = 0x08634D0AA021AAF843CAB354FABB0062F6502437C6A09C006C083479590189D7
# 2^256 mod r (/root/reference/aiken-verifier/aiken_halo2/lib/transcript.ak:99)
R_2_256 = 0x1824B159ACC5056F998C4FEFECBC4FF55884B7FA0003480200000001FFFFFFFE

MONT_BITS_FP = 392  # the gfx950 multiplier works on 14 x 28-bit limbs: R = 2^392
MONT_BITS_FR = 256


# ----------------------------------------------------------------------------- Fr / Fp helpers
def fr(x: int) -> int:
    return x % R


def fr_inv(x: int) -> int:
    x %= R
    if x == 0:
        raise ZeroDivisionError("Fr inverse of zero")
    return pow(x, R - 2, R)


def fp_inv(x: int) -> int:
    x %= P
    if x == 0:
        raise ZeroDivisionError("Fp inverse of zero")
    return pow(x, P - 2, P)


def fp_sqrt(a: int):
    """p = 3 mod 4: candidate a^((p+1)/4) (CompressUncompress.hs:98)."""
    a %= P
    c = pow(a, (P + 1) // 4, P)
    return c if c * c % P == a else None


# ----------------------------------------------------------------------------- Fp2 = Fp[u]/(u^2+1)
def f2(a, b=0):
    return (a % P, b % P)


F2_ZERO = (0, 0)
F2_ONE = (1, 0)
XI = (1, 1)  # 1 + u


def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_sqr(a):
    return f2_mul(a, a)


def f2_scale(a, k):
    return (a[0] * k % P, a[1] * k % P)


def f2_inv(a):
    n = fp_inv(a[0] * a[0] + a[1] * a[1])
    return (a[0] * n % P, (-a[1]) * n % P)


def f2_conj(a):
    return (a[0], (-a[1]) % P)


def f2_pow(a, e):
    res = F2_ONE
    base = a
    while e:
        if e & 1:
            res = f2_mul(res, base)
        base = f2_sqr(base)
        e >>= 1
    return res


def f2_sqrt(a):
    """Square root in Fp2 (p = 3 mod 4), or None."""
    if a == F2_ZERO:
        return F2_ZERO
    # Algorithm 9 of "Square root computation over even extension fields" (Adj, Rodriguez-Henriquez)
    a1 = f2_pow(a, (P - 3) // 4)
    alpha = f2_mul(f2_sqr(a1), a)
    a0 = f2_mul(f2_conj(alpha), alpha)  # alpha^(p+1)
    if a0 == ((P - 1) % P, 0):
        return None
    x0 = f2_mul(a1, a)
    if alpha == ((P - 1) % P, 0):
        res = f2_mul((0, 1), x0)
    else:
        b = f2_pow(f2_add(F2_ONE, alpha), (P - 1) // 2)
        res = f2_mul(b, x0)
    return res if f2_sqr(res) == a else None


# ----------------------------------------------------------------------------- Fp12 = Fp2[w]/(w^6 - xi), flat
# Element = list of 6 Fp2 coefficients of w^0..w^5.  Tower view used by the C/HIP code:
#   Fp6 = Fp2[v]/(v^3 - xi), Fp12 = Fp6[w]/(w^2 - v)  =>  (a0,a1,a2 | b0,b1,b2) == flat [a0,b0,a1,b1,a2,b2].
F12_ONE = [F2_ONE] + [F2_ZERO] * 5


def f12_mul(a, b):
    t = [F2_ZERO] * 11
    for i in range(6):
        if a[i] == F2_ZERO:
            continue
        for j in range(6):
            if b[j] == F2_ZERO:
                continue
            t[i + j] = f2_add(t[i + j], f2_mul(a[i], b[j]))
    out = list(t[:6])
    for k in range(6, 11):
        out[k - 6] = f2_add(out[k - 6], f2_mul(t[k], XI))
    return out


def f12_sqr(a):
    return f12_mul(a, a)


def f12_pow(a, e):
    res = F12_ONE
    base = a
    while e:
        if e & 1:
            res = f12_mul(res, base)
        base = f12_sqr(base)
        e >>= 1
    return res


def f12_conj(a):
    """a^(p^6): w -> -w."""
    return [a[0], f2_neg(a[1]), a[2], f2_neg(a[3]), a[4], f2_neg(a[5])]


_FROB_W = None


def _frob_consts():
    global _FROB_W
    if _FROB_W is None:
        # w^p = w * xi^((p-1)/6)
        g = f2_pow(XI, (P - 1) // 6)
        _FROB_W = [f2_pow(g, k) for k in range(6)]
    return _FROB_W


def f12_frob(a):
    """a^p."""
    c = _frob_consts()
    return [f2_mul(f2_conj(a[k]), c[k]) for k in range(6)]


def f12_inv(a):
    """Inverse via a^(p^6) trick: a * conj(a) lies in Fp6 = Fp2[v]; solve there with norms."""
    # N = a * conj(a) has only even coefficients (in Fp6 with v = w^2)
    n = f12_mul(a, f12_conj(a))
    c0, c1, c2 = n[0], n[2], n[4]
    # Fp6 inverse (v^3 = xi)
    t0 = f2_sub(f2_sqr(c0), f2_mul(XI, f2_mul(c1, c2)))
    t1 = f2_sub(f2_mul(XI, f2_sqr(c2)), f2_mul(c0, c1))
    t2 = f2_sub(f2_sqr(c1), f2_mul(c0, c2))
    d = f2_add(f2_mul(c0, t0), f2_mul(XI, f2_add(f2_mul(c2, t1), f2_mul(c1, t2))))
    di = f2_inv(d)
    i0, i1, i2 = f2_mul(t0, di), f2_mul(t1, di), f2_mul(t2, di)
    ninv = [i0, F2_ZERO, i1, F2_ZERO, i2, F2_ZERO]
    return f12_mul(f12_conj(a), ninv)


# ----------------------------------------------------------------------------- G1 (affine, None = infinity)
G1_GEN = (G1_X, G1_Y)


def g1_is_on_curve(pt) -> bool:
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - 4) % P == 0


def g1_neg(pt):
    if pt is None:
        return None
    return (pt[0], (-pt[1]) % P)


def g1_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * fp_inv(2 * y1) % P
    else:
        lam = (y2 - y1) * fp_inv(x2 - x1) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def _jac_double(X, Y, Z):
    if Y == 0 or Z == 0:
        return (1, 1, 0)
    A = X * X % P
    B = Y * Y % P
    C = B * B % P
    D = 2 * ((X + B) * (X + B) - A - C) % P
    E = 3 * A % P
    F = E * E % P
    X3 = (F - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def _jac_add_affine(X1, Y1, Z1, x2, y2):
    if Z1 == 0:
        return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % P
    U2 = x2 * Z1Z1 % P
    S2 = y2 * Z1 * Z1Z1 % P
    H = (U2 - X1) % P
    rr = (S2 - Y1) % P
    if H == 0:
        if rr == 0:
            return _jac_double(X1, Y1, Z1)
        return (1, 1, 0)
    HH = H * H % P
    HHH = H * HH % P
    V = X1 * HH % P
    X3 = (rr * rr - HHH - 2 * V) % P
    Y3 = (rr * (V - X3) - Y1 * HHH) % P
    Z3 = Z1 * H % P
    return (X3, Y3, Z3)


def g1_mul(pt, k: int):
    """Scalar multiplication (k reduced mod r is NOT applied: callers pass what they mean)."""
    if pt is None or k == 0:
        return None
    if k < 0:
        return g1_mul(g1_neg(pt), -k)
    X, Y, Z = 1, 1, 0
    x2, y2 = pt
    for bit in bin(k)[2:]:
        X, Y, Z = _jac_double(X, Y, Z)
        if bit == "1":
            X, Y, Z = _jac_add_affine(X, Y, Z, x2, y2)
    if Z == 0:
        return None
    zi = fp_inv(Z)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def g1_in_subgroup(pt) -> bool:
    return g1_mul(pt, R) is None


def g1_compress(pt) -> bytes:
    """zcash format: bit7 compressed, bit6 infinity, bit5 y lexicographically larger (bls_utils.ak:17-28)."""
    if pt is None:
        return bytes([0xC0]) + bytes(47)
    x, y = pt
    flags = 0x80 | (0x20 if y > (P - y) % P else 0)
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= flags
    return bytes(b)


def g1_decompress(b: bytes, check_subgroup: bool = True):
    """Returns affine point / None for infinity; raises ValueError on any malformed encoding."""
    if len(b) != 48:
        raise ValueError("G1: wrong length")
    flags = b[0] >> 5
    if not flags & 4:
        raise ValueError("G1: compression flag not set")
    x = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:], "big")
    if flags & 2:
        if x != 0 or flags & 1:
            raise ValueError("G1: bad infinity encoding")
        return None
    if x >= P:
        raise ValueError("G1: x not canonical")
    y = fp_sqrt(x * x * x + 4)
    if y is None:
        raise ValueError("G1: not on curve")
    if (y > (P - y) % P) != bool(flags & 1):
        y = (P - y) % P
    pt = (x, y)
    if check_subgroup and not g1_in_subgroup(pt):
        raise ValueError("G1: not in subgroup")
    return pt


# ----------------------------------------------------------------------------- G2 (affine over Fp2, None = infinity)
G2_GEN = (G2_X, G2_Y)
B2 = f2_scale(XI, 4)


def g2_is_on_curve(pt) -> bool:
    if pt is None:
        return True
    x, y = pt
    return f2_sub(f2_sqr(y), f2_add(f2_mul(f2_sqr(x), x), B2)) == F2_ZERO


def g2_neg(pt):
    return None if pt is None else (pt[0], f2_neg(pt[1]))


def g2_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if f2_add(y1, y2) == F2_ZERO:
            return None
        lam = f2_mul(f2_scale(f2_sqr(x1), 3), f2_inv(f2_scale(y1, 2)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_sqr(lam), x1), x2)
    y3 = f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1)
    return (x3, y3)


def g2_mul(pt, k: int):
    if pt is None or k == 0:
        return None
    if k < 0:
        return g2_mul(g2_neg(pt), -k)
    acc = None
    for bit in bin(k)[2:]:
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, pt)
    return acc


def g2_in_subgroup(pt) -> bool:
    return g2_mul(pt, R) is None


def _f2_lex_larger(y) -> bool:
    ny = f2_neg(y)
    # zcash: compare c1 first, then c0
    return (y[1], y[0]) > (ny[1], ny[0])


def g2_compress(pt) -> bytes:
    if pt is None:
        return bytes([0xC0]) + bytes(95)
    x, y = pt
    flags = 0x80 | (0x20 if _f2_lex_larger(y) else 0)
    b = bytearray(x[1].to_bytes(48, "big") + x[0].to_bytes(48, "big"))
    b[0] |= flags
    return bytes(b)


def g2_decompress(b: bytes, check_subgroup: bool = True):
    if len(b) != 96:
        raise ValueError("G2: wrong length")
    flags = b[0] >> 5
    if not flags & 4:
        raise ValueError("G2: compression flag not set")
    x1 = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:48], "big")
    x0 = int.from_bytes(b[48:], "big")
    if flags & 2:
        if x0 or x1 or flags & 1:
            raise ValueError("G2: bad infinity encoding")
        return None
    if x0 >= P or x1 >= P:
        raise ValueError("G2: x not canonical")
    x = (x0, x1)
    y = f2_sqrt(f2_add(f2_mul(f2_sqr(x), x), B2))
    if y is None:
        raise ValueError("G2: not on curve")
    if _f2_lex_larger(y) != bool(flags & 1):
        y = f2_neg(y)
    pt = (x, y)
    if check_subgroup and not g2_in_subgroup(pt):
        raise ValueError("G2: not in subgroup")
    return pt


# ----------------------------------------------------------------------------- pairing pieces
def miller_bits():
    """Bits of |x| below the leading one, MSB first."""
    return [int(c) for c in bin(BLS_X)[3:]]


def g2_line_table(q):
    """Per-step line coefficients (lambda', c = lambda'*x_T - y_T) for the optimal-ate Miller loop on a
    FIXED G2 argument q (affine, on the twist).  One entry per doubling step, plus one per set bit of |x|.

    With the untwist (x', y') -> (x'/w^2, y'/w^3) the line through T evaluated at P = (xP, yP) in G1, scaled
    by w^3 (an element of a proper subfield, killed by the final exponentiation), is
        l(P) = c  +  (-lambda' * xP) w^2  +  yP w^3 .
    """
    if q is None:
        raise ValueError("G2 argument of the pairing must not be infinity")
    table = []
    t = q
    for bit in miller_bits():
        x1, y1 = t
        lam = f2_mul(f2_scale(f2_sqr(x1), 3), f2_inv(f2_scale(y1, 2)))
        table.append((lam, f2_sub(f2_mul(lam, x1), y1)))
        t = g2_add(t, t)
        if bit:
            x1, y1 = t
            lam = f2_mul(f2_sub(q[1], y1), f2_inv(f2_sub(q[0], x1)))
            table.append((lam, f2_sub(f2_mul(lam, x1), y1)))
            t = g2_add(t, q)
    return table


def _line_eval(entry, p1):
    lam, c = entry
    xp, yp = p1
    return [c, F2_ZERO, f2_scale(f2_neg(lam), xp), (yp % P, 0), F2_ZERO, F2_ZERO]


def miller_loop(p1, q):
    """f_{|x|,Q}(P), conjugated because x < 0.  Infinity in either argument gives one."""
    if p1 is None or q is None:
        return list(F12_ONE)
    table = g2_line_table(q)
    f = list(F12_ONE)
    idx = 0
    for bit in miller_bits():
        f = f12_mul(f12_sqr(f), _line_eval(table[idx], p1))
        idx += 1
        if bit:
            f = f12_mul(f, _line_eval(table[idx], p1))
            idx += 1
    return f12_conj(f) if BLS_X_IS_NEG else f


def final_exponentiation(f):
    """Canonical f^((p^12-1)/r) by plain exponentiation (slow; host/test use only)."""
    # easy part first to shrink the exponent work: (p^6-1)(p^2+1)
    t = f12_mul(f12_conj(f), f12_inv(f))
    t = f12_mul(f12_frob(f12_frob(t)), t)
    return f12_pow(t, (P**4 - P**2 + 1) // R)


def pairing(p1, q):
    return final_exponentiation(miller_loop(p1, q))


def pairing_check_eq(a1, b1, a2, b2) -> bool:
    """e(a1, b1) == e(a2, b2), computed as FE(ML(a1,b1) * ML(-a2,b2)) == 1."""
    f = f12_mul(miller_loop(a1, b1), miller_loop(g1_neg(a2), b2))
    return final_exponentiation(f) == F12_ONE


# ----------------------------------------------------------------------------- Montgomery limbs
def to_mont_fp(x: int) -> int:
    return (x << MONT_BITS_FP) % P


def to_mont_fr(x: int) -> int:
    return (x << MONT_BITS_FR) % R


def fp_mont_bytes(x: int) -> bytes:
    """48-byte little-endian Montgomery form (12 x u32 storage limbs, R = 2^392)."""
    return to_mont_fp(x % P).to_bytes(48, "little")


def fr_mont_bytes(x: int) -> bytes:
    return to_mont_fr(x % R).to_bytes(32, "little")


def f2_mont_bytes(a) -> bytes:
    return fp_mont_bytes(a[0]) + fp_mont_bytes(a[1])


def fp_mont28_slot(x: int) -> bytes:
    """One 64-byte operand slot of the cooperative pairing engine: Montgomery form (R = 2^392) cut into 14 limbs of
    28 bits, one limb per little-endian dword, padded to 16 dwords."""
    m = to_mont_fp(x % P)
    out = b"".join(((m >> (28 * i)) & 0xFFFFFFF).to_bytes(4, "little") for i in range(14))
    return out + bytes(8)


def line_slots(lam, c) -> bytes:
    """The 8 shared constants of one Miller-loop line in engine order
    [-lam0, -lam1, (xi*-lam)0, (xi*-lam)1, c0, c1, (xi*c)0, (xi*c)1] (tools/gen_coop_tables.py: LN_*)."""
    nl = f2_neg(lam)
    nxl = f2_mul(XI, nl)
    xc = f2_mul(XI, c)
    return b"".join(fp_mont28_slot(v) for v in (nl[0], nl[1], nxl[0], nxl[1], c[0], c[1], xc[0], xc[1]))


# ----------------------------------------------------------------------------- GLV (model of the device code)
GLV_LAMBDA = BLS_X * BLS_X - 1
GLV_MU = (1 << 383) // GLV_LAMBDA


def glv_split(k: int):
    """k = k2*lambda + k1 with 0 <= k1 < lambda, exactly as csrc/h2v_curve.hpp: glv_split computes it
    (Barrett estimate floor(k*mu / 2^383), at most two corrections)."""
    q = (k * GLV_MU) >> 383
    rem = k - q * GLV_LAMBDA
    n_corr = 0
    while rem >= GLV_LAMBDA:
        rem -= GLV_LAMBDA
        q += 1
        n_corr += 1
    assert n_corr <= 2
    return rem, q
