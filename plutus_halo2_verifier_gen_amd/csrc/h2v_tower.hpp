// Extension tower Fp2 / Fp6 / Fp12 for the BLS12-381 pairing on gfx950.
//   Fp2 = Fp[u]/(u^2+1)          (reference: plinth-verifier/plutus-halo2/src/Plutus/Crypto/BlsTypes.hs:302-380)
//   Fp6 = Fp2[v]/(v^3 - xi), xi = 1+u ;  Fp12 = Fp6[w]/(w^2 - v)
// Karatsuba at every level (Fp2: 3 Fp mul, Fp6: 6 Fp2 mul, Fp12: 3 Fp6 mul), complex squaring for Fp2 and
// Fp12, and a dedicated sparse product for the Miller-loop line  c + b w^2 + a w^3  (a in Fp).
#pragma once
#include "h2v_field.hpp"

struct Fp2 { Fp c0, c1; };
struct Fp6 { Fp2 c0, c1, c2; };
struct Fp12 { Fp6 c0, c1; };

H2V_DI void fp2_set_zero(Fp2 &r) { fp_set_zero(r.c0); fp_set_zero(r.c1); }
H2V_DI void fp2_set_one(Fp2 &r) { fp_set_one(r.c0); fp_set_zero(r.c1); }
H2V_DI bool fp2_is_zero(const Fp2 &a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
H2V_DI bool fp2_eq(const Fp2 &a, const Fp2 &b) { return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1); }
H2V_DI void fp2_add(Fp2 &r, const Fp2 &a, const Fp2 &b) { fp_add(r.c0, a.c0, b.c0); fp_add(r.c1, a.c1, b.c1); }
H2V_DI void fp2_sub(Fp2 &r, const Fp2 &a, const Fp2 &b) { fp_sub(r.c0, a.c0, b.c0); fp_sub(r.c1, a.c1, b.c1); }
H2V_DI void fp2_neg(Fp2 &r, const Fp2 &a) { fp_neg(r.c0, a.c0); fp_neg(r.c1, a.c1); }
H2V_DI void fp2_dbl(Fp2 &r, const Fp2 &a) { fp_dbl(r.c0, a.c0); fp_dbl(r.c1, a.c1); }
H2V_DI void fp2_conj(Fp2 &r, const Fp2 &a) { r.c0 = a.c0; fp_neg(r.c1, a.c1); }
H2V_DN void fp2_mul(Fp2 &r, const Fp2 &a, const Fp2 &b) {
    Fp t0, t1, t2, t3;
    fp_mul(t0, a.c0, b.c0);
    fp_mul(t1, a.c1, b.c1);
    fp_add(t2, a.c0, a.c1);
    fp_add(t3, b.c0, b.c1);
    fp_mul(t2, t2, t3);
    fp_sub(t2, t2, t0);
    fp_sub(r.c1, t2, t1);
    fp_sub(r.c0, t0, t1);
}
H2V_DN void fp2_sqr(Fp2 &r, const Fp2 &a) {
    Fp t0, t1, t2;
    fp_add(t0, a.c0, a.c1);
    fp_sub(t1, a.c0, a.c1);
    fp_mul(t2, a.c0, a.c1);
    fp_mul(r.c0, t0, t1);
    fp_dbl(r.c1, t2);
}
H2V_DI void fp2_mul_fp(Fp2 &r, const Fp2 &a, const Fp &k) { fp_mul(r.c0, a.c0, k); fp_mul(r.c1, a.c1, k); }
// * xi = (1+u): (a0 - a1) + (a0 + a1) u
H2V_DI void fp2_mul_xi(Fp2 &r, const Fp2 &a) {
    Fp t0, t1;
    fp_sub(t0, a.c0, a.c1);
    fp_add(t1, a.c0, a.c1);
    r.c0 = t0;
    r.c1 = t1;
}
H2V_DN bool fp2_inv(Fp2 &r, const Fp2 &a) {
    Fp t0, t1;
    fp_sqr(t0, a.c0);
    fp_sqr(t1, a.c1);
    fp_add(t0, t0, t1);
    bool ok = fp_inv(t0, t0);
    fp_mul(r.c0, a.c0, t0);
    fp_mul(t1, a.c1, t0);
    fp_neg(r.c1, t1);
    return ok;
}

// ------------------------------------------------------------------ Fp6
H2V_DI void fp6_set_zero(Fp6 &r) { fp2_set_zero(r.c0); fp2_set_zero(r.c1); fp2_set_zero(r.c2); }
H2V_DI void fp6_add(Fp6 &r, const Fp6 &a, const Fp6 &b) { fp2_add(r.c0, a.c0, b.c0); fp2_add(r.c1, a.c1, b.c1); fp2_add(r.c2, a.c2, b.c2); }
H2V_DI void fp6_sub(Fp6 &r, const Fp6 &a, const Fp6 &b) { fp2_sub(r.c0, a.c0, b.c0); fp2_sub(r.c1, a.c1, b.c1); fp2_sub(r.c2, a.c2, b.c2); }
H2V_DI void fp6_neg(Fp6 &r, const Fp6 &a) { fp2_neg(r.c0, a.c0); fp2_neg(r.c1, a.c1); fp2_neg(r.c2, a.c2); }
H2V_DI bool fp6_eq(const Fp6 &a, const Fp6 &b) { return fp2_eq(a.c0, b.c0) && fp2_eq(a.c1, b.c1) && fp2_eq(a.c2, b.c2); }
// * v : (c0, c1, c2) -> (xi c2, c0, c1)
H2V_DI void fp6_mul_v(Fp6 &r, const Fp6 &a) {
    Fp2 t;
    fp2_mul_xi(t, a.c2);
    r.c2 = a.c1;
    r.c1 = a.c0;
    r.c0 = t;
}
// Karatsuba (6 Fp2 products)
H2V_DN void fp6_mul(Fp6 &r, const Fp6 &a, const Fp6 &b) {
    Fp2 v0, v1, v2, t0, t1, t2;
    fp2_mul(v0, a.c0, b.c0);
    fp2_mul(v1, a.c1, b.c1);
    fp2_mul(v2, a.c2, b.c2);
    Fp6 o;
    // c0 = v0 + xi((a1+a2)(b1+b2) - v1 - v2)
    fp2_add(t0, a.c1, a.c2); fp2_add(t1, b.c1, b.c2); fp2_mul(t2, t0, t1);
    fp2_sub(t2, t2, v1); fp2_sub(t2, t2, v2); fp2_mul_xi(t2, t2); fp2_add(o.c0, v0, t2);
    // c1 = (a0+a1)(b0+b1) - v0 - v1 + xi v2
    fp2_add(t0, a.c0, a.c1); fp2_add(t1, b.c0, b.c1); fp2_mul(t2, t0, t1);
    fp2_sub(t2, t2, v0); fp2_sub(t2, t2, v1); fp2_mul_xi(t0, v2); fp2_add(o.c1, t2, t0);
    // c2 = (a0+a2)(b0+b2) - v0 - v2 + v1
    fp2_add(t0, a.c0, a.c2); fp2_add(t1, b.c0, b.c2); fp2_mul(t2, t0, t1);
    fp2_sub(t2, t2, v0); fp2_sub(t2, t2, v2); fp2_add(o.c2, t2, v1);
    r = o;
}
H2V_DN bool fp6_inv(Fp6 &r, const Fp6 &a) {
    Fp2 t0, t1, t2, d, x;
    fp2_sqr(t0, a.c0); fp2_mul(x, a.c1, a.c2); fp2_mul_xi(x, x); fp2_sub(t0, t0, x);
    fp2_sqr(t1, a.c2); fp2_mul_xi(t1, t1); fp2_mul(x, a.c0, a.c1); fp2_sub(t1, t1, x);
    fp2_sqr(t2, a.c1); fp2_mul(x, a.c0, a.c2); fp2_sub(t2, t2, x);
    fp2_mul(d, a.c2, t1); fp2_mul(x, a.c1, t2); fp2_add(d, d, x); fp2_mul_xi(d, d);
    fp2_mul(x, a.c0, t0); fp2_add(d, d, x);
    bool ok = fp2_inv(d, d);
    fp2_mul(r.c0, t0, d); fp2_mul(r.c1, t1, d); fp2_mul(r.c2, t2, d);
    return ok;
}

// ------------------------------------------------------------------ Fp12
H2V_DI void fp12_set_one(Fp12 &r) { fp6_set_zero(r.c0); fp6_set_zero(r.c1); fp_set_one(r.c0.c0.c0); }
H2V_DI bool fp12_is_one(const Fp12 &a) {
    Fp12 o;
    fp12_set_one(o);
    return fp6_eq(a.c0, o.c0) && fp6_eq(a.c1, o.c1);
}
H2V_DN void fp12_mul(Fp12 &r, const Fp12 &a, const Fp12 &b) {
    Fp6 aa, bb, t0, t1;
    fp6_mul(aa, a.c0, b.c0);
    fp6_mul(bb, a.c1, b.c1);
    fp6_add(t0, a.c0, a.c1);
    fp6_add(t1, b.c0, b.c1);
    fp6_mul(t0, t0, t1);
    fp6_sub(t0, t0, aa);
    fp6_sub(r.c1, t0, bb);
    fp6_mul_v(bb, bb);
    fp6_add(r.c0, aa, bb);
}
// complex squaring: (c0 + c1 w)^2 = (c0+c1)(c0 + v c1) - c0c1 - v c0c1  +  2 c0c1 w
H2V_DN void fp12_sqr(Fp12 &r, const Fp12 &a) {
    Fp6 ab, t0, t1;
    fp6_mul(ab, a.c0, a.c1);
    fp6_add(t0, a.c0, a.c1);
    fp6_mul_v(t1, a.c1);
    fp6_add(t1, t1, a.c0);
    fp6_mul(t0, t0, t1);
    fp6_sub(t0, t0, ab);
    fp6_mul_v(t1, ab);
    fp6_sub(r.c0, t0, t1);
    fp6_add(r.c1, ab, ab);
}
H2V_DI void fp12_conj(Fp12 &r, const Fp12 &a) { r.c0 = a.c0; fp6_neg(r.c1, a.c1); }
H2V_DN bool fp12_inv(Fp12 &r, const Fp12 &a) {
    Fp6 t0, t1;
    fp6_mul(t0, a.c0, a.c0);
    fp6_mul(t1, a.c1, a.c1);
    fp6_mul_v(t1, t1);
    fp6_sub(t0, t0, t1);
    bool ok = fp6_inv(t0, t0);
    fp6_mul(r.c0, a.c0, t0);
    fp6_mul(t1, a.c1, t0);
    fp6_neg(r.c1, t1);
    return ok;
}
// f *= (c  +  b w^2  +  a w^3)  with c, b in Fp2 and a in Fp.  In tower coordinates the line is
//   L0 = (c, b, 0) in Fp6 (coefficient of w^0),  L1 = (0, a, 0) (coefficient of w).
H2V_DN void fp12_mul_line(Fp12 &f, const Fp2 &c, const Fp2 &b, const Fp &a) {
    // f0*L0 with L0 = (c, b, 0): 5 Fp2 mul ; f1*L1 with L1 = a*v: scale + shift
    Fp6 aa, bb, t;
    {   // aa = f.c0 * (c, b, 0)
        const Fp6 &x = f.c0;
        Fp2 v0, v1, s, u;
        fp2_mul(v0, x.c0, c);
        fp2_mul(v1, x.c1, b);
        // c0 = v0 + xi * (x2 * b)
        fp2_mul(s, x.c2, b); fp2_mul_xi(s, s); fp2_add(aa.c0, v0, s);
        // c1 = (x0+x1)(c+b) - v0 - v1
        fp2_add(s, x.c0, x.c1); fp2_add(u, c, b); fp2_mul(s, s, u); fp2_sub(s, s, v0); fp2_sub(aa.c1, s, v1);
        // c2 = x2*c + v1
        fp2_mul(s, x.c2, c); fp2_add(aa.c2, s, v1);
    }
    {   // bb = f.c1 * (a v): (y0,y1,y2)*v*a = (xi y2 a, y0 a, y1 a)
        const Fp6 &yv = f.c1;
        Fp2 s;
        fp2_mul_fp(s, yv.c2, a); fp2_mul_xi(bb.c0, s);
        fp2_mul_fp(bb.c1, yv.c0, a);
        fp2_mul_fp(bb.c2, yv.c1, a);
    }
    {   // t = (f0 + f1) * (L0 + L1) = (f0+f1) * (c, b + a, 0)
        Fp6 x;
        fp6_add(x, f.c0, f.c1);
        Fp2 b2 = b;
        fp_add(b2.c0, b2.c0, a);
        Fp2 v0, v1, s, u;
        fp2_mul(v0, x.c0, c);
        fp2_mul(v1, x.c1, b2);
        fp2_mul(s, x.c2, b2); fp2_mul_xi(s, s); fp2_add(t.c0, v0, s);
        fp2_add(s, x.c0, x.c1); fp2_add(u, c, b2); fp2_mul(s, s, u); fp2_sub(s, s, v0); fp2_sub(t.c1, s, v1);
        fp2_mul(s, x.c2, c); fp2_add(t.c2, s, v1);
    }
    fp6_sub(t, t, aa);
    fp6_sub(f.c1, t, bb);
    fp6_mul_v(bb, bb);
    fp6_add(f.c0, aa, bb);
}
H2V_DI void fp2_load_const(Fp2 &r, const uint32_t (&c0)[12], const uint32_t (&c1)[12]) {
#pragma unroll
    for (int i = 0; i < 12; i++) { r.c0.v[i] = c0[i]; r.c1.v[i] = c1[i]; }
}
// a -> a^p.  Flat coefficient of w^k is conjugated and multiplied by gamma^k; flat [a0,b0,a1,b1,a2,b2].
H2V_DN void fp12_frob(Fp12 &r, const Fp12 &a) {
    Fp2 g, t;
    Fp12 o;
    fp2_conj(o.c0.c0, a.c0.c0);
    fp2_load_const(g, FROB_G1_C0, FROB_G1_C1); fp2_conj(t, a.c1.c0); fp2_mul(o.c1.c0, t, g);
    fp2_load_const(g, FROB_G2_C0, FROB_G2_C1); fp2_conj(t, a.c0.c1); fp2_mul(o.c0.c1, t, g);
    fp2_load_const(g, FROB_G3_C0, FROB_G3_C1); fp2_conj(t, a.c1.c1); fp2_mul(o.c1.c1, t, g);
    fp2_load_const(g, FROB_G4_C0, FROB_G4_C1); fp2_conj(t, a.c0.c2); fp2_mul(o.c0.c2, t, g);
    fp2_load_const(g, FROB_G5_C0, FROB_G5_C1); fp2_conj(t, a.c1.c2); fp2_mul(o.c1.c2, t, g);
    r = o;
}
