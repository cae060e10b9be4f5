/* ORACLE - test infrastructure, not product code.
 *
 * Extension tower for the BLS12-381 pairing (public standard; the reference delegates it to
 * `midnight-curves =0.3.0` / Plutus builtins - verification_h2.hbs:125-128).  Fp2 arithmetic follows
 * plinth-verifier/plutus-halo2/src/Plutus/Crypto/BlsTypes.hs:302-380 (u^2 = -1).
 *   Fp2  = Fp[u]/(u^2+1)
 *   Fp6  = Fp2[v]/(v^3 - xi),  xi = 1+u
 *   Fp12 = Fp6[w]/(w^2 - v)
 */
#ifndef ORC_TOWER_H
#define ORC_TOWER_H
#include "field.h"

typedef struct { fp c0, c1; } fp2;
typedef struct { fp2 c0, c1, c2; } fp6;
typedef struct { fp6 c0, c1; } fp12;

static inline void fp2_zero(fp2 *r) { fp_zero(&r->c0); fp_zero(&r->c1); }
static inline void fp2_one(fp2 *r) { fp_one(&r->c0); fp_zero(&r->c1); }
static inline int fp2_is_zero(const fp2 *a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static inline int fp2_eq(const fp2 *a, const fp2 *b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
static inline void fp2_add(fp2 *r, const fp2 *a, const fp2 *b) { fp_add(&r->c0, &a->c0, &b->c0); fp_add(&r->c1, &a->c1, &b->c1); }
static inline void fp2_sub(fp2 *r, const fp2 *a, const fp2 *b) { fp_sub(&r->c0, &a->c0, &b->c0); fp_sub(&r->c1, &a->c1, &b->c1); }
static inline void fp2_neg(fp2 *r, const fp2 *a) { fp_neg(&r->c0, &a->c0); fp_neg(&r->c1, &a->c1); }
static inline void fp2_dbl(fp2 *r, const fp2 *a) { fp2_add(r, a, a); }
static inline void fp2_conj(fp2 *r, const fp2 *a) { r->c0 = a->c0; fp_neg(&r->c1, &a->c1); }
static inline void fp2_mul(fp2 *r, const fp2 *a, const fp2 *b) {
    fp t0, t1, t2, t3;
    fp_mul(&t0, &a->c0, &b->c0);
    fp_mul(&t1, &a->c1, &b->c1);
    fp_add(&t2, &a->c0, &a->c1);
    fp_add(&t3, &b->c0, &b->c1);
    fp_mul(&t2, &t2, &t3);
    fp_sub(&t2, &t2, &t0);
    fp_sub(&r->c1, &t2, &t1);
    fp_sub(&r->c0, &t0, &t1);
}
static inline void fp2_sqr(fp2 *r, const fp2 *a) {
    fp t0, t1, t2;
    fp_add(&t0, &a->c0, &a->c1);
    fp_sub(&t1, &a->c0, &a->c1);
    fp_mul(&t2, &a->c0, &a->c1);
    fp_mul(&r->c0, &t0, &t1);
    fp_dbl(&r->c1, &t2);
}
static inline void fp2_mul_fp(fp2 *r, const fp2 *a, const fp *k) { fp_mul(&r->c0, &a->c0, k); fp_mul(&r->c1, &a->c1, k); }
/* multiply by xi = 1 + u : (a0 - a1) + (a0 + a1) u */
static inline void fp2_mul_xi(fp2 *r, const fp2 *a) {
    fp t0, t1;
    fp_sub(&t0, &a->c0, &a->c1);
    fp_add(&t1, &a->c0, &a->c1);
    r->c0 = t0;
    r->c1 = t1;
}
static inline int fp2_inv(fp2 *r, const fp2 *a) {
    fp t0, t1;
    fp_sqr(&t0, &a->c0);
    fp_sqr(&t1, &a->c1);
    fp_add(&t0, &t0, &t1);
    if (!fp_inv(&t0, &t0)) return 0;
    fp_mul(&r->c0, &a->c0, &t0);
    fp_mul(&t1, &a->c1, &t0);
    fp_neg(&r->c1, &t1);
    return 1;
}
static inline void fp2_pow(fp2 *r, const fp2 *a, const uint64_t *e, int nlimbs) {
    fp2 acc;
    fp2_one(&acc);
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        fp2_sqr(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) fp2_mul(&acc, &acc, a);
    }
    *r = acc;
}

/* ------------------------------------------------------------------ Fp6 */
static inline void fp6_zero(fp6 *r) { fp2_zero(&r->c0); fp2_zero(&r->c1); fp2_zero(&r->c2); }
static inline void fp6_one(fp6 *r) { fp2_one(&r->c0); fp2_zero(&r->c1); fp2_zero(&r->c2); }
static inline void fp6_add(fp6 *r, const fp6 *a, const fp6 *b) { fp2_add(&r->c0, &a->c0, &b->c0); fp2_add(&r->c1, &a->c1, &b->c1); fp2_add(&r->c2, &a->c2, &b->c2); }
static inline void fp6_sub(fp6 *r, const fp6 *a, const fp6 *b) { fp2_sub(&r->c0, &a->c0, &b->c0); fp2_sub(&r->c1, &a->c1, &b->c1); fp2_sub(&r->c2, &a->c2, &b->c2); }
static inline void fp6_neg(fp6 *r, const fp6 *a) { fp2_neg(&r->c0, &a->c0); fp2_neg(&r->c1, &a->c1); fp2_neg(&r->c2, &a->c2); }
static inline int fp6_eq(const fp6 *a, const fp6 *b) { return fp2_eq(&a->c0, &b->c0) && fp2_eq(&a->c1, &b->c1) && fp2_eq(&a->c2, &b->c2); }
/* multiply by v: (c0, c1, c2) -> (xi*c2, c0, c1) */
static inline void fp6_mul_v(fp6 *r, const fp6 *a) {
    fp2 t;
    fp2_mul_xi(&t, &a->c2);
    r->c2 = a->c1;
    r->c1 = a->c0;
    r->c0 = t;
}
static inline void fp6_mul(fp6 *r, const fp6 *a, const fp6 *b) {
    /* schoolbook, 9 Fp2 products: simple and obviously right */
    fp2 a0b0, a0b1, a0b2, a1b0, a1b1, a1b2, a2b0, a2b1, a2b2, t;
    fp2_mul(&a0b0, &a->c0, &b->c0); fp2_mul(&a0b1, &a->c0, &b->c1); fp2_mul(&a0b2, &a->c0, &b->c2);
    fp2_mul(&a1b0, &a->c1, &b->c0); fp2_mul(&a1b1, &a->c1, &b->c1); fp2_mul(&a1b2, &a->c1, &b->c2);
    fp2_mul(&a2b0, &a->c2, &b->c0); fp2_mul(&a2b1, &a->c2, &b->c1); fp2_mul(&a2b2, &a->c2, &b->c2);
    fp6 o;
    fp2_add(&t, &a1b2, &a2b1); fp2_mul_xi(&t, &t); fp2_add(&o.c0, &a0b0, &t);
    fp2_mul_xi(&t, &a2b2); fp2_add(&o.c1, &a0b1, &a1b0); fp2_add(&o.c1, &o.c1, &t);
    fp2_add(&o.c2, &a0b2, &a1b1); fp2_add(&o.c2, &o.c2, &a2b0);
    *r = o;
}
static inline int fp6_inv(fp6 *r, const fp6 *a) {
    fp2 t0, t1, t2, d, x;
    /* t0 = c0^2 - xi c1 c2 ; t1 = xi c2^2 - c0 c1 ; t2 = c1^2 - c0 c2 */
    fp2_sqr(&t0, &a->c0); fp2_mul(&x, &a->c1, &a->c2); fp2_mul_xi(&x, &x); fp2_sub(&t0, &t0, &x);
    fp2_sqr(&t1, &a->c2); fp2_mul_xi(&t1, &t1); fp2_mul(&x, &a->c0, &a->c1); fp2_sub(&t1, &t1, &x);
    fp2_sqr(&t2, &a->c1); fp2_mul(&x, &a->c0, &a->c2); fp2_sub(&t2, &t2, &x);
    /* d = c0 t0 + xi (c2 t1 + c1 t2) */
    fp2_mul(&d, &a->c2, &t1); fp2_mul(&x, &a->c1, &t2); fp2_add(&d, &d, &x); fp2_mul_xi(&d, &d);
    fp2_mul(&x, &a->c0, &t0); fp2_add(&d, &d, &x);
    if (!fp2_inv(&d, &d)) return 0;
    fp2_mul(&r->c0, &t0, &d); fp2_mul(&r->c1, &t1, &d); fp2_mul(&r->c2, &t2, &d);
    return 1;
}

/* ------------------------------------------------------------------ Fp12 */
static inline void fp12_one(fp12 *r) { fp6_one(&r->c0); fp6_zero(&r->c1); }
static inline int fp12_eq(const fp12 *a, const fp12 *b) { return fp6_eq(&a->c0, &b->c0) && fp6_eq(&a->c1, &b->c1); }
static inline int fp12_is_one(const fp12 *a) { fp12 o; fp12_one(&o); return fp12_eq(a, &o); }
static inline void fp12_mul(fp12 *r, const fp12 *a, const fp12 *b) {
    fp6 aa, bb, t0, t1;
    fp6_mul(&aa, &a->c0, &b->c0);
    fp6_mul(&bb, &a->c1, &b->c1);
    fp6_add(&t0, &a->c0, &a->c1);
    fp6_add(&t1, &b->c0, &b->c1);
    fp6_mul(&t0, &t0, &t1);
    fp6_sub(&t0, &t0, &aa);
    fp6_sub(&r->c1, &t0, &bb);
    fp6_mul_v(&bb, &bb);
    fp6_add(&r->c0, &aa, &bb);
}
static inline void fp12_sqr(fp12 *r, const fp12 *a) { fp12_mul(r, a, a); }
static inline void fp12_conj(fp12 *r, const fp12 *a) { r->c0 = a->c0; fp6_neg(&r->c1, &a->c1); }
static inline int fp12_inv(fp12 *r, const fp12 *a) {
    /* 1/(c0 + c1 w) = (c0 - c1 w)/(c0^2 - v c1^2) */
    fp6 t0, t1;
    fp6_mul(&t0, &a->c0, &a->c0);
    fp6_mul(&t1, &a->c1, &a->c1);
    fp6_mul_v(&t1, &t1);
    fp6_sub(&t0, &t0, &t1);
    if (!fp6_inv(&t0, &t0)) return 0;
    fp6_mul(&r->c0, &a->c0, &t0);
    fp6_mul(&t1, &a->c1, &t0);
    fp6_neg(&r->c1, &t1);
    return 1;
}
/* Frobenius a -> a^p.  Flat view: coefficient of w^k is conj'd and multiplied by gamma^k, with
 * flat [a0,b0,a1,b1,a2,b2] == tower (c0=(a0,a1,a2), c1=(b0,b1,b2)). */
static inline void fp12_frob(fp12 *r, const fp12 *a) {
    fp2 g1, g2, g3, g4, g5, t;
    fp_set(&g1.c0, FROB_G1_0); fp_set(&g1.c1, FROB_G1_1);
    fp_set(&g2.c0, FROB_G2_0); fp_set(&g2.c1, FROB_G2_1);
    fp_set(&g3.c0, FROB_G3_0); fp_set(&g3.c1, FROB_G3_1);
    fp_set(&g4.c0, FROB_G4_0); fp_set(&g4.c1, FROB_G4_1);
    fp_set(&g5.c0, FROB_G5_0); fp_set(&g5.c1, FROB_G5_1);
    fp12 o;
    fp2_conj(&o.c0.c0, &a->c0.c0);                                   /* w^0 */
    fp2_conj(&t, &a->c1.c0); fp2_mul(&o.c1.c0, &t, &g1);             /* w^1 */
    fp2_conj(&t, &a->c0.c1); fp2_mul(&o.c0.c1, &t, &g2);             /* w^2 */
    fp2_conj(&t, &a->c1.c1); fp2_mul(&o.c1.c1, &t, &g3);             /* w^3 */
    fp2_conj(&t, &a->c0.c2); fp2_mul(&o.c0.c2, &t, &g4);             /* w^4 */
    fp2_conj(&t, &a->c1.c2); fp2_mul(&o.c1.c2, &t, &g5);             /* w^5 */
    *r = o;
}
#endif
