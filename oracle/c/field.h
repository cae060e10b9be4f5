/* ORACLE - test infrastructure, not product code.
 *
 * CPU restatement of the prime-field arithmetic the reference delegates to `midnight-curves =0.3.0`
 * (Cargo.toml:25-27; not vendored under /root/reference) and that its Plinth twin spells out in
 * plinth-verifier/plutus-halo2/src/Plutus/Crypto/BlsTypes.hs:96-380 (Scalar :105-212, Fp :214-300):
 * add / sub / neg / mul / powMod (square-and-multiply, :186-192) / recip (:201-212).
 * Here: 64-bit limbs, Montgomery form, R = 2^384 (Fp) and 2^256 (Fr); all values kept fully reduced.
 */
#ifndef ORC_FIELD_H
#define ORC_FIELD_H
#include <stdint.h>
#include <string.h>
#include "consts.h"

typedef unsigned __int128 u128;
typedef struct { uint64_t l[6]; } fp;
typedef struct { uint64_t l[4]; } fr;

/* ---- generic n-limb helpers (n is a compile-time constant at every call site) */
static inline __attribute__((always_inline)) int mp_geq(const uint64_t *a, const uint64_t *b, int n) {
    for (int i = n - 1; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static inline __attribute__((always_inline)) uint64_t mp_add(uint64_t *r, const uint64_t *a, const uint64_t *b, int n) {
    u128 c = 0;
    for (int i = 0; i < n; i++) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static inline __attribute__((always_inline)) uint64_t mp_sub(uint64_t *r, const uint64_t *a, const uint64_t *b, int n) {
    uint64_t borrow = 0;
    for (int i = 0; i < n; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
static inline __attribute__((always_inline)) int mp_is_zero(const uint64_t *a, int n) {
    uint64_t x = 0;
    for (int i = 0; i < n; i++) x |= a[i];
    return x == 0;
}
static inline __attribute__((always_inline)) void mod_add(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *m, int n) {
    uint64_t t[6];
    uint64_t c = mp_add(t, a, b, n);
    if (c || mp_geq(t, m, n)) mp_sub(t, t, m, n);
    memcpy(r, t, n * 8);
}
static inline __attribute__((always_inline)) void mod_sub(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *m, int n) {
    uint64_t t[6];
    if (mp_sub(t, a, b, n)) mp_add(t, t, m, n);
    memcpy(r, t, n * 8);
}
/* CIOS Montgomery multiplication */
static inline __attribute__((always_inline)) void mont_mul(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *m, uint64_t n0, int n) {
    uint64_t t[8] = {0};
    for (int i = 0; i < n; i++) {
        u128 s;
        uint64_t carry = 0;
        for (int j = 0; j < n; j++) {
            s = (u128)a[j] * b[i] + t[j] + carry;
            t[j] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
        s = (u128)t[n] + carry;
        t[n] = (uint64_t)s;
        t[n + 1] = (uint64_t)(s >> 64);
        uint64_t q = t[0] * n0;
        s = (u128)q * m[0] + t[0];
        carry = (uint64_t)(s >> 64);
        for (int j = 1; j < n; j++) {
            s = (u128)q * m[j] + t[j] + carry;
            t[j - 1] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
        s = (u128)t[n] + carry;
        t[n - 1] = (uint64_t)s;
        t[n] = t[n + 1] + (uint64_t)(s >> 64);
    }
    if (t[n] || mp_geq(t, m, n)) mp_sub(t, t, m, n);
    memcpy(r, t, n * 8);
}

/* ------------------------------------------------------------------ Fp */
static inline __attribute__((always_inline)) void fp_set(fp *r, const uint64_t *c) { memcpy(r->l, c, 48); }
static inline __attribute__((always_inline)) void fp_zero(fp *r) { memset(r->l, 0, 48); }
static inline __attribute__((always_inline)) void fp_one(fp *r) { fp_set(r, FP_ONE); }
static inline __attribute__((always_inline)) int fp_is_zero(const fp *a) { return mp_is_zero(a->l, 6); }
static inline __attribute__((always_inline)) int fp_eq(const fp *a, const fp *b) { return memcmp(a->l, b->l, 48) == 0; }
static inline __attribute__((always_inline)) void fp_add(fp *r, const fp *a, const fp *b) { mod_add(r->l, a->l, b->l, FP_MOD, 6); }
static inline __attribute__((always_inline)) void fp_sub(fp *r, const fp *a, const fp *b) { mod_sub(r->l, a->l, b->l, FP_MOD, 6); }
static inline __attribute__((always_inline)) void fp_neg(fp *r, const fp *a) {
    if (fp_is_zero(a)) { *r = *a; return; }
    mp_sub(r->l, FP_MOD, a->l, 6);
}
static inline __attribute__((always_inline)) void fp_dbl(fp *r, const fp *a) { fp_add(r, a, a); }
static inline __attribute__((always_inline)) void fp_mul(fp *r, const fp *a, const fp *b) { mont_mul(r->l, a->l, b->l, FP_MOD, FP_N0, 6); }
static inline __attribute__((always_inline)) void fp_sqr(fp *r, const fp *a) { fp_mul(r, a, a); }
static inline __attribute__((always_inline)) void fp_pow(fp *r, const fp *a, const uint64_t *e, int nlimbs) {
    fp acc;
    fp_one(&acc);
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        fp_sqr(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) fp_mul(&acc, &acc, a);
    }
    *r = acc;
}
/* returns 0 if a == 0 (no inverse) */
static inline __attribute__((always_inline)) int fp_inv(fp *r, const fp *a) {
    if (fp_is_zero(a)) return 0;
    fp_pow(r, a, FP_INV_EXP, 6);
    return 1;
}
/* canonical integer (out of Montgomery form) */
static inline __attribute__((always_inline)) void fp_to_plain(uint64_t out[6], const fp *a) {
    uint64_t one[6] = {1, 0, 0, 0, 0, 0};
    mont_mul(out, a->l, one, FP_MOD, FP_N0, 6);
}
static inline __attribute__((always_inline)) void fp_from_plain(fp *r, const uint64_t in[6]) { mont_mul(r->l, in, FP_R2, FP_MOD, FP_N0, 6); }
/* 48-byte big-endian <-> fp; returns 0 if the integer is >= p */
static inline __attribute__((always_inline)) int fp_from_be48(fp *r, const uint8_t *b) {
    uint64_t t[6];
    for (int i = 0; i < 6; i++) {
        uint64_t v = 0;
        for (int j = 0; j < 8; j++) v = (v << 8) | b[(5 - i) * 8 + j];
        t[i] = v;
    }
    if (mp_geq(t, FP_MOD, 6)) return 0;
    fp_from_plain(r, t);
    return 1;
}
static inline __attribute__((always_inline)) void fp_to_be48(uint8_t *b, const fp *a) {
    uint64_t t[6];
    fp_to_plain(t, a);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 8; j++) b[(5 - i) * 8 + j] = (uint8_t)(t[i] >> (56 - 8 * j));
}
/* y > p - y  <=>  y > (p-1)/2   ("lexicographically larger", bls_utils.ak:35-43) */
static inline __attribute__((always_inline)) int fp_is_lex_larger(const fp *a) {
    uint64_t t[6];
    fp_to_plain(t, a);
    return !mp_geq(FP_HALF, t, 6);
}

/* ------------------------------------------------------------------ Fr */
static inline __attribute__((always_inline)) void fr_set(fr *r, const uint64_t *c) { memcpy(r->l, c, 32); }
static inline __attribute__((always_inline)) void fr_zero(fr *r) { memset(r->l, 0, 32); }
static inline __attribute__((always_inline)) void fr_one(fr *r) { fr_set(r, FR_ONE); }
static inline __attribute__((always_inline)) int fr_is_zero(const fr *a) { return mp_is_zero(a->l, 4); }
static inline __attribute__((always_inline)) int fr_eq(const fr *a, const fr *b) { return memcmp(a->l, b->l, 32) == 0; }
static inline __attribute__((always_inline)) void fr_add(fr *r, const fr *a, const fr *b) { mod_add(r->l, a->l, b->l, FR_MOD, 4); }
static inline __attribute__((always_inline)) void fr_sub(fr *r, const fr *a, const fr *b) { mod_sub(r->l, a->l, b->l, FR_MOD, 4); }
static inline __attribute__((always_inline)) void fr_neg(fr *r, const fr *a) {
    if (fr_is_zero(a)) { *r = *a; return; }
    mp_sub(r->l, FR_MOD, a->l, 4);
}
static inline __attribute__((always_inline)) void fr_mul(fr *r, const fr *a, const fr *b) { mont_mul(r->l, a->l, b->l, FR_MOD, FR_N0, 4); }
static inline __attribute__((always_inline)) void fr_sqr(fr *r, const fr *a) { fr_mul(r, a, a); }
static inline __attribute__((always_inline)) void fr_pow_u64(fr *r, const fr *a, uint64_t e) {
    fr acc;
    fr_one(&acc);
    for (int i = 63; i >= 0; i--) {
        fr_sqr(&acc, &acc);
        if ((e >> i) & 1) fr_mul(&acc, &acc, a);
    }
    *r = acc;
}
/* recip (BlsTypes.hs:201-212 / bls_utils.ak:98-117 use EEA; any exact inverse is the same field element).
 * Returns 0 for a == 0: the reference's recip_eea divides by zero there => script failure => reject. */
static inline __attribute__((always_inline)) int fr_inv(fr *r, const fr *a) {
    if (fr_is_zero(a)) return 0;
    fr acc;
    fr_one(&acc);
    for (int i = 255; i >= 0; i--) {
        fr_sqr(&acc, &acc);
        if ((FR_INV_EXP[i / 64] >> (i % 64)) & 1) fr_mul(&acc, &acc, a);
    }
    *r = acc;
    return 1;
}
static inline __attribute__((always_inline)) void fr_to_plain(uint64_t out[4], const fr *a) {
    uint64_t one[4] = {1, 0, 0, 0};
    mont_mul(out, a->l, one, FR_MOD, FR_N0, 4);
}
static inline __attribute__((always_inline)) void fr_from_plain(fr *r, const uint64_t in[4]) { mont_mul(r->l, in, FR_R2, FR_MOD, FR_N0, 4); }
static inline __attribute__((always_inline)) void fr_from_u64(fr *r, uint64_t v) {
    uint64_t t[4] = {v, 0, 0, 0};
    fr_from_plain(r, t);
}
/* 32-byte little-endian; returns 0 (and still reduces) when the integer is >= r */
static inline __attribute__((always_inline)) int fr_from_le32(fr *r, const uint8_t *b) {
    uint64_t t[4];
    for (int i = 0; i < 4; i++) {
        uint64_t v = 0;
        for (int j = 7; j >= 0; j--) v = (v << 8) | b[i * 8 + j];
        t[i] = v;
    }
    int canonical = !mp_geq(t, FR_MOD, 4);
    /* from_plain of a value < 2^256 is fine for Montgomery mul even when >= r (result reduced) */
    fr_from_plain(r, t);
    return canonical;
}
static inline __attribute__((always_inline)) void fr_to_le32(uint8_t *b, const fr *a) {
    uint64_t t[4];
    fr_to_plain(t, a);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) b[i * 8 + j] = (uint8_t)(t[i] >> (8 * j));
}
#endif
