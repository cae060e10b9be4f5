// Lazily reduced Fp for the long G1 chains (MSM ladder, subgroup test, square root): 14 x 28-bit limbs that stay in
// the multiplier's own radix between operations.
//
// Why: with 12 x 32-bit storage every multiplication re-cuts both operands to 28-bit limbs, re-packs the result and
// runs a conditional subtraction (~230 of the 618 instructions of fp_mul_raw), and every addition is two dependent
// carry chains plus a select (~60-85 issue slots).  Here additions are 14 independent v_add, subtractions add a
// pre-spread multiple of p first, and a value is only brought back to normal form where the bounds below need it.
//
// Bounds (checked by hand per formula in h2v_curve28.hpp; the generator asserts the bias tables):
//   an element is described by (v, lam):  value < v*p  and every limb < lam * 2^28  (top limb: whatever v implies)
//   * f28_mul / f28_sqr need lam_a * lam_b <= 17 (14 products + 14 reduction terms per 64-bit column) and
//     v_a * v_b <= 2048 = R/p-ish (R = 2^392), and return (2, 1)
//   * f28_add: (v_a + v_b, lam_a + lam_b);  f28_mul_small(k): (k v, k lam);  lam must stay <= 15 (32-bit limbs)
//   * f28_sub<K, M>(a, b) = a + (K p spread so that every limb >= M 2^28) - b needs v_b <= K - 1, lam_b <= M and
//     returns (v_a + K, lam_a + M + 2)
//   * f28_carry: limbs back below 2^28 (value unchanged): (v, 1)
// Montgomery constant and domain are those of h2v_field.hpp (R = 2^392), so conversion is a re-cut of the limbs.
#pragma once
#include "h2v_field.hpp"

struct F28 { uint32_t l[14]; };
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
struct F28Regs { u32x4 a, b, c; u32x2 d; };

H2V_DI F28Regs f28_pack(const F28 &x) {
    F28Regs r;
    r.a = u32x4{x.l[0], x.l[1], x.l[2], x.l[3]};
    r.b = u32x4{x.l[4], x.l[5], x.l[6], x.l[7]};
    r.c = u32x4{x.l[8], x.l[9], x.l[10], x.l[11]};
    r.d = u32x2{x.l[12], x.l[13]};
    return r;
}
H2V_DI F28 f28_unpack(const u32x4 a, const u32x4 b, const u32x4 c, const u32x2 d) {
    F28 x;
    x.l[0] = a.x; x.l[1] = a.y; x.l[2] = a.z; x.l[3] = a.w;
    x.l[4] = b.x; x.l[5] = b.y; x.l[6] = b.z; x.l[7] = b.w;
    x.l[8] = c.x; x.l[9] = c.y; x.l[10] = c.z; x.l[11] = c.w;
    x.l[12] = d.x; x.l[13] = d.y;
    return x;
}
// out-of-line multiplier, operands and result in VGPRs (see the ABI note in h2v_field.hpp)
H2V_DN F28Regs f28_mul_raw(u32x4 a0, u32x4 a1, u32x4 a2, u32x2 a3, u32x4 b0, u32x4 b1, u32x4 b2, u32x2 b3) {
    const F28 a = f28_unpack(a0, a1, a2, a3), b = f28_unpack(b0, b1, b2, b3);
    F28 t;
    fp_mont28(t.l, a.l, b.l);
    return f28_pack(t);
}
H2V_DN F28Regs f28_sqr_raw(u32x4 a0, u32x4 a1, u32x4 a2, u32x2 a3) {
    const F28 a = f28_unpack(a0, a1, a2, a3);
    F28 t;
    fp_montsqr28(t.l, a.l);
    return f28_pack(t);
}
H2V_DI void f28_mul(F28 &r, const F28 &a, const F28 &b) {
    const F28Regs x = f28_pack(a), y = f28_pack(b);
    const F28Regs z = f28_mul_raw(x.a, x.b, x.c, x.d, y.a, y.b, y.c, y.d);
    r = f28_unpack(z.a, z.b, z.c, z.d);
}
H2V_DI void f28_sqr(F28 &r, const F28 &a) {
    const F28Regs x = f28_pack(a);
    const F28Regs z = f28_sqr_raw(x.a, x.b, x.c, x.d);
    r = f28_unpack(z.a, z.b, z.c, z.d);
}
// inlined forms (no call, no argument marshalling) for the one loop that is short enough to afford the code size
H2V_DI void f28_mul_inl(F28 &r, const F28 &a, const F28 &b) { F28 t; fp_mont28(t.l, a.l, b.l); r = t; }
H2V_DI void f28_sqr_inl(F28 &r, const F28 &a) { F28 t; fp_montsqr28(t.l, a.l); r = t; }
H2V_DI void f28_add(F28 &r, const F28 &a, const F28 &b) {
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + b.l[i];
}
template <int K>
H2V_DI void f28_mul_small(F28 &r, const F28 &a) {
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] * (uint32_t)K;   // K = 2, 3, 8: shifts / shift-adds
}
// r = a - b + K*p   (bias table generated and asserted by tools/gen_device_consts.py)
#define F28_SUB(r, a, b, K, M)                                                               \
    do {                                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 14; i_++)                                    \
            (r).l[i_] = (a).l[i_] + (F28_BIAS_##K##_##M[i_] - (b).l[i_]);                    \
    } while (0)
// r = K*p - b
#define F28_NEG(r, b, K, M)                                                                  \
    do {                                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 14; i_++) (r).l[i_] = F28_BIAS_##K##_##M[i_] - (b).l[i_]; \
    } while (0)
H2V_DI void f28_carry(F28 &a) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        const uint32_t t = a.l[i] + c;
        a.l[i] = t & FP28_MASK;
        c = t >> 28;
    }
    a.l[13] += c;
}
// Fold a carried element with value < 32p below 2p (and a hair): the quotient by p is estimated from the top limb (never too
// large, at most one too small) and q p subtracted with signed carries.  (tools/gen_six_tables.py: fold - the same integer
// steps, checked on every multiple of p up to 32p, its neighbours and random values.)
H2V_DI void f28_fold(F28 &a) {
    const uint32_t q = __umulhi(a.l[13], FP_FOLD_M);
    int64_t t = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        t += (int64_t)a.l[i] - (int64_t)((uint64_t)q * FP_MOD28[i]);
        a.l[i] = i < 13 ? (uint32_t)t & FP28_MASK : (uint32_t)t;
        t >>= 28;
    }
}
// Exact test "a == 0 (mod p)" for a carried element with value < 5p: compare with 0, p, 2p, 3p, 4p.
H2V_DI bool f28_is_zero_v5(const F28 &a) {
    uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        d0 |= a.l[i];
        d1 |= a.l[i] ^ F28_KP[1][i];
        d2 |= a.l[i] ^ F28_KP[2][i];
        d3 |= a.l[i] ^ F28_KP[3][i];
        d4 |= a.l[i] ^ F28_KP[4][i];
    }
    return d0 == 0 || d1 == 0 || d2 == 0 || d3 == 0 || d4 == 0;
}
H2V_DI void f28_set_one(F28 &r) {
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = FP_ONE28[i];
}
H2V_DI void f28_set_zero(F28 &r) {
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = 0;
}
// canonical Fp (fully reduced Montgomery limbs) -> (1, 1)
H2V_DI void f28_from_fp(F28 &r, const Fp &a) { fp_to28(r.l, a); }
// any carried element with v <= 1024 -> canonical Fp: one multiplication by the Montgomery one brings the value
// below 2p, then the usual re-pack and conditional subtraction.
H2V_DI void f28_to_fp(Fp &r, const F28 &a) {
    F28 one, t;
    f28_set_one(one);
    f28_mul(t, a, one);
    uint32_t w[12];
    fp_from28(w, t.l);
    FpF::cond_sub(r, w, 0);
}
// r = a^e for a fixed public exponent (MSB-first square-and-multiply); a is (2, 1) or better
template <int NW>
H2V_DN void f28_pow_const(F28 &r, const F28 &a, const uint32_t (&e)[NW]) {
    F28 acc;
    f28_set_one(acc);
    bool started = false;
#pragma unroll 1
    for (int i = NW * 32 - 1; i >= 0; i--) {
        if (started) f28_sqr_inl(acc, acc);     // inlined: the loop is 380 squarings + 190 products long, and the
        if ((e[i >> 5] >> (i & 31)) & 1) {      // argument marshalling of a call is a fifth of its instructions
            if (started) f28_mul_inl(acc, acc, a);
            else { acc = a; started = true; }
        }
    }
    r = acc;
}

// r = a^((p+1)/4) along the generated sliding-window chain (bls_consts.h: FP_SQRT_CHAIN; odd powers a, a^3 .. a^15 kept in
// registers and picked with a wave-uniform switch): 377 squarings + 78 + 8 products instead of 380 + 190.  a: (2, 1) or better.
H2V_DN void f28_sqrt_chain(F28 &r, const F28 &a) {
    F28 t0 = a, t1, t2, t3, t4, t5, t6, t7, a2;
    f28_sqr(a2, a);
    f28_mul(t1, t0, a2); f28_mul(t2, t1, a2); f28_mul(t3, t2, a2); f28_mul(t4, t3, a2);
    f28_mul(t5, t4, a2); f28_mul(t6, t5, a2); f28_mul(t7, t6, a2);
#define F28_PICK(dst, idx)                                                                   \
    do {                                                                                     \
        switch (idx) {                                                                       \
        case 0: dst = t0; break; case 1: dst = t1; break; case 2: dst = t2; break; case 3: dst = t3; break; \
        case 4: dst = t4; break; case 5: dst = t5; break; case 6: dst = t6; break; default: dst = t7; break; \
        }                                                                                    \
    } while (0)
    F28 acc, op;
    F28_PICK(acc, FP_SQRT_CHAIN[0][1]);
#pragma unroll 1
    for (int k = 1; k < FP_SQRT_CHAIN_LEN; k++) {
        const int nsq = FP_SQRT_CHAIN[k][0], idx = FP_SQRT_CHAIN[k][1];
#pragma unroll 1
        for (int q = 0; q < nsq; q++) f28_sqr_inl(acc, acc);
        if (idx != 255) {
            F28_PICK(op, idx);
            f28_mul_inl(acc, acc, op);
        }
    }
#undef F28_PICK
    r = acc;
}
