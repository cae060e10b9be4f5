// gfx950 device arithmetic for the BLS12-381 base field Fp (12 x u32) and scalar field Fr (8 x u32).
//
// Montgomery form (R = 2^384 / 2^256), values kept fully reduced after every operation so that equality
// is limb equality.  The multiplier is v_mad_u64_u32 (32x32+64 -> 64): one CIOS row is N of them for the
// product and N for the reduction, carries ride in the 64-bit accumulator.
// These are the operations the reference names at
//   plinth-verifier/plutus-halo2/src/Plutus/Crypto/BlsTypes.hs:96-300 (Scalar / Fp: add, sub, neg, mul, powMod, recip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bls_consts.h"

#define H2V_DI __device__ __forceinline__
#define H2V_DN __device__ __noinline__

template <int N>
struct Big {
    uint32_t v[N];
};
using Fp = Big<12>;
using Fr = Big<8>;

struct FpParams {
    static constexpr int N = 12;
    H2V_DI static uint32_t mod(int i) { return FP_MOD[i]; }
    H2V_DI static uint32_t one(int i) { return FP_ONE[i]; }
    H2V_DI static uint32_t r2(int i) { return FP_R2[i]; }
    static constexpr uint32_t n0 = FP_N0;
};
struct FrParams {
    static constexpr int N = 8;
    H2V_DI static uint32_t mod(int i) { return FR_MOD[i]; }
    H2V_DI static uint32_t one(int i) { return FR_ONE[i]; }
    H2V_DI static uint32_t r2(int i) { return FR_R2[i]; }
    static constexpr uint32_t n0 = FR_N0;
};

template <class PR>
struct Field {
    static constexpr int N = PR::N;
    using T = Big<N>;

    H2V_DI static void set_zero(T &r) {
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = 0;
    }
    H2V_DI static void set_one(T &r) {
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = PR::one(i);
    }
    H2V_DI static bool is_zero(const T &a) {
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < N; i++) x |= a.v[i];
        return x == 0;
    }
    H2V_DI static bool eq(const T &a, const T &b) {
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < N; i++) x |= a.v[i] ^ b.v[i];
        return x == 0;
    }
    // a >= modulus ?
    H2V_DI static bool geq_mod(const uint32_t *a) {
        // borrow of a - mod
        uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t d = (uint64_t)a[i] - PR::mod(i) - br;
            br = (d >> 63) & 1;
        }
        return br == 0;
    }
    // r = a - mod if a >= mod (a < 2*mod), with optional extra top carry
    H2V_DI static void cond_sub(T &r, const uint32_t *a, uint32_t top) {
        uint32_t d[N];
        uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t t = (uint64_t)a[i] - PR::mod(i) - br;
            d[i] = (uint32_t)t;
            br = (t >> 63) & 1;
        }
        bool use = (top != 0) || (br == 0);
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = use ? d[i] : a[i];
    }
    H2V_DI static void add(T &r, const T &a, const T &b) {
        uint32_t s[N];
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            c += (uint64_t)a.v[i] + b.v[i];
            s[i] = (uint32_t)c;
            c >>= 32;
        }
        cond_sub(r, s, (uint32_t)c);
    }
    H2V_DI static void sub(T &r, const T &a, const T &b) {
        uint32_t d[N];
        uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t t = (uint64_t)a.v[i] - b.v[i] - br;
            d[i] = (uint32_t)t;
            br = (t >> 63) & 1;
        }
        uint32_t mask = (uint32_t)0 - (uint32_t)br;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            c += (uint64_t)d[i] + (PR::mod(i) & mask);
            r.v[i] = (uint32_t)c;
            c >>= 32;
        }
    }
    H2V_DI static void neg(T &r, const T &a) {
        T z;
        set_zero(z);
        sub(r, z, a);
    }
    H2V_DI static void dbl(T &r, const T &a) { add(r, a, a); }

    // CIOS Montgomery product r = a*b/R mod m
    H2V_DI static void mul(T &r, const T &a, const T &b) {
        uint32_t t[N + 2];
#pragma unroll
        for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t c = 0;
            const uint32_t bi = b.v[i];
#pragma unroll
            for (int j = 0; j < N; j++) {
                c = (uint64_t)a.v[j] * bi + t[j] + c;
                t[j] = (uint32_t)c;
                c >>= 32;
            }
            c += t[N];
            t[N] = (uint32_t)c;
            t[N + 1] = (uint32_t)(c >> 32);
            const uint32_t q = t[0] * PR::n0;
            c = (uint64_t)q * PR::mod(0) + t[0];
            c >>= 32;
#pragma unroll
            for (int j = 1; j < N; j++) {
                c = (uint64_t)q * PR::mod(j) + t[j] + c;
                t[j - 1] = (uint32_t)c;
                c >>= 32;
            }
            c += t[N];
            t[N - 1] = (uint32_t)c;
            t[N] = t[N + 1] + (uint32_t)(c >> 32);
        }
        cond_sub(r, t, t[N]);
    }
    H2V_DI static void sqr(T &r, const T &a) { mul(r, a, a); }

    H2V_DI static void to_mont(T &r, const T &plain) {
        T k;
#pragma unroll
        for (int i = 0; i < N; i++) k.v[i] = PR::r2(i);
        mul(r, plain, k);
    }
    H2V_DI static void from_mont(T &r, const T &a) {
        T one;
        set_zero(one);
        one.v[0] = 1;
        mul(r, a, one);
    }
};

using FpF = Field<FpParams>;
using FrF = Field<FrParams>;

// ---- out-of-line wrappers: keep the big kernels' code size and compile time in check.
H2V_DN void fp_mul(Fp &r, const Fp &a, const Fp &b) { FpF::mul(r, a, b); }
H2V_DI void fp_sqr(Fp &r, const Fp &a) { fp_mul(r, a, a); }
H2V_DI void fp_add(Fp &r, const Fp &a, const Fp &b) { FpF::add(r, a, b); }
H2V_DI void fp_sub(Fp &r, const Fp &a, const Fp &b) { FpF::sub(r, a, b); }
H2V_DI void fp_neg(Fp &r, const Fp &a) { FpF::neg(r, a); }
H2V_DI void fp_dbl(Fp &r, const Fp &a) { FpF::add(r, a, a); }
H2V_DI bool fp_is_zero(const Fp &a) { return FpF::is_zero(a); }
H2V_DI bool fp_eq(const Fp &a, const Fp &b) { return FpF::eq(a, b); }
H2V_DI void fp_set_one(Fp &r) { FpF::set_one(r); }
H2V_DI void fp_set_zero(Fp &r) { FpF::set_zero(r); }

H2V_DN void fr_mul(Fr &r, const Fr &a, const Fr &b) { FrF::mul(r, a, b); }
H2V_DI void fr_add(Fr &r, const Fr &a, const Fr &b) { FrF::add(r, a, b); }
H2V_DI void fr_sub(Fr &r, const Fr &a, const Fr &b) { FrF::sub(r, a, b); }

// a^e for a fixed public exponent given as limbs (uniform control flow across lanes)
template <int EL>
H2V_DN void fp_pow_const(Fp &r, const Fp &a, const uint32_t (&e)[EL]) {
    Fp acc;
    fp_set_one(acc);
    bool started = false;
    for (int i = EL * 32 - 1; i >= 0; i--) {
        if (started) fp_sqr(acc, acc);
        if ((e[i >> 5] >> (i & 31)) & 1) {
            if (started) fp_mul(acc, acc, a);
            else { acc = a; started = true; }
        }
    }
    r = acc;
}
// returns false when a == 0
H2V_DN bool fp_inv(Fp &r, const Fp &a) {
    if (fp_is_zero(a)) { fp_set_zero(r); return false; }
    fp_pow_const<12>(r, a, FP_INV_EXP);
    return true;
}
H2V_DN bool fr_inv(Fr &r, const Fr &a) {
    if (FrF::is_zero(a)) { FrF::set_zero(r); return false; }
    Fr acc;
    FrF::set_one(acc);
    bool started = false;
    for (int i = 255; i >= 0; i--) {
        if (started) fr_mul(acc, acc, acc);
        if ((FR_INV_EXP[i >> 5] >> (i & 31)) & 1) {
            if (started) fr_mul(acc, acc, a);
            else { acc = a; started = true; }
        }
    }
    r = acc;
    return true;
}
// y > (p-1)/2 on the canonical integer ("lexicographically larger", bls_utils.ak:35-43)
H2V_DI bool fp_is_lex_larger(const Fp &a_mont) {
    Fp a;
    FpF::from_mont(a, a_mont);
    // a > half  <=>  half - a borrows
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t d = (uint64_t)FP_HALF_PLAIN[i] - a.v[i] - br;
        br = (d >> 63) & 1;
    }
    return br != 0;
}
