// gfx950 device arithmetic for the BLS12-381 base field Fp (12 x u32) and scalar field Fr (8 x u32).
//
// Montgomery form (R = 2^392 for Fp, 2^256 for Fr), values kept fully reduced after every operation so that
// equality is limb equality.  Fr uses a saturated 32-bit CIOS; Fp uses the 14 x 28-bit product-scanning multiplier
// further down (storage stays 12 x 32).
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
        uint32_t br = 0, bo;
#pragma unroll
        for (int i = 0; i < N; i++) { d[i] = __builtin_subc(a[i], PR::mod(i), br, &bo); br = bo; }
        const bool use = (top != 0) || (br == 0);
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = use ? d[i] : a[i];
    }
    // add / sub as v_add_co / v_addc_co carry chains (the u64-emulated forms cost ~4x the instructions)
    H2V_DI static void add(T &r, const T &a, const T &b) {
        uint32_t s[N], d[N];
        uint32_t c = 0, co;
#pragma unroll
        for (int i = 0; i < N; i++) { s[i] = __builtin_addc(a.v[i], b.v[i], c, &co); c = co; }
        uint32_t br = 0, bo;
#pragma unroll
        for (int i = 0; i < N; i++) { d[i] = __builtin_subc(s[i], PR::mod(i), br, &bo); br = bo; }
        const bool use = (c != 0) || (br == 0);
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = use ? d[i] : s[i];
    }
    H2V_DI static void sub(T &r, const T &a, const T &b) {
        uint32_t d[N];
        uint32_t br = 0, bo;
#pragma unroll
        for (int i = 0; i < N; i++) { d[i] = __builtin_subc(a.v[i], b.v[i], br, &bo); br = bo; }
        const uint32_t mask = 0u - br;
        uint32_t c = 0, co;
#pragma unroll
        for (int i = 0; i < N; i++) { r.v[i] = __builtin_addc(d[i], PR::mod(i) & mask, c, &co); c = co; }
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

// ------------------------------------------------------------------ Fp multiplier: 14 x 28-bit product scanning
// Measured on MI355X (tools/ubench/imad.hip): v_mad_u64_u32 5.4 cycles per wave-instruction, 64-bit add 5.2,
// add-with-carry pairs 5 each, plain 32-bit ALU 2.7.  A saturated 32-bit CIOS row therefore pays as much for the
// carry/add plumbing as for the multiplies.  With 28-bit limbs a whole product-scanning column
//     acc += a_i*b_(k-i)  (k+1 terms)  ;  acc += m_i*p_(k-i)  (k terms)
// fits a 64-bit accumulator (28 terms * 2^56 < 2^61), so every inner step is ONE v_mad_u64_u32 accumulating into
// the same register pair and carries are handled once per column (FIPS Montgomery, R = 2^392).
// Storage stays 12 x 32-bit (cheap add/sub, compact buffers); limbs are re-cut at the multiplier's edges.
#define FP28_MASK 0x0fffffffu
H2V_DI void fp_to28(uint32_t (&o)[14], const Fp &a) {
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const int bit = 28 * i, w = bit >> 5, sh = bit & 31;
        const uint32_t lo = a.v[w];
        const uint32_t hi = (w + 1 < 12) ? a.v[w + 1] : 0u;
        const uint32_t v = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
        o[i] = v & FP28_MASK;
    }
}
H2V_DI void fp_from28(uint32_t (&r)[12], const uint32_t (&t)[14]) {
#pragma unroll
    for (int j = 0; j < 12; j++) {
        const int bit = 32 * j, i = bit / 28, sh = bit % 28;
        uint32_t v = t[i] >> sh;
        if (i + 1 < 14) v |= t[i + 1] << (28 - sh);
        if (i + 2 < 14 && 56 - sh < 32) v |= t[i + 2] << (56 - sh);
        r[j] = v;
    }
}
// t = a*b/2^392 mod p in 28-bit limbs (value < 2p; limbs normalised, top limb may carry the excess)
H2V_DI void fp_mont28(uint32_t (&t)[14], const uint32_t (&a)[14], const uint32_t (&b)[14]) {
    uint32_t m[14];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * FP_MOD28[k - i];
        m[k] = ((uint32_t)acc * FP_N0_28) & FP28_MASK;
        acc += (uint64_t)m[k] * FP_MOD28[0];
        acc >>= 28;
    }
#pragma unroll
    for (int k = 14; k < 27; k++) {
#pragma unroll
        for (int i = k - 13; i < 14; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = k - 13; i < 14; i++) acc += (uint64_t)m[i] * FP_MOD28[k - i];
        t[k - 14] = (uint32_t)acc & FP28_MASK;
        acc >>= 28;
    }
    t[13] = (uint32_t)acc;
}
// a^2: the off-diagonal products are computed once against the doubled operand
H2V_DI void fp_montsqr28(uint32_t (&t)[14], const uint32_t (&a)[14]) {
    uint32_t m[14], d[14];
#pragma unroll
    for (int i = 0; i < 14; i++) d[i] = a[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc += (uint64_t)d[i] * a[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a[k / 2] * a[k / 2];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * FP_MOD28[k - i];
        m[k] = ((uint32_t)acc * FP_N0_28) & FP28_MASK;
        acc += (uint64_t)m[k] * FP_MOD28[0];
        acc >>= 28;
    }
#pragma unroll
    for (int k = 14; k < 27; k++) {
#pragma unroll
        for (int i = k - 13; 2 * i < k; i++) acc += (uint64_t)d[i] * a[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a[k / 2] * a[k / 2];
#pragma unroll
        for (int i = k - 13; i < 14; i++) acc += (uint64_t)m[i] * FP_MOD28[k - i];
        t[k - 14] = (uint32_t)acc & FP28_MASK;
        acc >>= 28;
    }
    t[13] = (uint32_t)acc;
}
H2V_DI Fp fp_mul_inl(const Fp &a, const Fp &b) {
    uint32_t a28[14], b28[14], t[14], w[12];
    fp_to28(a28, a);
    fp_to28(b28, b);
    fp_mont28(t, a28, b28);
    fp_from28(w, t);
    Fp r;
    FpF::cond_sub(r, w, 0);
    return r;
}
H2V_DI Fp fp_sqr_inl(const Fp &a) {
    uint32_t a28[14], t[14], w[12];
    fp_to28(a28, a);
    fp_montsqr28(t, a28);
    fp_from28(w, t);
    Fp r;
    FpF::cond_sub(r, w, 0);
    return r;
}
// ---- out-of-line entry points: keep code size and compile time in check.  Operands and result travel in VGPRs:
// the device ABI passes 16-byte vectors in registers, whereas a second 48-byte struct argument went through
// scratch memory (one of the sources of the MSM kernel's 10 GB of HBM traffic per launch).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct FpRegs { u32x4 a, b, c; };
H2V_DI FpRegs fp_pack(const Fp &x) {
    FpRegs r;
    r.a = u32x4{x.v[0], x.v[1], x.v[2], x.v[3]};
    r.b = u32x4{x.v[4], x.v[5], x.v[6], x.v[7]};
    r.c = u32x4{x.v[8], x.v[9], x.v[10], x.v[11]};
    return r;
}
H2V_DI Fp fp_unpack(const u32x4 a, const u32x4 b, const u32x4 c) {
    Fp x;
    x.v[0] = a.x; x.v[1] = a.y; x.v[2] = a.z; x.v[3] = a.w;
    x.v[4] = b.x; x.v[5] = b.y; x.v[6] = b.z; x.v[7] = b.w;
    x.v[8] = c.x; x.v[9] = c.y; x.v[10] = c.z; x.v[11] = c.w;
    return x;
}
H2V_DN FpRegs fp_mul_raw(u32x4 a0, u32x4 a1, u32x4 a2, u32x4 b0, u32x4 b1, u32x4 b2) {
    return fp_pack(fp_mul_inl(fp_unpack(a0, a1, a2), fp_unpack(b0, b1, b2)));
}
H2V_DN FpRegs fp_sqr_raw(u32x4 a0, u32x4 a1, u32x4 a2) { return fp_pack(fp_sqr_inl(fp_unpack(a0, a1, a2))); }
H2V_DI void fp_mul(Fp &r, const Fp &a, const Fp &b) {
    const FpRegs x = fp_pack(a), y = fp_pack(b);
    const FpRegs z = fp_mul_raw(x.a, x.b, x.c, y.a, y.b, y.c);
    r = fp_unpack(z.a, z.b, z.c);
}
H2V_DI void fp_sqr(Fp &r, const Fp &a) {
    const FpRegs x = fp_pack(a);
    const FpRegs z = fp_sqr_raw(x.a, x.b, x.c);
    r = fp_unpack(z.a, z.b, z.c);
}
H2V_DI void fp_add(Fp &r, const Fp &a, const Fp &b) { FpF::add(r, a, b); }
H2V_DI void fp_sub(Fp &r, const Fp &a, const Fp &b) { FpF::sub(r, a, b); }
H2V_DI void fp_neg(Fp &r, const Fp &a) { FpF::neg(r, a); }
H2V_DI void fp_dbl(Fp &r, const Fp &a) { FpF::add(r, a, a); }
H2V_DI bool fp_is_zero(const Fp &a) { return FpF::is_zero(a); }
H2V_DI bool fp_eq(const Fp &a, const Fp &b) { return FpF::eq(a, b); }
H2V_DI void fp_set_one(Fp &r) { FpF::set_one(r); }
H2V_DI void fp_set_zero(Fp &r) { FpF::set_zero(r); }

// Fr product out of line with operands and result in VGPRs (the by-reference form cost two scratch round trips per
// call, which a lone wave of the combiner kernel cannot hide: 5.2k cycles per multiplying bundle instead of ~2.5k)
struct FrRegs { u32x4 a, b; };
H2V_DN FrRegs fr_mul_raw(u32x4 a0, u32x4 a1, u32x4 b0, u32x4 b1) {
    Fr x, y, z;
    x.v[0] = a0.x; x.v[1] = a0.y; x.v[2] = a0.z; x.v[3] = a0.w; x.v[4] = a1.x; x.v[5] = a1.y; x.v[6] = a1.z; x.v[7] = a1.w;
    y.v[0] = b0.x; y.v[1] = b0.y; y.v[2] = b0.z; y.v[3] = b0.w; y.v[4] = b1.x; y.v[5] = b1.y; y.v[6] = b1.z; y.v[7] = b1.w;
    FrF::mul(z, x, y);
    FrRegs r;
    r.a = u32x4{z.v[0], z.v[1], z.v[2], z.v[3]};
    r.b = u32x4{z.v[4], z.v[5], z.v[6], z.v[7]};
    return r;
}
H2V_DI void fr_mul(Fr &r, const Fr &a, const Fr &b) {
    const FrRegs z = fr_mul_raw(u32x4{a.v[0], a.v[1], a.v[2], a.v[3]}, u32x4{a.v[4], a.v[5], a.v[6], a.v[7]},
                                u32x4{b.v[0], b.v[1], b.v[2], b.v[3]}, u32x4{b.v[4], b.v[5], b.v[6], b.v[7]});
    r.v[0] = z.a.x; r.v[1] = z.a.y; r.v[2] = z.a.z; r.v[3] = z.a.w; r.v[4] = z.b.x; r.v[5] = z.b.y; r.v[6] = z.b.z; r.v[7] = z.b.w;
}
// conversions through the same out-of-line multiplier (the Field<> forms inline a whole product each)
H2V_DI void fr_to_mont(Fr &r, const Fr &plain) {
    Fr k;
#pragma unroll
    for (int i = 0; i < 8; i++) k.v[i] = FR_R2[i];
    fr_mul(r, plain, k);
}
H2V_DI void fr_from_mont(Fr &r, const Fr &a) {
    Fr one;
#pragma unroll
    for (int i = 0; i < 8; i++) one.v[i] = i == 0 ? 1u : 0u;
    fr_mul(r, a, one);
}
H2V_DI void fr_add(Fr &r, const Fr &a, const Fr &b) { FrF::add(r, a, b); }
H2V_DI void fr_sub(Fr &r, const Fr &a, const Fr &b) { FrF::sub(r, a, b); }

H2V_DI void fp_to_mont(Fp &r, const Fp &plain) {
    Fp k;
#pragma unroll
    for (int i = 0; i < 12; i++) k.v[i] = FP_R2[i];
    fp_mul(r, plain, k);
}
H2V_DI void fp_from_mont(Fp &r, const Fp &a) {
    Fp one;
    fp_set_zero(one);
    one.v[0] = 1;
    fp_mul(r, a, one);
}
// ------------------------------------------------------------------ inversion
// Any exact inverse is the same field element as the reference's recip (BlsTypes.hs:201-212 / recip_eea,
// bls_utils.ak:98-117).
#include "h2v_modinv.hpp"
// Inversion: batched division steps (h2v_modinv.hpp).  The operand is a Montgomery residue aR; its integer
// inverse is a^-1 R^-1, and one Montgomery product with R^3 returns a^-1 R.  Returns false when a == 0.
H2V_DN bool fp_inv(Fp &r, const Fp &a) {
    if (fp_is_zero(a)) { fp_set_zero(r); return false; }
    Fp x, c;
    ModInv30<13>::inverse<12>(x.v, a.v, FP_MOD30, FP_MINV30);
#pragma unroll
    for (int i = 0; i < 12; i++) c.v[i] = FP_R3[i];
    fp_mul(r, x, c);
    return true;
}
H2V_DN bool fr_inv(Fr &r, const Fr &a) {
    if (FrF::is_zero(a)) { FrF::set_zero(r); return false; }
    Fr x, c;
    ModInv30<9>::inverse<8>(x.v, a.v, FR_MOD30, FR_MINV30);
#pragma unroll
    for (int i = 0; i < 8; i++) c.v[i] = FR_R3[i];
    fr_mul(r, x, c);
    return true;
}
// y > (p-1)/2 on the canonical integer ("lexicographically larger", bls_utils.ak:35-43)
H2V_DI bool fp_is_lex_larger(const Fp &a_mont) {
    Fp a;
    fp_from_mont(a, a_mont);
    // a > half  <=>  half - a borrows
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t d = (uint64_t)FP_HALF_PLAIN[i] - a.v[i] - br;
        br = (d >> 63) & 1;
    }
    return br != 0;
}
