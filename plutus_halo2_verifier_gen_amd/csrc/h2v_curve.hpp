// G1 group law for BLS12-381 (y^2 = x^3 + 4) on gfx950: Jacobian coordinates, Z == 0 <=> infinity.
// Affine points travel as (x, y) Montgomery limbs with (0, 0) standing for the point at infinity
// ((0,0) is not on the curve, so the sentinel is unambiguous).
// Complete behaviour (P+P, P+(-P), infinity operands) is implemented explicitly: proof bytes are adversarial
// input and the accept/reject result must match the reference's exact group law (the Plutus builtins
// bls12_381_G1_add / scalarMul used by aiken-verifier/aiken_halo2/lib/bls_utils.ak:77-86).
#pragma once
#include "h2v_field.hpp"

struct G1A { Fp x, y; };       // affine; (0,0) = infinity
struct G1J { Fp x, y, z; };    // Jacobian

H2V_DI bool g1a_is_inf(const G1A &p) { return fp_is_zero(p.x) && fp_is_zero(p.y); }
H2V_DI void g1a_set_inf(G1A &p) { fp_set_zero(p.x); fp_set_zero(p.y); }
H2V_DI bool g1j_is_inf(const G1J &p) { return fp_is_zero(p.z); }
H2V_DI void g1j_set_inf(G1J &p) { fp_set_one(p.x); fp_set_one(p.y); fp_set_zero(p.z); }
H2V_DI void g1j_from_affine(G1J &r, const G1A &a) {
    if (g1a_is_inf(a)) { g1j_set_inf(r); return; }
    r.x = a.x; r.y = a.y; fp_set_one(r.z);
}
// dbl-2009-l (a = 0): 2M + 5S.  Ordered so that at most five field elements are live at any call (the MSM ladder keeps
// the accumulator in VGPRs across the out-of-line multiplier calls; a wider live set spills to scratch = HBM traffic).
H2V_DI void g1j_dbl_inl(G1J &r, const G1J &p) {
    if (g1j_is_inf(p) || fp_is_zero(p.y)) { g1j_set_inf(r); return; }
    Fp X = p.x, Y = p.y, Z = p.z, A, B, C, D, t;
    fp_mul(Z, Y, Z); fp_dbl(Z, Z);                 // Z3 = 2 Y Z
    fp_sqr(A, X);
    fp_sqr(B, Y);
    fp_add(t, X, B); fp_sqr(t, t);
    fp_sqr(C, B);
    fp_sub(t, t, A); fp_sub(t, t, C); fp_dbl(D, t);  // D = 2((X+B)^2 - A - C)
    fp_dbl(t, A); fp_add(A, t, A);                 // E = 3A (in A)
    fp_sqr(X, A); fp_dbl(t, D); fp_sub(X, X, t);   // X3 = E^2 - 2D
    fp_sub(t, D, X); fp_mul(Y, A, t);              // E (D - X3)
    fp_dbl(C, C); fp_dbl(C, C); fp_dbl(C, C);
    fp_sub(Y, Y, C);                               // - 8C
    r.x = X; r.y = Y; r.z = Z;
}
H2V_DN void g1j_dbl(G1J &r, const G1J &p) { g1j_dbl_inl(r, p); }
// full addition: 12M + 4S on the generic path.  q is read through the reference where it is needed (it lives in the
// ladder's table in private memory) and negated on the fly when neg_q; at most six field elements are live at a call.
H2V_DI void g1j_add_signed_inl(G1J &r, const G1J &p, const G1J &q, const bool neg_q) {
    if (g1j_is_inf(q)) { r = p; return; }
    if (g1j_is_inf(p)) { r = q; if (neg_q) fp_neg(r.y, r.y); return; }
    Fp X1 = p.x, Y1 = p.y, Z1 = p.z, a, b, c, t;
    fp_sqr(a, q.z);                                  // z2z2
    fp_mul(X1, X1, a);                               // u1
    fp_mul(t, q.z, a); fp_mul(Y1, Y1, t);            // s1
    fp_sqr(a, Z1);                                   // z1z1
    fp_mul(b, q.x, a);                               // u2
    fp_mul(t, Z1, a);
    { Fp qy = q.y; if (neg_q) fp_neg(qy, qy); fp_mul(c, qy, t); }  // s2
    fp_sub(b, b, X1);                                // h
    fp_sub(c, c, Y1);                                // rr
    if (fp_is_zero(b)) {
        if (fp_is_zero(c)) { g1j_dbl_inl(r, p); return; }
        g1j_set_inf(r); return;
    }
    fp_mul(Z1, Z1, q.z); fp_mul(Z1, Z1, b);          // Z3
    fp_sqr(a, b);                                    // hh
    fp_mul(b, b, a);                                 // hhh
    fp_mul(a, X1, a);                                // v
    fp_sqr(X1, c); fp_sub(X1, X1, b); fp_dbl(t, a); fp_sub(X1, X1, t);   // X3
    fp_sub(t, a, X1); fp_mul(c, c, t);               // rr (v - X3)
    fp_mul(t, Y1, b); fp_sub(Y1, c, t);              // Y3
    r.x = X1; r.y = Y1; r.z = Z1;
}
H2V_DI void g1j_add_inl(G1J &r, const G1J &p, const G1J &q) { g1j_add_signed_inl(r, p, q, false); }
H2V_DN void g1j_add(G1J &r, const G1J &p, const G1J &q) { g1j_add_inl(r, p, q); }
H2V_DN void g1j_to_affine(G1A &r, const G1J &p) {
    if (g1j_is_inf(p)) { g1a_set_inf(r); return; }
    Fp zi, zi2;
    fp_inv(zi, p.z);
    fp_sqr(zi2, zi);
    fp_mul(r.x, p.x, zi2);
    fp_mul(zi2, zi2, zi);
    fp_mul(r.y, p.y, zi2);
}
// ------------------------------------------------------------------ GLV scalar split
// phi(x, y) = (beta' x, y) is [lambda] on G1 with lambda = x^2 - 1 ~ sqrt(r), so an Fr scalar k < r splits by plain
// division k = k2*lambda + k1 into two halves below 2^128: [k]P = [k1]P + [k2]phi(P).  Barrett estimate
// q = floor(k * floor(2^383/lambda) / 2^383), then at most two corrections (model: bls12_381.py: glv_split).
H2V_DI void glv_split(uint32_t (&k1)[4], uint32_t (&k2)[4], const uint32_t (&k)[8]) {
    // prod = k * MU (16 words), column-wise with a 64-bit accumulator + overflow word
    uint32_t prod[16];
    uint64_t acc = 0;
    uint32_t over = 0;
#pragma unroll
    for (int c = 0; c < 15; c++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int j = c - i;
            if (j < 0 || j > 7) continue;
            const uint64_t t = (uint64_t)k[i] * GLV_MU[j];
            acc += t;
            over += acc < t ? 1u : 0u;
        }
        prod[c] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)over << 32);
        over = 0;
    }
    prod[15] = (uint32_t)acc;
    // q = prod >> 383  (word 11, bit 31)
    uint32_t q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = (prod[11 + i] >> 31) | (prod[12 + i] << 1);
    // rem = k - q*lambda  (fits 5 words: rem < 3*lambda)
    uint32_t ql[8];
    acc = 0;
    over = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int j = c - i;
            if (j < 0 || j > 3) continue;
            const uint64_t t = (uint64_t)q[i] * GLV_LAMBDA[j];
            acc += t;
            over += acc < t ? 1u : 0u;
        }
        ql[c] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)over << 32);
        over = 0;
    }
    uint32_t rem[5];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const uint64_t d = (uint64_t)k[i] - ql[i] - br;
        rem[i] = (uint32_t)d;
        br = (d >> 63) & 1;
    }
#pragma unroll 1
    for (int it = 0; it < 2; it++) {
        // rem >= lambda ?
        uint32_t d[5];
        uint64_t b2 = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const uint64_t t = (uint64_t)rem[i] - (i < 4 ? GLV_LAMBDA[i] : 0u) - b2;
            d[i] = (uint32_t)t;
            b2 = (t >> 63) & 1;
        }
        if (b2 == 0) {
#pragma unroll
            for (int i = 0; i < 5; i++) rem[i] = d[i];
            uint64_t c = 1;
#pragma unroll
            for (int i = 0; i < 4; i++) { c += q[i]; q[i] = (uint32_t)c; c >>= 32; }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { k1[i] = rem[i]; k2[i] = q[i]; }
}
