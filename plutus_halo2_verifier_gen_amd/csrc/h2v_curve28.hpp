// G1 group law (a = 0 short Weierstrass; the constant b never enters the formulas, so the same code serves
// y^2 = x^3 + 4 and the isomorphic curves the subgroup test works on) over the lazily reduced field of h2v_fp28.hpp.
//
// The point at infinity is an explicit flag (`inf`) beside the coordinates: with lazily reduced coordinates "Z == 0"
// is not a limb pattern.  Completeness is kept exactly as in h2v_curve.hpp (P+P, P+(-P), infinity operands): proof
// bytes are adversarial and the verdict must match the reference's group law
// (aiken-verifier/aiken_halo2/lib/bls_utils.ak:77-86, the Plutus builtins bls12_381_G1_add / scalarMul).
//
// Coordinate bounds (v, lam) of every stored point, see h2v_fp28.hpp for the notation:
//     X (31, 1)   Y (20, 1)   Z (4, 2)           -- closed under g1j28_dbl and g1j28_add (derivations inline);
//     table entries (never negated in place) have Y (19, 1)
#pragma once
#include "h2v_curve.hpp"
#include "h2v_fp28.hpp"

struct G1J28 { F28 x, y, z; };

// affine canonical point -> Jacobian (1,1),(1,1),(1,1)
H2V_DI void g1j28_from_affine(G1J28 &r, const G1A &a) {
    f28_from_fp(r.x, a.x);
    f28_from_fp(r.y, a.y);
    f28_set_one(r.z);
}
// -> canonical Jacobian coordinates of h2v_curve.hpp
H2V_DI void g1j28_to_g1j(G1J &r, const G1J28 &p, const bool inf) {
    if (inf) { g1j_set_inf(r); return; }
    F28 z = p.z;
    f28_carry(z);
    f28_to_fp(r.x, p.x);
    f28_to_fp(r.y, p.y);
    f28_to_fp(r.z, z);
}
// dbl-2009-l, 2M + 5S.  p must not be infinity and must not have y == 0 (no 2-torsion on these curves: the group
// order is odd).  In: X (<=43, 1), Y (<=45, 1), Z (v_Y v_Z <= 2048, lam <= 2).  Out: X (31,1) Y (19,1) Z (4,2).
#define F28_MUL_(r, a, b) do { if (INL) f28_mul_inl(r, a, b); else f28_mul(r, a, b); } while (0)
#define F28_SQR_(r, a) do { if (INL) f28_sqr_inl(r, a); else f28_sqr(r, a); } while (0)
template <bool INL>
H2V_DI void g1j28_dbl_t(G1J28 &r, const G1J28 &p) {
    F28 X = p.x, Y = p.y, Z = p.z, A, B, C, D, t;
    F28_MUL_(Z, Y, Z); f28_mul_small<2>(Z, Z);        // Z3 = 2 Y Z                      (4, 2)
    F28_SQR_(A, X);                                   // A = X^2                          (2, 1)
    F28_SQR_(B, Y);                                   // B = Y^2                          (2, 1)
    f28_add(t, X, B); F28_SQR_(t, t);                 // (X + B)^2, operand (v_X+2, 2)    (2, 1)
    F28_SQR_(C, B);                                   // C = B^2                          (2, 1)
    f28_add(D, A, C);                                //                                  (4, 2)
    F28_SUB(t, t, D, 5, 2);                          // (X+B)^2 - A - C                  (7, 5)
    f28_mul_small<2>(D, t); f28_carry(D);            // D                                (14, 1)
    f28_mul_small<3>(A, A);                          // E = 3A                           (6, 3)
    F28_SQR_(X, A);                                   // E^2   lam 9, v 36                (2, 1)
    f28_mul_small<2>(t, D);                          // 2D                               (28, 2)
    F28_SUB(X, X, t, 29, 2); f28_carry(X);           // X3 = E^2 - 2D                    (31, 1)
    F28_SUB(t, D, X, 32, 1);                         // D - X3                           (46, 4)
    F28_MUL_(Y, A, t);                                // E (D - X3)  lam 12, v 276        (2, 1)
    f28_mul_small<8>(C, C);                          // 8C                               (16, 8)
    F28_SUB(Y, Y, C, 17, 8); f28_carry(Y);           // Y3                               (19, 1)
    r.x = X; r.y = Y; r.z = Z;
}
#undef F28_MUL_
#undef F28_SQR_
H2V_DI void g1j28_dbl(G1J28 &r, const G1J28 &p) { g1j28_dbl_t<false>(r, p); }
// r = p + (neg_q ? -q : q), 12M + 4S; neither operand is infinity (the callers keep the flags).
// Returns 0: generic sum in r; 1: p == +-q with equal y -> r = 2p; 2: p == -(+-q) -> the sum is infinity (r untouched).
// In: both operands with the stored-point bounds.  Out: X (10,1) Y (5,1) Z (2,1).
H2V_DI int g1j28_add(G1J28 &r, const G1J28 &p, const G1J28 &q, const bool neg_q) {
    F28 X1 = p.x, Y1 = p.y, Z1 = p.z, a, b, c, t;
    f28_sqr(a, q.z);                                 // Z2^2       lam 4, v 16           (2, 1)
    f28_mul(X1, X1, a);                              // U1         v 62                  (2, 1)
    f28_mul(t, q.z, a); f28_mul(Y1, Y1, t);          // S1         v 8 ; 38              (2, 1)
    f28_sqr(a, Z1);                                  // Z1^2                             (2, 1)
    f28_mul(b, q.x, a);                              // U2                               (2, 1)
    f28_mul(t, Z1, a);                               // Z1^3                             (2, 1)
    {
        F28 qy = q.y, nq;
        F28_NEG(nq, qy, 20, 1);                      // -Y2                              (20, 3)
        if (neg_q) qy = nq;
        f28_mul(c, qy, t);                           // S2         lam 3, v 40           (2, 1)
    }
    F28_SUB(b, b, X1, 3, 1); f28_carry(b);           // H = U2 - U1                      (5, 1)
    F28_SUB(c, c, Y1, 3, 1); f28_carry(c);           // R = S2 - S1                      (5, 1)
    if (f28_is_zero_v5(b)) {
        if (f28_is_zero_v5(c)) { g1j28_dbl(r, p); return 1; }
        return 2;
    }
    f28_mul(Z1, Z1, q.z); f28_mul(Z1, Z1, b);        // Z3         lam 4, v 16 ; v 10    (2, 1)
    f28_sqr(a, b);                                   // HH                               (2, 1)
    f28_mul(b, b, a);                                // HHH                              (2, 1)
    f28_mul(a, X1, a);                               // V = U1 HH                        (2, 1)
    f28_sqr(X1, c);                                  // R^2                              (2, 1)
    F28_SUB(X1, X1, b, 3, 1);                        // R^2 - HHH                        (5, 4)
    f28_mul_small<2>(t, a);                          // 2V                               (4, 2)
    F28_SUB(X1, X1, t, 5, 2); f28_carry(X1);         // X3                               (10, 1)
    F28_SUB(t, a, X1, 11, 1);                        // V - X3                           (13, 4)
    f28_mul(c, c, t);                                // R (V - X3)   lam 4, v 65         (2, 1)
    f28_mul(t, Y1, b);                               // S1 HHH                           (2, 1)
    F28_SUB(Y1, c, t, 3, 1); f28_carry(Y1);          // Y3                               (5, 1)
    r.x = X1; r.y = Y1; r.z = Z1;
    return 0;
}
// Mixed addition r = p + (neg_q ? -q : q) with q AFFINE (qx, qy carried, v <= 2): 8M + 3S.  NO exceptional cases:
// the caller guarantees p != +-q and both finite.  That holds in the MSM ladder by construction - p = [a]P with
// a = 16 * (a non-zero prefix of a scalar below 2^128), q = [d]P with 1 <= d <= 8, and P of prime order r > 2^254, so
// a +- d is never 0 mod r - and nowhere else is this function used.
// In: p with the stored-point bounds.  Out: X (10,1) Y (5,1) Z (2,1).
#define F28_MUL_(r, a, b) do { if (INL) f28_mul_inl(r, a, b); else f28_mul(r, a, b); } while (0)
#define F28_SQR_(r, a) do { if (INL) f28_sqr_inl(r, a); else f28_sqr(r, a); } while (0)
template <bool INL>
H2V_DI void g1j28_madd_ladder_t(G1J28 &r, const G1J28 &p, const F28 &qx, const F28 &qy_in, const bool neg_q) {
    F28 X1 = p.x, Y1 = p.y, Z1 = p.z, a, b, c, t;
    F28_SQR_(a, Z1);                                  // Z1^2        lam 4, v 16          (2, 1)
    F28_MUL_(b, qx, a);                               // U2                               (2, 1)
    F28_MUL_(t, Z1, a);                               // Z1^3                             (2, 1)
    {
        F28 qy = qy_in, nq;
        F28_NEG(nq, qy, 3, 1);                       // -Y2                              (3, 3)
        if (neg_q) qy = nq;
        F28_MUL_(c, qy, t);                           // S2          lam 3, v 6           (2, 1)
    }
    F28_SUB(b, b, X1, 32, 1);                        // H = U2 - X1                      (34, 4)
    F28_SUB(c, c, Y1, 21, 1);                        // R = S2 - Y1                      (23, 4)
    F28_MUL_(Z1, Z1, b);                              // Z3 = Z1 H   lam 8, v 136         (2, 1)
    F28_SQR_(a, b);                                   // HH          lam 16, v 1156       (2, 1)
    F28_MUL_(b, b, a);                                // HHH         lam 4, v 68          (2, 1)
    F28_MUL_(a, X1, a);                               // V = X1 HH   v 62                 (2, 1)
    F28_SQR_(X1, c);                                  // R^2         lam 16, v 529        (2, 1)
    F28_SUB(X1, X1, b, 3, 1);                        // R^2 - HHH                        (5, 4)
    f28_mul_small<2>(t, a);                          // 2V                               (4, 2)
    F28_SUB(X1, X1, t, 5, 2); f28_carry(X1);         // X3                               (10, 1)
    F28_SUB(t, a, X1, 11, 1);                        // V - X3                           (13, 4)
    F28_MUL_(c, c, t);                                // R (V - X3)  lam 16, v 299        (2, 1)
    F28_MUL_(t, Y1, b);                               // Y1 HHH      v 40                 (2, 1)
    F28_SUB(Y1, c, t, 3, 1); f28_carry(Y1);          // Y3                               (5, 1)
    r.x = X1; r.y = Y1; r.z = Z1;
}
#undef F28_MUL_
#undef F28_SQR_
H2V_DI void g1j28_madd_ladder(G1J28 &r, const G1J28 &p, const F28 &qx, const F28 &qy_in, const bool neg_q) {
    g1j28_madd_ladder_t<false>(r, p, qx, qy_in, neg_q);
}
template <int LANE>
H2V_DI F28 f28_bcast(const F28 &a) {   // lane LANE (0..3) of every quad in all four lanes of the quad: one DPP move per limb
    F28 r;
    constexpr int ctrl = LANE | (LANE << 2) | (LANE << 4) | (LANE << 6);   // quad_perm: [LANE, LANE, LANE, LANE]
#pragma unroll
    for (int i = 0; i < 14; i++) {
        r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.l[i], ctrl, 0xf, 0xf, false);
        asm volatile("" : "+v"(r.l[i]));   // keep the move a move: see the note at g1j28_madd_quad
    }
    return r;
}
// QUAD-COOPERATIVE POINT ARITHMETIC.  The four lanes of a hardware quad (lane & 3) hold the same point and run the
// independent field multiplications of a formula side by side, one each; values cross lanes with quad-broadcast DPP
// moves (VALU rate).  Every lane of a quad must take the same path (the callers' control flow is uniform per quad).
//
// n doublings of the point lane 0 of each quad holds, spread over its lanes 0..2 (the result is valid in all four).  dbl-2009-l has 7 multiplications but depth 3:
//     level 1   lane 0: A = X^2        lane 1: B = Y^2        lane 2: Y Z
//     level 2   lane 0: (3A)^2         lane 1: C = B^2        lane 2: (X + B)^2
//     level 3   E (D - X3)             (operands are wave-uniform by then: every lane computes it)
// Users: the window weights 2^(c w) of k_pip_reduce (a chain of up to 120 doublings on an otherwise idle wave: 0.9 ms on
// one lane) and the four-lanes-per-half shape of the per-proof MSM (msm_body<8>).  Values cross lanes with quad-broadcast DPP moves.  Bounds: those of g1j28_dbl_t, line by line.
H2V_DN void g1j28_dbl_n_coop3(G1J28 &p, const uint32_t n) {
    const int l = threadIdx.x & 3;
    F28 X = f28_bcast<0>(p.x), Y = f28_bcast<0>(p.y), Z = f28_bcast<0>(p.z);
#pragma unroll 1
    for (uint32_t k = 0; k < n; k++) {
        F28 a, b, r1, r2, e3, xb, o2;
#pragma unroll
        for (int i = 0; i < 14; i++) { a.l[i] = l == 0 ? X.l[i] : Y.l[i]; b.l[i] = l == 0 ? X.l[i] : (l == 1 ? Y.l[i] : Z.l[i]); }
        f28_mul_inl(r1, a, b);                            // A | B | Y Z                       (2, 1)
        const F28 B = f28_bcast<1>(r1);
        f28_mul_small<3>(e3, r1);                         // lane 0: E = 3A                    (6, 3)
        f28_add(xb, X, B);                                // X + B                             (v_X + 2, 2)
#pragma unroll
        for (int i = 0; i < 14; i++) o2.l[i] = l == 0 ? e3.l[i] : (l == 1 ? B.l[i] : xb.l[i]);
        f28_mul_inl(r2, o2, o2);                          // E^2 | C = B^2 | (X + B)^2         (2, 1)
        const F28 A = f28_bcast<0>(r1), YZ = f28_bcast<2>(r1), E2 = f28_bcast<0>(r2), T = f28_bcast<2>(r2);
        F28 C = f28_bcast<1>(r2), D, t, E, X3, Y3;
        f28_add(D, A, C);                                 //                                   (4, 2)
        F28_SUB(t, T, D, 5, 2);                           // (X+B)^2 - A - C                   (7, 5)
        f28_mul_small<2>(D, t); f28_carry(D);             // D                                 (14, 1)
        f28_mul_small<3>(E, A);                           // E = 3A                            (6, 3)
        f28_mul_small<2>(t, D);                           // 2D                                (28, 2)
        F28_SUB(X3, E2, t, 29, 2); f28_carry(X3);         // X3 = E^2 - 2D                     (31, 1)
        F28_SUB(t, D, X3, 32, 1);                         // D - X3                            (46, 4)
        f28_mul_inl(Y3, E, t);                            // E (D - X3)                        (2, 1)
        f28_mul_small<8>(C, C);                           // 8C                                (16, 8)
        F28_SUB(Y3, Y3, C, 17, 8); f28_carry(Y3);         // Y3                                (19, 1)
        f28_mul_small<2>(Z, YZ);                          // Z3 = 2 Y Z                        (4, 2)
        X = X3; Y = Y3;
    }
    p.x = X; p.y = Y; p.z = Z;
}

// Mixed addition p += (neg_q ? -q : q), q affine, for the quad: the 11 multiplications of g1j28_madd_ladder_t in 6 levels
//     level 1   Z1^2                                       (every lane)
//     level 2   lane 0: U2 = x2 Z1^2        lane 1: Z1^3
//     level 3   lane 0: S2 = y2 Z1^3        lane 1: Z3 = Z1 H        lane 2: HH = H^2
//     level 4   lane 0: HHH = H HH          lane 1: V = X1 HH        lane 2: R^2
//     level 5   lane 0: R (V - X3)          lane 1: Y1 HHH
// Same caller guarantee (p != +-q, both finite), same bounds line by line: in: stored-point bounds, out: X (10,1) Y (5,1) Z (2,1).
// (Compiler note, ROCm 7.2: with the broadcasts left to the optimiser it folds them into the consuming subtraction -
// `v_subrev_u32_dpp vN, vN, vM quad_perm:[1,1,1,1]`, destination = permuted source - and Y3 came out wrong in lane 0 of
// every quad, right in lanes 1..3 (h2v_probe_quad_madd; tests/test_gpu_parity.py); f28_bcast therefore pins its result with an empty asm.)
H2V_DN void g1j28_madd_quad(G1J28 &p, const F28 &qx, const F28 &qy_in, const bool neg_q) {
    const int l = threadIdx.x & 3;
    const F28 X1 = p.x, Y1 = p.y, Z1 = p.z;
    F28 a, x, y, r, qy = qy_in;
    {
        F28 nq;
        F28_NEG(nq, qy, 3, 1);                            // -Y2                              (3, 3)
        if (neg_q) qy = nq;
    }
    f28_sqr_inl(a, Z1);                                   // Z1^2                             (2, 1)
#pragma unroll
    for (int i = 0; i < 14; i++) x.l[i] = l == 1 ? Z1.l[i] : qx.l[i];
    f28_mul_inl(r, x, a);                                 // U2 | Z1^3                        (2, 1)
    const F28 U2 = f28_bcast<0>(r), Z1c = f28_bcast<1>(r);
    F28 H, R, t;
    F28_SUB(H, U2, X1, 32, 1);                            // H = U2 - X1                      (34, 4)
#pragma unroll
    for (int i = 0; i < 14; i++) { x.l[i] = l == 0 ? qy.l[i] : (l == 1 ? Z1.l[i] : H.l[i]); y.l[i] = l == 0 ? Z1c.l[i] : H.l[i]; }
    f28_mul_inl(r, x, y);                                 // S2 | Z3 | HH                     (2, 1)
    const F28 S2 = f28_bcast<0>(r), Z3 = f28_bcast<1>(r), HH = f28_bcast<2>(r);
    F28_SUB(R, S2, Y1, 21, 1);                            // R = S2 - Y1                      (23, 4)
#pragma unroll
    for (int i = 0; i < 14; i++) { x.l[i] = l == 0 ? H.l[i] : (l == 1 ? X1.l[i] : R.l[i]); y.l[i] = l == 2 ? R.l[i] : HH.l[i]; }
    f28_mul_inl(r, x, y);                                 // HHH | V | R^2                    (2, 1)
    const F28 HHH = f28_bcast<0>(r), V = f28_bcast<1>(r), RR = f28_bcast<2>(r);
    F28 X3, Y3;
    F28_SUB(X3, RR, HHH, 3, 1);                           // R^2 - HHH                        (5, 4)
    f28_mul_small<2>(t, V);                               // 2V                               (4, 2)
    F28_SUB(X3, X3, t, 5, 2); f28_carry(X3);              // X3                               (10, 1)
    F28_SUB(t, V, X3, 11, 1);                             // V - X3                           (13, 4)
#pragma unroll
    for (int i = 0; i < 14; i++) { x.l[i] = l == 0 ? R.l[i] : Y1.l[i]; y.l[i] = l == 0 ? t.l[i] : HHH.l[i]; }
    f28_mul_inl(r, x, y);                                 // R (V - X3) | Y1 HHH              (2, 1)
    const F28 c = f28_bcast<0>(r), yh = f28_bcast<1>(r);
    F28_SUB(Y3, c, yh, 3, 1); f28_carry(Y3);              // Y3                               (5, 1)
    p.x = X3; p.y = Y3; p.z = Z3;
}

// Brings n <= 8 finite Jacobian points to affine with ONE inversion (Montgomery's trick); x and y come out carried
// with v <= 2.  ax / ay may alias nothing in `pts`.
// (inlined: the arrays stay the caller's own stack objects, addressed with scratch instructions)
template <int N>
H2V_DI void g1j28_batch_to_affine(F28 (&ax)[N], F28 (&ay)[N], const G1J28 (&pts)[N]) {
    F28 pre[N];
    pre[0] = pts[0].z;
    f28_carry(pre[0]);
#pragma unroll 1
    for (int i = 1; i < N; i++) f28_mul(pre[i], pre[i - 1], pts[i].z);      // lam 1 x 2
    Fp pc, ic;
    f28_to_fp(pc, pre[N - 1]);
    (void)fp_inv(ic, pc);                              // Z of a finite point is never 0 mod p
    F28 inv;
    f28_from_fp(inv, ic);
#pragma unroll 1
    for (int i = N - 1; i >= 0; i--) {
        F28 zi, z2, t;
        if (i > 0) { f28_mul(zi, inv, pre[i - 1]); f28_mul(t, inv, pts[i].z); inv = t; } else zi = inv;
        f28_sqr(z2, zi);
        f28_mul(ax[i], pts[i].x, z2);
        f28_mul(z2, z2, zi);
        f28_mul(ay[i], pts[i].y, z2);
    }
}
// out-of-line forms for the cold paths (window-table construction)
H2V_DN void g1j28_dbl_ool(G1J28 &r, const G1J28 &p) { g1j28_dbl(r, p); }
H2V_DN int g1j28_add_ool(G1J28 &r, const G1J28 &p, const G1J28 &q) { return g1j28_add(r, p, q, false); }
H2V_DI void g1_store_table_entry(uint32_t *dst, const F28 &x, const F28 &y) {
    uint4 *q = reinterpret_cast<uint4 *>(dst);
    q[0] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
    q[1] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
    q[2] = make_uint4(x.l[8], x.l[9], x.l[10], x.l[11]);
    q[3] = make_uint4(x.l[12], x.l[13], y.l[0], y.l[1]);
    q[4] = make_uint4(y.l[2], y.l[3], y.l[4], y.l[5]);
    q[5] = make_uint4(y.l[6], y.l[7], y.l[8], y.l[9]);
    q[6] = make_uint4(y.l[10], y.l[11], y.l[12], y.l[13]);
}
// Window table of a base point for the MSM ladder: tab[m-1] = m * P, m = 1..8, AFFINE (x, y: 2 x 14 carried limbs,
// 28 dwords per entry, 224 per table).  4 doublings + 3 additions, then one inversion for the seven multiples
// (Montgomery's trick).  `base` is a finite curve point; for a point of G1 every multiple is finite.  (For a curve
// point outside G1 - the decompression kernel builds tables before the subgroup verdict is known - a multiple may be
// the point at infinity: the inversion then yields zeros, nothing faults, and the slot is marked invalid anyway.)
// Montgomery's trick for the seven multiples of a window table, with the running products parked in the table itself (the
// first 14 dwords of the entry they are needed for; every entry is rewritten with its affine point afterwards) and the affine
// coordinates stored as they come: no per-lane arrays of prefix products and results - 1.2 KB less private memory per lane
// in every kernel that builds tables, which is what the runtime sizes a queue's scratch arena by.
template <int N>
H2V_DI void g1j28_table_to_affine(uint32_t *tab /* entry m at tab + 28 m; entry 0 is the base point */, const G1J28 (&pts)[N]) {
    F28 run = pts[0].z;
    f28_carry(run);
#pragma unroll 1
    for (int i = 1; i < N; i++) {
        uint4 *q = reinterpret_cast<uint4 *>(tab + i * 28);      // run = z_0 .. z_(i-1): what entry i's step reads back
        q[0] = make_uint4(run.l[0], run.l[1], run.l[2], run.l[3]); q[1] = make_uint4(run.l[4], run.l[5], run.l[6], run.l[7]);
        q[2] = make_uint4(run.l[8], run.l[9], run.l[10], run.l[11]); q[3] = make_uint4(run.l[12], run.l[13], 0u, 0u);
        f28_mul(run, run, pts[i].z);                             // lam 1 x 2
    }
    Fp pc, ic;
    f28_to_fp(pc, run);
    (void)fp_inv(ic, pc);                              // Z of a finite point is never 0 mod p
    F28 inv;
    f28_from_fp(inv, ic);
#pragma unroll 1
    for (int i = N - 1; i >= 0; i--) {
        F28 zi, z2, t, x, y;
        if (i > 0) {
            const uint4 *q = reinterpret_cast<const uint4 *>(tab + i * 28);
            const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
            F28 pre;
            pre.l[0] = a.x; pre.l[1] = a.y; pre.l[2] = a.z; pre.l[3] = a.w; pre.l[4] = b.x; pre.l[5] = b.y; pre.l[6] = b.z; pre.l[7] = b.w;
            pre.l[8] = c.x; pre.l[9] = c.y; pre.l[10] = c.z; pre.l[11] = c.w; pre.l[12] = d.x; pre.l[13] = d.y;
            f28_mul(zi, inv, pre);
            f28_mul(t, inv, pts[i].z);
            inv = t;
        } else {
            zi = inv;
        }
        f28_sqr(z2, zi);
        f28_mul(x, pts[i].x, z2);
        f28_mul(z2, z2, zi);
        f28_mul(y, pts[i].y, z2);
        g1_store_table_entry(tab + (i + 1) * 28, x, y);
    }
}
H2V_DN void g1_build_window_table(uint32_t *tab, const G1A &base) {
    G1J28 t1, e[7];
    g1j28_from_affine(t1, base);
    // odd multiples by mixed additions of P itself (affine): never exceptional for a point of G1 (m P = +-P only for
    // m = 0, 2 mod r); for a curve point outside G1 the entries may be garbage, which nobody reads (see above)
    g1j28_dbl(e[0], t1);
    g1j28_madd_ladder(e[1], e[0], t1.x, t1.y, false);
    g1j28_dbl(e[2], e[0]);
    g1j28_madd_ladder(e[3], e[2], t1.x, t1.y, false);
    g1j28_dbl(e[4], e[1]);
    g1j28_madd_ladder(e[5], e[4], t1.x, t1.y, false);
    g1j28_dbl(e[6], e[2]);
    // 112-byte entries written as seven 16-byte stores (a table is 16-byte aligned: 896-byte stride)
    g1_store_table_entry(tab, t1.x, t1.y);
    g1j28_table_to_affine<7>(tab, e);
}
// both tables of a point: [0] for P, [1] for phi(P) = (beta' x, y) (the GLV halves of the MSM)
H2V_DN void g1_build_window_tables_glv(uint32_t *tab2, const G1A &base) {
    g1_build_window_table(tab2, base);
    Fp beta;
#pragma unroll
    for (int k = 0; k < 12; k++) beta.v[k] = FP_BETA_GLV[k];
    F28 b28;
    f28_from_fp(b28, beta);
    // phi commutes with scalar multiplication: the second table is the first with every x multiplied by beta'
#pragma unroll 1
    for (int m = 0; m < 8; m++) {
        F28 x, y, bx;
#pragma unroll
        for (int k = 0; k < 14; k++) { x.l[k] = tab2[m * 28 + k]; y.l[k] = tab2[m * 28 + 14 + k]; }
        f28_mul(bx, x, b28);
        g1_store_table_entry(tab2 + 224 + m * 28, bx, y);
    }
}
// acc (+flag) += (neg ? -q : q), q finite
H2V_DI void g1j28_acc_add(G1J28 &acc, bool &acc_inf, const G1J28 &q, const bool neg) {
    if (acc_inf) {
        acc = q;
        if (neg) { F28 ny; F28_NEG(ny, q.y, 20, 1); f28_carry(ny); acc.y = ny; }   // (20, 1)
        acc_inf = false;
        return;
    }
    const int k = g1j28_add(acc, acc, q, neg);
    if (k == 2) acc_inf = true;
}

// [|x|]P, |x| = 0xd201000000010000, complete (P of any order on an a = 0 curve of odd order)
H2V_DN void g1j28_mul_x_abs(G1J28 &r, bool &r_inf, const G1J28 &p, const bool p_inf) {
    G1J28 acc = p;
    bool inf = p_inf;
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
        if (!inf) g1j28_dbl_t<true>(acc, acc);   // the chain's 63 doublings with the multiplier inlined
        if (((BLS_X_ABS >> i) & 1) && !p_inf) g1j28_acc_add(acc, inf, p, false);
    }
    r = acc;
    r_inf = inf;
}
// Is the finite affine point of an a = 0 curve in the r-torsion?  sigma(P) = (beta x, y) == [-x^2]P (forces
// (sigma^2 + sigma + 1)P = [r]P = O; r does not divide the cofactor, so the test is exact); the two 63-step chains run on the lazily reduced field, the final comparison on canonical limbs.
H2V_DN bool g1a_in_subgroup28(const G1A &a) {
    G1J28 p, t;
    bool tinf = false;
    g1j28_from_affine(p, a);
    g1j28_mul_x_abs(t, tinf, p, false);
    g1j28_mul_x_abs(t, tinf, t, tinf);   // [x^2]P
    if (tinf) return false;
    G1J tj;
    g1j28_to_g1j(tj, t, false);
    Fp beta, z2, z3, l, ny;
#pragma unroll
    for (int i = 0; i < 12; i++) beta.v[i] = FP_BETA[i];
    fp_sqr(z2, tj.z);
    fp_mul(z3, z2, tj.z);
    fp_mul(l, a.x, beta); fp_mul(l, l, z2);
    if (!fp_eq(l, tj.x)) return false;
    fp_mul(l, a.y, z3);
    fp_neg(ny, tj.y);
    return fp_eq(l, ny);
}
