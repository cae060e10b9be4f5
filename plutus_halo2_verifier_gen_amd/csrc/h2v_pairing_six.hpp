// Pairing check, SIX LANES PER PROOF (ten proofs per wave): the engine for callers that keep the chip full anyway.
//
// Why: with the chip full the pairing kernel is bound by the issue rate of v_mad_u64_u32, so what counts is multiply-adds
// per pairing, not the length of the chain.  The one-coefficient-per-lane engine (h2v_pairing_coop.hpp, 16 lanes per proof)
// leaves 4 of 16 lanes idle outside the line steps, and a lane that owns ONE Fp coefficient of an Fp2 product has to
// take the schoolbook four products.  Here lane k < 6 of a group owns Fp2 coefficient k WHOLE, so
//   * 60 of 64 lanes work in every engine call (ten groups per wave), and
//   * an Fp2 x Fp2 term is a Karatsuba term - three Fp products into three sets of 64-bit column accumulators
//         U += x0 y0      V += x1 y1      W += (x0 + x1)(y0 + y1)        re = U - V      im = W - U - V
//     with ONE Montgomery reduction per part at the end of the call: (3 NT + 2) x 196 multiply-adds per Fp2 coefficient
//     where two lanes of the other engine spend 2 (2 NT + 1) x 196 (MUL, NT = 6: 3920 against 5096).
// The sums x0 + x1, y0 + y1 are formed LIMB-WISE in registers (uncarried), so W - U - V equals sum(x0 y1 + x1 y0) column by column: the
// imaginary part's columns are non-negative (unsigned reduction; W itself may wrap modulo 2^64 on the way - harmless);
// U - V is taken in two's complement columns (|column| < 2^63) and reduced with arithmetic carries, p added at the end.
// Wrapped terms (x xi) take the xi on the A side: XA = xi a = (a0 - a1, a0 + a1).  The cyclotomic squaring is five
// products into U, U, V, W, W (re = U - V, im = W + V: S_b M_b and 2 b0 b1 serve both parts); the lane forms 3 r -/+ 2 g itself
// and folds it below 2p with a quotient estimate from the top limb (f28_fold).  Per line the products
// b = (-lambda) xP of the two loops are taken by lanes 0..3 of the group just before the two line steps (one product each).
// Tables and the headroom argument: tools/gen_six_tables.py (value-level check against big-integer Fp12 arithmetic and a
// limb-level model of this code with the column bounds asserted).  Program, line tables, semantics: h2v_pairing_coop.hpp.
//
// LDS: 38 operand slots of 56 bytes per proof (14 limbs, no padding: 8-byte reads), 23 KB per wave - six or seven waves per
// CU, and at most 256 registers per lane, so that a SIMD holds a second wave beside this one: a lone wave issues a
// v_mad_u64_u32 only every ~11 cycles, two waves one every ~5.5 (profiles/r03_imad_ubench.txt).  (A first version with staged
// sums and 80-byte slots - 47 KB per wave, 356 registers - was correct and 20 % SLOWER than the narrow engine in the full
// pipeline: 4.91 against 4.06 ms per step, the kernel alone 4.80 ms: its waves held their SIMDs alone.)
// Ten pairings share one instruction stream: for one batch at a time the other engines are faster; see launch_pairing.
#pragma once
#include "h2v_pairing_coop.hpp"
#include "six_tables.h"

#define H2V_NO_TAIL_MARK __attribute__((disable_tail_calls))
#define SIX_GROUPS 10
#define SIX_SLOT_DW 14
#define SIX_GROUP_DW (SIX_N_GROUP_SLOTS * SIX_SLOT_DW)
#define SIX_TAB_DW ((6 * 4 * SIX_N_MUL + 6 * 4 * SIX_N_SQR + 2 * 6 * 4 * SIX_N_LINE + 6 * 16) / 4)
#define SIX_TAB_OFF (SIX_N_SHARED_SLOTS * SIX_SLOT_DW)
#define SIX_GRP_OFF ((SIX_TAB_OFF + SIX_TAB_DW + 1) & ~1)
#define SIX_LDS_BYTES ((size_t)(SIX_GRP_OFF + SIX_GROUPS * SIX_GROUP_DW) * 4)
#define SIX_TAB_MUL_B 0
#define SIX_TAB_SQR_B (6 * 4 * SIX_N_MUL)
#define SIX_TAB_LINE1_B (SIX_TAB_SQR_B + 6 * 4 * SIX_N_SQR)
#define SIX_TAB_LINE2_B (SIX_TAB_LINE1_B + 6 * 4 * SIX_N_LINE)
#define SIX_TAB_CSQR_B (SIX_TAB_LINE2_B + 6 * 4 * SIX_N_LINE)
#define SIX_C_M(k) ((k) < 3 ? 19 + (k) : 31 + (k))      // gen_six_tables.py: C_M

struct Six {
    int grp_off;  // dword offset of the group's slots in coop_lds
    int k;        // Fp2 coefficient 0..5
    bool act;     // lanes 60..63 shadow group 9 and never store
};
struct SixF2 { F28 re, im; };
struct SixRegs { F28Regs re, im; };

H2V_DI uint32_t *six_slot(const Six &c, int s) {
    return coop_lds + (s < SIX_SHARED_BASE ? c.grp_off + s * SIX_SLOT_DW : (s - SIX_SHARED_BASE) * SIX_SLOT_DW);
}
H2V_DI void six_store(uint32_t *p, const F28 &a) {   // limbs as staged (carried, or the doubled operands' 2^29)
    uint2 *q = reinterpret_cast<uint2 *>(p);
#pragma unroll
    for (int i = 0; i < 7; i++) q[i] = make_uint2(a.l[2 * i], a.l[2 * i + 1]);
}
H2V_DI void six_store(uint32_t *p, const Fp &a) {
    F28 t;
    f28_from_fp(t, a);
    six_store(p, t);
}
// Two operands with 64-bit LDS reads (slots are 8-byte aligned 56-byte records); one asm block for the reason given at
// coop_load28_pair.
#define SIX_RD7(o0, o1, o2, o3, o4, o5, o6, a)                                                                              \
    "ds_read_b64 %" #o0 ", %" #a "\n\tds_read_b64 %" #o1 ", %" #a " offset:8\n\tds_read_b64 %" #o2 ", %" #a " offset:16\n\t"     \
    "ds_read_b64 %" #o3 ", %" #a " offset:24\n\tds_read_b64 %" #o4 ", %" #a " offset:32\n\tds_read_b64 %" #o5 ", %" #a " offset:40\n\t" \
    "ds_read_b64 %" #o6 ", %" #a " offset:48\n\t"
H2V_DI void six_load_pair(uint32_t (&x)[14], uint32_t (&y)[14], const uint32_t *px, const uint32_t *py) {
    const uint32_t ax = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t *)px;
    const uint32_t ay = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t *)py;
    u32x2_t a0, a1, a2, a3, a4, a5, a6, b0, b1, b2, b3, b4, b5, b6;
    asm volatile(SIX_RD7(0, 1, 2, 3, 4, 5, 6, 14) SIX_RD7(7, 8, 9, 10, 11, 12, 13, 15) "s_waitcnt lgkmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3),
                   "=&v"(b4), "=&v"(b5), "=&v"(b6)
                 : "v"(ax), "v"(ay)
                 : "memory");
    x[0] = a0.x; x[1] = a0.y; x[2] = a1.x; x[3] = a1.y; x[4] = a2.x; x[5] = a2.y; x[6] = a3.x; x[7] = a3.y;
    x[8] = a4.x; x[9] = a4.y; x[10] = a5.x; x[11] = a5.y; x[12] = a6.x; x[13] = a6.y;
    y[0] = b0.x; y[1] = b0.y; y[2] = b1.x; y[3] = b1.y; y[4] = b2.x; y[5] = b2.y; y[6] = b3.x; y[7] = b3.y;
    y[8] = b4.x; y[9] = b4.y; y[10] = b5.x; y[11] = b5.y; y[12] = b6.x; y[13] = b6.y;
}
H2V_DI void six_load(uint32_t (&x)[14], const uint32_t *px) {
    const uint32_t ax = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t *)px;
    u32x2_t a0, a1, a2, a3, a4, a5, a6;
    asm volatile(SIX_RD7(0, 1, 2, 3, 4, 5, 6, 7) "s_waitcnt lgkmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6)
                 : "v"(ax)
                 : "memory");
    x[0] = a0.x; x[1] = a0.y; x[2] = a1.x; x[3] = a1.y; x[4] = a2.x; x[5] = a2.y; x[6] = a3.x; x[7] = a3.y;
    x[8] = a4.x; x[9] = a4.y; x[10] = a5.x; x[11] = a5.y; x[12] = a6.x; x[13] = a6.y;
}
H2V_DI void six_mac(uint64_t (&acc)[28], const uint32_t (&x)[14], const uint32_t (&y)[14]) {
#pragma unroll
    for (int i = 0; i < 14; i++)
#pragma unroll
        for (int j = 0; j < 14; j++) acc[i + j] += (uint64_t)x[i] * y[j];
}
// Montgomery reduction of 28 columns, R = 2^392.  SIGNED: two's complement columns, arithmetic carries, + p at the end
// (the value is above -p: gen_six_tables.py).  Result limbs carried.
template <bool SIGNED>
H2V_DI void six_reduce(F28 &r, uint64_t (&acc)[28]) {
#pragma unroll
    for (int k = 0; k < 14; k++) {
        const uint32_t m = ((uint32_t)acc[k] * FP_N0_28) & FP28_MASK;
#pragma unroll
        for (int j = 0; j < 14; j++) acc[k + j] += (uint64_t)m * FP_MOD28[j];
        acc[k + 1] += SIGNED ? (uint64_t)((int64_t)acc[k] >> 28) : acc[k] >> 28;
    }
    if (SIGNED) {
        int64_t carry = 0;
#pragma unroll
        for (int k = 0; k < 13; k++) {
            carry += (int64_t)acc[14 + k] + (int64_t)FP_MOD28[k];
            r.l[k] = (uint32_t)carry & FP28_MASK;
            carry >>= 28;
        }
        r.l[13] = (uint32_t)(carry + (int64_t)acc[27] + (int64_t)FP_MOD28[13]);
    } else {
        uint64_t carry = 0;
#pragma unroll
        for (int k = 0; k < 13; k++) {
            carry += acc[14 + k];
            r.l[k] = (uint32_t)carry & FP28_MASK;
            carry >>= 28;
        }
        r.l[13] = (uint32_t)(carry + acc[27]);
    }
}
// NT Karatsuba terms from the lane's table row (4 slot bytes per term: x0 y0 x1 y1) -> (re, im), both below 3p.
// The result does not come back through the call: it is written, carried, into the lane's own A slots (2k, 2k + 1) - every
// operand slot is dead once the last term has been read (one wave per block: the lanes have all read before any writes) -
// and the caller reads it from there (six_result).  A 28-dword struct is returned through private memory by the calling
// convention: a store, a wait for it, and a load per engine call, ~550 calls per pairing (round 3: the kernel's scratch
// traffic).  keep (per lane): leave the A slots as they are - the staged input survives (line steps of a skipped loop).
H2V_DN void six_kara(const Six c, const int tab_row_byte, const int nt, const bool keep) {
    const uint8_t *tab = reinterpret_cast<const uint8_t *>(coop_lds + SIX_TAB_OFF) + tab_row_byte;
    uint64_t U[28], V[28], W[28];
#pragma unroll
    for (int i = 0; i < 28; i++) { U[i] = 0; V[i] = 0; W[i] = 0; }
#pragma unroll 1
    for (int t = 0; t < nt; t++) {
        uint32_t x0[14], y0[14], x1[14], y1[14];
        six_load_pair(x0, y0, six_slot(c, tab[4 * t]), six_slot(c, tab[4 * t + 1]));
        six_mac(U, x0, y0);
        six_load_pair(x1, y1, six_slot(c, tab[4 * t + 2]), six_slot(c, tab[4 * t + 3]));
        six_mac(V, x1, y1);
#pragma unroll
        for (int i = 0; i < 14; i++) { x1[i] += x0[i]; y1[i] += y0[i]; }
        six_mac(W, x1, y1);
    }
#pragma unroll
    for (int i = 0; i < 27; i++) {   // (column 27 holds no product)
        W[i] -= U[i] + V[i];
        U[i] -= V[i];
    }
    SixF2 r;
    six_reduce<false>(r.im, W);
    six_reduce<true>(r.re, U);
    if (c.act && !keep) {
        six_store(six_slot(c, SIX_SLOT_A + 2 * c.k), r.re);
        six_store(six_slot(c, SIX_SLOT_A + 2 * c.k + 1), r.im);
    }
}
// cyclotomic squaring: five products (x + x2) y into U, U, V, W, W (3 slot bytes each; gen_six_tables.py: csqr_table) ->
// re = U - V (signed columns), im = W + V, reduced: re below 2.2p, im below 1.2p.  The lane finishes with 3 r -/+ 2 g (six_csqr).
// Result into the lane's B slots (the doubled operand D, dead by then); the A slots keep g for the caller.
H2V_DN void six_csqr_engine(const Six c, const int tab_row_byte) {
    const uint8_t *tab = reinterpret_cast<const uint8_t *>(coop_lds + SIX_TAB_OFF) + tab_row_byte;
    uint64_t U[28], V[28], W[28];
#pragma unroll
    for (int i = 0; i < 28; i++) { U[i] = 0; V[i] = 0; W[i] = 0; }
    uint32_t x[14], x2[14], y[14];
#define SIX_CSQR_SET(ACC, T)                                                                             \
    do {                                                                                                 \
        six_load_pair(x, x2, six_slot(c, tab[3 * (T)]), six_slot(c, tab[3 * (T) + 1]));                  \
        six_load(y, six_slot(c, tab[3 * (T) + 2]));                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < 14; i_++) x[i_] += x2[i_];                               \
        six_mac(ACC, x, y);                                                                              \
    } while (0)
    SIX_CSQR_SET(U, 0);
    SIX_CSQR_SET(U, 1);
    SIX_CSQR_SET(V, 2);
    SIX_CSQR_SET(W, 3);
    SIX_CSQR_SET(W, 4);
#undef SIX_CSQR_SET
#pragma unroll
    for (int i = 0; i < 27; i++) {   // (column 27 holds no product)
        U[i] -= V[i];
        W[i] += V[i];
    }
    SixF2 r;
    six_reduce<true>(r.re, U);
    six_reduce<false>(r.im, W);
    if (c.act) {
        six_store(six_slot(c, SIX_SLOT_B + 2 * c.k), r.re);
        six_store(six_slot(c, SIX_SLOT_B + 2 * c.k + 1), r.im);
    }
}
// one product of two staged slots, reduced (< 2p)
H2V_DN F28Regs six_prod(const Six c, const int xs, const int ys) {
    uint64_t acc[28];
#pragma unroll
    for (int i = 0; i < 28; i++) acc[i] = 0;
    uint32_t x[14], y[14];
    six_load_pair(x, y, six_slot(c, xs), six_slot(c, ys));
    six_mac(acc, x, y);
    F28 r;
    six_reduce<false>(r, acc);
    return f28_pack(r);
}
// an engine's result (or a staged value) back from the lane's slots `base + 2k`, `base + 2k + 1`; shadow lanes read group 9's
H2V_DI SixF2 six_result(const Six &c, const int base) {
    SixF2 r;
    six_load_pair(r.re.l, r.im.l, six_slot(c, base + 2 * c.k), six_slot(c, base + 2 * c.k + 1));
    return r;
}
H2V_DI SixF2 six_unpack(const SixRegs &z) {
    SixF2 r;
    r.re = f28_unpack(z.re.a, z.re.b, z.re.c, z.re.d);
    r.im = f28_unpack(z.im.a, z.im.b, z.im.c, z.im.d);
    return r;
}

// ---- staging (values: v <= 6, carried).  Slots: six_tables.h / gen_six_tables.py
H2V_DI void six_stage_a(const Six &c, const SixF2 &a, const int xa_from) {
    if (!c.act) return;
    six_store(six_slot(c, SIX_SLOT_A + 2 * c.k), a.re);
    six_store(six_slot(c, SIX_SLOT_A + 2 * c.k + 1), a.im);
    if (c.k >= xa_from) {   // xi a = (re - im + 7p, re + im)
        F28 t, x0, x1;
        F28_NEG(t, a.im, 7, 1);
        f28_add(x0, a.re, t);
        f28_carry(x0);
        f28_add(x1, a.re, a.im);
        f28_carry(x1);
        six_store(six_slot(c, SIX_SLOT_XA + 2 * (c.k - 1)), x0);
        six_store(six_slot(c, SIX_SLOT_XA + 2 * (c.k - 1) + 1), x1);
    }
}
H2V_DI void six_stage_b(const Six &c, const SixF2 &b) {
    if (!c.act) return;
    six_store(six_slot(c, SIX_SLOT_B + 2 * c.k), b.re);
    six_store(six_slot(c, SIX_SLOT_B + 2 * c.k + 1), b.im);
}
H2V_DI void six_stage_d(const Six &c, const SixF2 &a) {   // D = 2a, uncarried
    if (!c.act) return;
    F28 d0, d1;
    f28_mul_small<2>(d0, a.re);
    f28_mul_small<2>(d1, a.im);
    six_store(six_slot(c, SIX_SLOT_B + 2 * c.k), d0);
    six_store(six_slot(c, SIX_SLOT_B + 2 * c.k + 1), d1);
}
H2V_DI SixF2 six_mul(const Six &c, const SixF2 &a, const SixF2 &b) {
    six_stage_a(c, a, 1);
    six_stage_b(c, b);
    __syncthreads();
    six_kara(c, SIX_TAB_MUL_B + c.k * 4 * SIX_N_MUL, SIX_N_MUL, false);
    __syncthreads();
    return six_result(c, SIX_SLOT_A);
}
H2V_DI SixF2 six_sqr(const Six &c, const SixF2 &a) {
    six_stage_a(c, a, 3);
    six_stage_d(c, a);
    __syncthreads();
    six_kara(c, SIX_TAB_SQR_B + c.k * 4 * SIX_N_SQR, SIX_N_SQR, false);
    __syncthreads();
    return six_result(c, SIX_SLOT_A);
}
// f times the line of loop LOOP; skip (per proof: that loop's G1 argument is infinity): f comes back unchanged
template <int LOOP>
H2V_DI SixF2 six_line(const Six &c, const SixF2 &f, const bool skip) {
    six_stage_a(c, f, 3);
    __syncthreads();
    six_kara(c, (LOOP == 1 ? SIX_TAB_LINE1_B : SIX_TAB_LINE2_B) + c.k * 4 * SIX_N_LINE, SIX_N_LINE, skip);
    __syncthreads();
    return six_result(c, SIX_SLOT_A);
}
H2V_DI SixF2 six_csqr(const Six &c, const SixF2 &a) {
    if (c.act) {
        F28 na, d0, d1, m;
        F28_NEG(na, a.im, 7, 1);               // 7p - im                                  (7, 3)
        f28_add(m, a.re, na);                  // re - im + 7p                             (13, 4)
        f28_carry(na);
        f28_carry(m);
        f28_mul_small<2>(d0, a.re);            // D = 2a, uncarried                        (12, 2)
        f28_mul_small<2>(d1, a.im);
        six_store(six_slot(c, SIX_SLOT_A + 2 * c.k), a.re);       // (S = re + im is formed by the engine)
        six_store(six_slot(c, SIX_SLOT_A + 2 * c.k + 1), a.im);
        six_store(six_slot(c, SIX_SLOT_C_NA + c.k), na);
        six_store(six_slot(c, SIX_SLOT_B + 2 * c.k), d0);
        six_store(six_slot(c, SIX_SLOT_B + 2 * c.k + 1), d1);
        six_store(six_slot(c, SIX_C_M(c.k)), m);
        if (c.k == 2) {                        // ND2 = 2 NA_2, uncarried                  (14, 2)
            F28 nd;
            f28_mul_small<2>(nd, na);
            six_store(six_slot(c, SIX_SLOT_C_ND2), nd);
        }
    }
    __syncthreads();
    six_csqr_engine(c, SIX_TAB_CSQR_B + c.k * 16);
    __syncthreads();
    SixF2 r = six_result(c, SIX_SLOT_B);
    const SixF2 g = six_result(c, SIX_SLOT_A);      // (a itself, read back: nothing of it stays live across the engine call)
    // h_k = 3 Q_k - 2 g_k (k even) / + 2 g_k (k odd), folded: 3 r + (13p - 2g | 2g) is below 20p (r < 2.2p, g < 6p), the fold
    // brings it below 2p.  (The other engines multiply g by the constants -/+ 2/3 inside the sum: two products more per lane.)
    const bool minus = (c.k & 1) == 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const uint32_t g0 = g.re.l[i] << 1, g1 = g.im.l[i] << 1;
        r.re.l[i] = 3u * r.re.l[i] + (minus ? F28_BIAS_13_2[i] - g0 : g0);
        r.im.l[i] = 3u * r.im.l[i] + (minus ? F28_BIAS_13_2[i] - g1 : g1);
    }
    f28_carry(r.re);
    f28_carry(r.im);
    f28_fold(r.re);
    f28_fold(r.im);
    return r;
}
H2V_DI SixF2 six_conj(const Six &c, const SixF2 &a) {   // w -> -w: odd coefficients change sign.  a: v <= 5
    SixF2 r = a;
    if (c.k & 1) {
        F28_NEG(r.re, a.re, 6, 1);
        F28_NEG(r.im, a.im, 6, 1);
        f28_carry(r.re);
        f28_carry(r.im);
    }
    return r;
}
H2V_DI SixF2 six_frob(const Six &c, const SixF2 &a) {   // a -> a^p: coefficient k becomes conj(a_k) * gamma^k.  Result v <= 5
    Fp g0, g1;
#pragma unroll
    for (int i = 0; i < 12; i++) { g0.v[i] = FROB_GAMMA[c.k][0][i]; g1.v[i] = FROB_GAMMA[c.k][1][i]; }
    F28 h0, h1, x, y;
    f28_from_fp(h0, g0);
    f28_from_fp(h1, g1);
    SixF2 r;
    f28_mul(x, a.re, h0); f28_mul(y, a.im, h1); f28_add(r.re, x, y);          // re g0 + im g1
    f28_mul(x, a.re, h1); f28_mul(y, a.im, h0); F28_SUB(r.im, x, y, 3, 1);    // re g1 - im g0
    f28_carry(r.re);
    f28_carry(r.im);
    return r;
}
// 1/f = conj(f) / N with N = f conj(f) in Fp6 (the even coefficients).  1/N without tower code on one lane: the conjugates of
// N over Fp2 are its images under Frobenius p^2 and p^4, so adj = N^(p^2) N^(p^4) and N adj = Norm(N) lies in Fp2 (coefficient 0);
// lane 0 of the group inverts that one Fp2 element (one Fp inversion), and 1/N = adj / Norm(N).  Four Frobenius maps and four
// engine products more than the tower inversion the other engines run on their first lane - and 1.2 KB less scratch per lane
// (Fp6 temporaries and the frames of fp6_inv), which is what a queue's scratch arena is sized by.
// (H2V_NO_TAIL_MARK: see k_pairing_six)
H2V_DN H2V_NO_TAIL_MARK SixRegs six_inv_raw(const Six c, const SixRegs fr, bool &ok) {
    const SixF2 f = six_unpack(fr);
    const SixF2 fc = six_conj(c, f);
    const SixF2 nrm = six_mul(c, f, fc);                       // N
    const SixF2 n2 = six_frob(c, six_frob(c, nrm));            // N^(p^2)        (v <= 5)
    const SixF2 n4 = six_frob(c, six_frob(c, n2));             // N^(p^4)
    const SixF2 adj = six_mul(c, n2, n4);
    const SixF2 nn = six_mul(c, nrm, adj);                     // Norm(N): coefficient 0 only
    Fp2 z, zi;
    f28_to_fp(z.c0, nn.re);
    f28_to_fp(z.c1, nn.im);
    fp_set_zero(zi.c0);
    fp_set_zero(zi.c1);
    bool good = true;
    if (c.k == 0) good = fp2_inv(zi, z);
    ok = good;
    SixF2 ninv;                                                // 1 / Norm(N) as an Fp12 element: coefficient 0, zero elsewhere
    f28_from_fp(ninv.re, zi.c0);
    f28_from_fp(ninv.im, zi.c1);
    const SixF2 r = six_mul(c, fc, six_mul(c, adj, ninv));
    SixRegs o;
    o.re = f28_pack(r.re);
    o.im = f28_pack(r.im);
    return o;
}
// The constants of line `idx` as the lane's share of the copy into the wave-shared slots: 8 slots x 16 dwords in the plan
// (14 limbs + 2 of padding), lane -> slot lane / 8, dwords 2 (lane % 8)..
H2V_DI uint2 six_line_share(const uint32_t *lines28, int idx, int lane) {
    return reinterpret_cast<const uint2 *>(lines28 + (size_t)idx * 8 * 16)[lane];
}
H2V_DI void six_line_store(int shared_slot, const uint2 v, int lane) {
    if ((lane & 7) == 7) return;    // the padding
    uint32_t *dst = coop_lds + (shared_slot - SIX_SHARED_BASE + (lane >> 3)) * SIX_SLOT_DW + 2 * (lane & 7);
    dst[0] = v.x;
    dst[1] = v.y;
}

// The engines return nothing and take no pointer into their caller's frame, so LLVM would mark their call sites `tail`; a
// callee with a `tail`-marked call site is excluded from the interprocedural register allocation's no-callee-saved-registers
// form (TargetFrameLowering::isSafeForNoCSROpt), and six_kara - which uses every register - would save and restore all 112
// callee-saved VGPRs through private memory on every call (seen: 106 stores + 106 loads per call).  The callers therefore
// opt out of tail-call marking.
extern "C" __global__ void __launch_bounds__(64, 2) H2V_NO_TAIL_MARK
k_pairing_six(H2vDevPlan plan, uint32_t n, const uint32_t *__restrict__ pts, const uint8_t *__restrict__ valid, const uint8_t *__restrict__ valid_sub,
              const uint32_t *__restrict__ er_jac, const uint32_t *__restrict__ el_jac /* folded el (recursion) or NULL */,
              uint32_t *__restrict__ status, uint8_t *__restrict__ accept, uint32_t *__restrict__ dbg) {
    const int lane = threadIdx.x;
    const int grp6 = (lane * 43) >> 8;          // lane / 6 for lane < 64
    Six c;
    c.act = lane < 6 * SIX_GROUPS;
    const int grp = c.act ? grp6 : SIX_GROUPS - 1;
    c.k = c.act ? lane - 6 * grp6 : lane - 6 * SIX_GROUPS;
    c.grp_off = SIX_GRP_OFF + grp * SIX_GROUP_DW;
    const int leader = grp * 6;
    const bool is_leader = c.act && c.k == 0;
    const uint32_t i = blockIdx.x * SIX_GROUPS + grp;
    const bool live = i < n;
    const uint32_t ii = live ? i : n - 1;       // dead groups shadow the last proof, never write
    const uint32_t slots = H2V_SLOTS(plan);

    {   // operand tables -> LDS
        const uint32_t *src[5] = {reinterpret_cast<const uint32_t *>(&SIX_TAB_MUL[0][0]), reinterpret_cast<const uint32_t *>(&SIX_TAB_SQR[0][0]),
                                  reinterpret_cast<const uint32_t *>(&SIX_TAB_LINE1[0][0]), reinterpret_cast<const uint32_t *>(&SIX_TAB_LINE2[0][0]),
                                  reinterpret_cast<const uint32_t *>(&SIX_TAB_CSQR[0][0])};
        const int off[6] = {SIX_TAB_MUL_B / 4, SIX_TAB_SQR_B / 4, SIX_TAB_LINE1_B / 4, SIX_TAB_LINE2_B / 4, SIX_TAB_CSQR_B / 4, SIX_TAB_DW};
#pragma unroll
        for (int t = 0; t < 5; t++)
            for (int q = lane; q < off[t + 1] - off[t]; q += 64) coop_lds[SIX_TAB_OFF + off[t] + q] = src[t][q];
    }
    if (lane < 14) coop_lds[(SIX_SLOT_ZERO - SIX_SHARED_BASE) * SIX_SLOT_DW + lane] = 0u;   // the shared zero operand
    // ---- leader: status, the two G1 arguments (el ; -er normalised to affine)
    uint32_t st = 0;
    uint32_t flags = 0;  // bit0: el is infinity, bit1: er is infinity
    if (is_leader) {
        st = status[ii];
        for (uint32_t j = 0; j < slots; j++)
            if (!valid[(size_t)ii * slots + j] || (valid_sub && !valid_sub[(size_t)ii * slots + j])) st |= H2V_ST_BAD_POINT;
        G1A el, er;
        G1J ej;
        const uint32_t *pp = pts + ((size_t)ii * slots + plan.pi_point) * 24;
#pragma unroll
        for (int q = 0; q < 12; q++) {
            el.x.v[q] = pp[q]; el.y.v[q] = pp[12 + q];
            ej.x.v[q] = er_jac[(size_t)ii * 36 + q]; ej.y.v[q] = er_jac[(size_t)ii * 36 + 12 + q]; ej.z.v[q] = er_jac[(size_t)ii * 36 + 24 + q];
        }
        if (st != 0) { g1a_set_inf(el); g1j_set_inf(ej); }  // rejected already: keep the arithmetic well-defined
        g1j_to_affine(er, ej);
        if (el_jac && st == 0) {
            G1J lj;
#pragma unroll
            for (int q = 0; q < 12; q++) { lj.x.v[q] = el_jac[(size_t)ii * 36 + q]; lj.y.v[q] = el_jac[(size_t)ii * 36 + 12 + q]; lj.z.v[q] = el_jac[(size_t)ii * 36 + 24 + q]; }
            g1j_to_affine(el, lj);
        }
        if (g1a_is_inf(el)) flags |= 1;
        if (g1a_is_inf(er)) flags |= 2;
        fp_neg(er.y, er.y);
        six_store(six_slot(c, SIX_SLOT_PX1), el.x);
        six_store(six_slot(c, SIX_SLOT_PY1), el.y);
        six_store(six_slot(c, SIX_SLOT_PX2), er.x);
        six_store(six_slot(c, SIX_SLOT_PY2), er.y);
    }
    flags = __shfl(flags, leader);
    st = __shfl(st, leader);
    const bool skip1 = (flags & 1) != 0, skip2 = (flags & 2) != 0;
    bool inv_ok = true;
    __syncthreads();

    SixF2 vars[COOP_N_VARS];
    for (int pc = 0; pc < COOP_PROGRAM_LEN; pc++) {
        const uint32_t ins = COOP_PROGRAM[pc];
        const int op = ins & 0xff, d = (ins >> 8) & 0xff, a = (ins >> 16) & 0xff, b = ins >> 24;
        if (op == COOP_OP_END) break;
        switch (op) {
        case COOP_OP_MUL: {
            const SixF2 x = vars[a], y = vars[b];
            vars[d] = six_mul(c, x, y);
        } break;
        case COOP_OP_CSQR: {
            SixF2 x = vars[a];
#pragma unroll 1
            for (int rep = 0; rep < b; rep++) x = six_csqr(c, x);
            vars[d] = x;
        } break;
        case COOP_OP_MILLER: {
            // Per bit of |x| below the leading one: F = F^2, then one or two rounds of { the products b = (-lambda) xP of both
            // loops' lines (lanes 0..3: loop = k >> 1, part = k & 1), line of loop 1, line of loop 2 }.  The constants of the
            // next round's two lines are fetched a round ahead into two registers per lane.
            SixF2 f = vars[COOP_VAR_F];
            int ln = 0;
            uint2 c1 = six_line_share(plan.lines28_sg2, 0, lane), c2 = six_line_share(plan.lines28_g2, 0, lane);
#pragma unroll 1
            for (int bit = 62; bit >= 0; bit--) {
                f = six_sqr(c, f);
                const int steps = ((BLS_X_ABS >> bit) & 1) ? 2 : 1;
#pragma unroll 1
                for (int s2 = 0; s2 < steps; s2++, ln++) {
                    six_line_store(SIX_SLOT_LN1, c1, lane);
                    six_line_store(SIX_SLOT_LN2, c2, lane);
                    if (ln + 1 < H2V_MILLER_LINES) { c1 = six_line_share(plan.lines28_sg2, ln + 1, lane); c2 = six_line_share(plan.lines28_g2, ln + 1, lane); }
                    __syncthreads();
                    {
                        const int u = (c.k >> 1) & 1, part = c.k & 1;
                        const F28Regs z = six_prod(c, (u ? SIX_SLOT_LN2 : SIX_SLOT_LN1) + part, u ? SIX_SLOT_PX2 : SIX_SLOT_PX1);
                        if (c.act && c.k < 4) six_store(six_slot(c, (u ? SIX_SLOT_T2 : SIX_SLOT_T1) + part), f28_unpack(z.a, z.b, z.c, z.d));
                    }
                    f = six_line<1>(c, f, skip1);                   // (its staging barrier also covers the T slots)
                    f = six_line<2>(c, f, skip2);
                }
            }
            vars[COOP_VAR_F] = f;
        } break;
        case COOP_OP_EXPX: {   // d = a^x (x < 0: conjugate of a^|x|)
            SixF2 x = vars[a];
#pragma unroll 1
            for (int bit = 62; bit >= 0; bit--) {
                x = six_csqr(c, x);
                if ((BLS_X_ABS >> bit) & 1) {
                    const SixF2 y = vars[a];
                    x = six_mul(c, x, y);
                }
            }
            vars[d] = six_conj(c, x);
        } break;
        case COOP_OP_WARMUP: break;   // (the other engines' first-line products: here every round takes its own)
        case COOP_OP_CONJ: vars[d] = six_conj(c, vars[a]); break;
        case COOP_OP_FROB: vars[d] = six_frob(c, vars[a]); break;
        case COOP_OP_INV: {
            bool ok = true;
            SixRegs fr;
            fr.re = f28_pack(vars[a].re);
            fr.im = f28_pack(vars[a].im);
            vars[d] = six_unpack(six_inv_raw(c, fr, ok));
            inv_ok = ok;
        } break;
        case COOP_OP_MOV: vars[d] = vars[a]; break;
        case COOP_OP_SETONE: {
            SixF2 o;
            f28_set_zero(o.re);
            f28_set_zero(o.im);
            if (c.k == 0) f28_set_one(o.re);
            vars[d] = o;
        } break;
        case COOP_OP_DUMP: {
            if (dbg && live && c.act) {
                Fp o, oc;
                f28_to_fp(oc, vars[a].re);
                fp_from_mont(o, oc);
#pragma unroll
                for (int q = 0; q < 12; q++) dbg[((size_t)i * 24 + 12 * d + 2 * c.k) * 12 + q] = o.v[q];
                f28_to_fp(oc, vars[a].im);
                fp_from_mont(o, oc);
#pragma unroll
                for (int q = 0; q < 12; q++) dbg[((size_t)i * 24 + 12 * d + 2 * c.k + 1) * 12 + q] = o.v[q];
            }
        } break;
        default: break;
        }
    }
    // == 1 ?
    Fp res0, res1;
    f28_to_fp(res0, vars[COOP_VAR_F].re);
    f28_to_fp(res1, vars[COOP_VAR_F].im);
    bool mine = fp_is_zero(res1);
    if (c.k == 0) { Fp one; fp_set_one(one); mine = mine && fp_eq(res0, one); }
    else mine = mine && fp_is_zero(res0);
    const unsigned long long bal = __ballot(mine);
    const bool is_one = ((bal >> leader) & 0x3full) == 0x3full;
    inv_ok = __shfl((int)inv_ok, leader) != 0;
    if (is_leader && live) {
        if (st == 0 && !(is_one && inv_ok)) st |= H2V_ST_PAIRING;
        status[i] = st;
        accept[i] = st == 0 ? 1 : 0;
    }
}
