// Cooperative pairing check: ONE PROOF PER 32-LANE GROUP (2 proofs per wave).
//
// Why: a batch of 4096 proofs gives a one-lane-per-proof pairing kernel only 64 waves on a chip with 1024 SIMDs,
// and every Fp12 operation there is ~54 dependent Fp multiplications.  Here lane g < 12 of a group owns ONE Fp
// coefficient c_{k,part} (k = g >> 1, part = g & 1) of each Fp12 value (flat basis Fp12 = Fp2[w]/(w^6 - xi)), so an
// Fp12 value costs 12 VGPRs per lane, and a product is computed by all 12 lanes at once:
//     c_{k,part} = sum over (i, j = k-i mod 6) of  a_i x b_j  (x xi when wrapped)      -> 12 Fp x Fp terms per lane,
// accumulated UNREDUCED in 28 64-bit column accumulators (12 terms x 14 products x 2^56 < 2^64) with a single
// Montgomery reduction at the end: 12*196 + 196 v_mad_u64_u32 per lane instead of 54*392 on one lane.
// Operands are staged in LDS as 28-bit limbs (64-byte slots); which slots a lane multiplies comes from the
// generated, big-integer-verified tables of coop_tables.h (tools/gen_coop_tables.py).  Lanes 12..15 of a group
// compute the NEXT Miller-loop line's per-proof products (-lambda)*xP in the same engine call.
// Each coefficient is shared by TWO lanes (g, g+16): they take alternate terms of the sum and exchange their
// unreduced column accumulators with one cross-lane shuffle before the reduction, which halves the dependent chain
// and doubles the number of waves (4096 proofs -> 2048 waves = 2 per SIMD, where v_mad_u64_u32 issues at full rate).
//
// Semantics are those of k_pairing_check (h2v_kernels.hip): accept <=> e(el, s_g2) == e(er, G2)
// (/root/reference/aiken-verifier/templates/verification_h2.hbs:125-128).
#pragma once
#include "coop_program.h"
#include "coop_tables.h"
#include "h2v_curve.hpp"
#include "h2v_fp28.hpp"
#include "h2v_plan.h"
#include "h2v_tower.hpp"

#define COOP_SLOT_DW 20                      // dwords per operand slot: 14 limbs + pad; 20 keeps 16-byte alignment and
                                             // spreads 16 consecutive slots over 16 distinct bank groups (stride 16
                                             // measured 41 % bank-conflict cycles)
#define COOP_GROUP_SLOTS 64                  // slots reserved per group (COOP_N_GROUP_SLOTS used)
#define COOP_GROUP_DW (COOP_GROUP_SLOTS * COOP_SLOT_DW)   // (padding the group stride by 4 / 8 / 16 dwords changed nothing: 2.653-2.662 ms)
#define COOP_GROUPS_PER_WAVE 2               // the normal kernel; the narrow one has 4 (COOP_GROUPS_NARROW), the wide one 1
#define COOP_GROUPS_NARROW 4
#define COOP_GROUPS_TWELVE 5
#define COOP_TAB_DW ((16 * 2 * COOP_N_MUL_TERMS + 2 * 16 * 2 * COOP_N_LINE_TERMS + 16 * 2 * COOP_N_CSQR_TERMS + 16 * 2 * COOP_N_SQR_TERMS) / 4)  // operand tables, copied at start

// File-scope LDS so that every device function addresses it as LDS (ds_read/ds_write), not through flat pointers; sized
// at launch (COOP_LDS_BYTES(groups)): wave-shared slots, operand tables, then one slot area per group of the wave.
extern __shared__ __attribute__((aligned(16))) uint32_t coop_lds[];
#define COOP_SHR_OFF 0
#define COOP_TAB_OFF (COOP_SHR_OFF + COOP_N_SHARED_SLOTS * COOP_SLOT_DW)
#define COOP_GRP_OFF ((COOP_TAB_OFF + COOP_TAB_DW + 3) & ~3)
#define COOP_LDS_BYTES(groups) ((size_t)(COOP_GRP_OFF + (groups) * COOP_GROUP_DW) * 4)
#define COOP_TAB_MUL_B 0                                   // byte offsets of the three tables inside the LDS copy
#define COOP_TAB_LINE1_B (16 * 2 * COOP_N_MUL_TERMS)
#define COOP_TAB_LINE2_B (COOP_TAB_LINE1_B + 16 * 2 * COOP_N_LINE_TERMS)
#define COOP_TAB_CSQR_B (COOP_TAB_LINE2_B + 16 * 2 * COOP_N_LINE_TERMS)
#define COOP_TAB_SQR_B (COOP_TAB_CSQR_B + 16 * 2 * COOP_N_CSQR_TERMS)

struct Coop {
    int grp_off;  // dword offset of this group's slots in coop_lds
    int g;        // coefficient role 0..15 (lane & 15)
    int h;        // staging role (lane >> 4) & 1: the h = 0 lanes stage the A-side operands, the h = 1 lanes the B side
    // Which terms of a coefficient's sum the lane takes: t = q, q + nq, ...  Normal kernel: two lanes per coefficient
    // (nq = 2, q = h), two proofs per wave.  WIDE kernel (one proof per wave: the single pairing of the RLC batch mode and
    // batches too small to give every SIMD a wave otherwise): four lanes per coefficient (nq = 4, q = h + 2 * (lane >> 5)),
    // the upper half-wave mirroring the lower one's staging; the four reduced partial sums meet through
    // v_permlane16_swap and v_permlane32_swap.  Per engine call a lane then multiplies 3 / 2 / 1 terms instead of 6 / 4 / 2.
    // NARROW kernel (four proofs per wave, for callers that keep the chip full with several batches in flight: 21 % fewer
    // instructions per pairing, twice the chain): ONE lane per coefficient (nq = 1, q = 0), no exchange; every lane plays
    // both staging roles (both = true, h = 0).
    int q, nq;
    bool both;
    // TWELVE kernel (five proofs per wave): the narrow engine without its four idle lanes per proof - a group is 12 lanes, one per
    // coefficient; lanes 60..63 of the wave play spare lanes of the last group and never store.  No lane is left over for the next
    // line's products (-lambda) xP: lanes 0..7 of a group take the eight of a round in a pass of their own before its two line steps.
    bool twelve;
};

H2V_DI uint32_t *coop_slot(const Coop &c, int s) {
    const int off = s < COOP_SHARED_BASE ? c.grp_off + s * COOP_SLOT_DW : COOP_SHR_OFF + (s - COOP_SHARED_BASE) * COOP_SLOT_DW;
    return coop_lds + off;
}
H2V_DI void coop_store28(uint32_t *p, const F28 &a) {   // limbs < 2^28 (carried), except where the headroom note below allows 2^29
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(a.l[0], a.l[1], a.l[2], a.l[3]);
    q[1] = make_uint4(a.l[4], a.l[5], a.l[6], a.l[7]);
    q[2] = make_uint4(a.l[8], a.l[9], a.l[10], a.l[11]);
    q[3] = make_uint4(a.l[12], a.l[13], 0u, 0u);
}
H2V_DI void coop_store28(uint32_t *p, const Fp &a) {
    F28 t;
    f28_from_fp(t, a);
    coop_store28(p, t);
}
H2V_DI void coop_load28(uint32_t (&l)[14], const uint32_t *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    l[0] = a.x; l[1] = a.y; l[2] = a.z; l[3] = a.w; l[4] = b.x; l[5] = b.y; l[6] = b.z; l[7] = b.w;
    l[8] = c.x; l[9] = c.y; l[10] = c.z; l[11] = c.w; l[12] = d.x; l[13] = d.y;
}
// Both operands of one term with 128-bit LDS reads.  Written as one asm block because the optimiser re-cuts uint4 LDS loads
// whose slot address it cannot see into two-dword reads (24 ds_read2_b32 per term instead of 6 ds_read_b128 + 2 ds_read_b64);
// slots are 80-byte aligned records (COOP_SLOT_DW = 20), so every 16-byte piece is aligned.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
H2V_DI void coop_load28_pair(uint32_t (&x)[14], uint32_t (&y)[14], const uint32_t *px, const uint32_t *py) {
    const uint32_t ax = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t *)px;
    const uint32_t ay = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t *)py;
    u32x4_t x0, x1, x2, y0, y1, y2;
    u32x2_t x3, y3;
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b64 %3, %8 offset:48\n\t"
                 "ds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:16\n\tds_read_b128 %6, %9 offset:32\n\tds_read_b64 %7, %9 offset:48\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3)
                 : "v"(ax), "v"(ay)
                 : "memory");
    x[0] = x0.x; x[1] = x0.y; x[2] = x0.z; x[3] = x0.w; x[4] = x1.x; x[5] = x1.y; x[6] = x1.z; x[7] = x1.w;
    x[8] = x2.x; x[9] = x2.y; x[10] = x2.z; x[11] = x2.w; x[12] = x3.x; x[13] = x3.y;
    y[0] = y0.x; y[1] = y0.y; y[2] = y0.z; y[3] = y0.w; y[4] = y1.x; y[5] = y1.y; y[6] = y1.z; y[7] = y1.w;
    y[8] = y2.x; y[9] = y2.y; y[10] = y2.z; y[11] = y2.w; y[12] = y3.x; y[13] = y3.y;
}

// Values between engine calls are lazily reduced F28 elements (h2v_fp28.hpp).  Bounds, in that header's (v, lam)
// notation:  every Fp12 variable has lam = 1 and v <= 6 (engine outputs 3, conj 6, frob 5, inverse 3; the
// generated program is checked for this by tools/gen_coop_program.py).  Accumulator headroom: a lane's column sums
// (terms per lane) x 14 products of (lam_x lam_y) 2^56, tripled for the cyclotomic squaring, plus 14 reduction products:
// the weighted product count must stay below 2^8.  MUL: 6 x 14 = 84 with carried operands (lam = 1); the doubled operand
// D = 2a of the squarings is stored UNCARRIED (lam = 2): SQR 4 x 14 x 2 = 112, CSQR 2 x 14 x 2 x 3 = 168 (no CSQR term has two uncarried operands: D and ND2 only meet carried ones).  Staged operands:
//   A = a (6)   NA = 7p - a_im (7)   B = b (6)   XB = (b0 - b1 + 7p, b0 + b1) (13)   D = 2a (12)   (squarings: see coop_csqr / coop_sqr)
// Each of the two lanes sharing a coefficient reduces its half of the terms on its own: a half is below
// (6 * 7 * 13 / 2520 + 1) p = 1.22 p for MUL and (3 * 2 * 168 / 2520 + 1) p = 1.4 p for the tripled cyclotomic
// squaring (p / R < 1/2520), so the engine's result (the sum of the two halves) is below 3p.

// out = sum_t X[tab[2t]] * Y[tab[2t+1]]  (mod p), one Montgomery reduction.  NT <= 12 (accumulator headroom).
// TRIPLE: the column accumulators are multiplied by 3 before the reduction (6 terms * 3 still fits 64 bits).
// NQ: lanes per coefficient (Coop::nq) as a compile-time constant - one instantiation per engine (narrow / normal / wide), so
// that the term loop, the exchange and the tripling carry no wave-uniform branches.
template <int NT, bool TRIPLE, int NQ>
H2V_DN F28Regs coop_accumulate(const Coop c, const int tab_row_byte) {
    const uint8_t *tab = reinterpret_cast<const uint8_t *>(coop_lds + COOP_TAB_OFF) + tab_row_byte;
    static_assert(NT <= 12, "column accumulators hold at most 12 unreduced products");
    static_assert(!TRIPLE || NT <= 6, "tripling needs a factor 3 of headroom");
    static_assert((NT & 1) == 0 && NT >= 4, "terms are split over two or four lanes, each with at least one");
    uint64_t acc[28];
    {   // first term initialises the columns (no zero-fill of 56 registers)
        uint32_t x[14], y[14];
        coop_load28_pair(x, y, coop_slot(c, tab[2 * (NQ == 1 ? 0 : c.q)]), coop_slot(c, tab[2 * (NQ == 1 ? 0 : c.q) + 1]));
#pragma unroll
        for (int i = 0; i < 14; i++)
#pragma unroll
            for (int j = 0; j < 14; j++) {
                if (i == 0 || j == 13) acc[i + j] = (uint64_t)x[i] * y[j];
                else acc[i + j] += (uint64_t)x[i] * y[j];
            }
        acc[27] = 0;
    }
    // (one lane per coefficient, cyclotomic squaring: the row's LAST term - the constant -/+ 2/3 against the lane's own coefficient -
    //  is left out; coop_csqr forms 3 r -/+ 2 g itself and folds it, a product cheaper)
    constexpr int NTE = (NQ == 1 && TRIPLE) ? NT - 1 : NT;
#pragma unroll 1
    for (int t = (NQ == 1 ? 0 : c.q) + NQ; t < NTE; t += NQ) {
        uint32_t x[14], y[14];
        coop_load28_pair(x, y, coop_slot(c, tab[2 * t]), coop_slot(c, tab[2 * t + 1]));
#pragma unroll
        for (int i = 0; i < 14; i++)
#pragma unroll
            for (int j = 0; j < 14; j++) acc[i + j] += (uint64_t)x[i] * y[j];
    }
    if (TRIPLE && NQ >= 2) {   // x3 as one shift-add per column (the compiler's choice was two v_mad_u64_u32 per column)
#pragma unroll
        for (int i = 0; i < 28; i++) {
            uint64_t t3;
            asm("v_lshl_add_u64 %0, %1, 1, %1" : "=v"(t3) : "v"(acc[i]));
            acc[i] = t3;
        }
    }
    // Montgomery reduction of this lane's half of the sum (operand scanning), R = 2^392
#pragma unroll
    for (int k = 0; k < 14; k++) {
        const uint32_t m = ((uint32_t)acc[k] * FP_N0_28) & FP28_MASK;
#pragma unroll
        for (int j = 0; j < 14; j++) acc[k + j] += (uint64_t)m * FP_MOD28[j];
        acc[k + 1] += acc[k] >> 28;
    }
    F28 r;
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 13; k++) {
        carry += acc[14 + k];
        r.l[k] = (uint32_t)carry & FP28_MASK;
        carry >>= 28;
    }
    carry += acc[27];
    r.l[13] = (uint32_t)carry;
    // The reduction is linear, so the two lanes of a coefficient (lane ^ 16) reduce their halves separately and
    // exchange 14 reduced limbs instead of 28 64-bit columns.  v_permlane16_swap exchanges the odd 16-lane rows of
    // its first operand with the even rows of its second: called on two copies of a register, one copy ends up
    // holding (own | partner) and the other (partner | own) by row, so their sum is own + partner in every lane.
    if (NQ >= 2) {
#pragma unroll
        for (int i = 0; i < 14; i++) {
            const auto sw = __builtin_amdgcn_permlane16_swap(r.l[i], r.l[i], false, false);
            r.l[i] = sw[0] + sw[1];
        }
    }
    if (NQ == 4) {   // the other half-wave holds the sum of the other two quarters
#pragma unroll
        for (int i = 0; i < 14; i++) {
            const auto sw = __builtin_amdgcn_permlane32_swap(r.l[i], r.l[i], false, false);
            r.l[i] = sw[0] + sw[1];
        }
    }
    // value: two halves < 2 * 1.4 p; four quarters < 4 * (3 * 168 / 2520 + 1) p = 4.8 p; one lane: (12 * 7 * 13 / 2520 + 1) p =
    // 1.43 p (MUL), 1.56 p (SQR), 3.8 p (CSQR, above) (every operand bound of the staging code assumes v <= 6, CONJ and INV
    // v <= 5); limbs back below 2^28
    if (NQ >= 2) f28_carry(r);   // (one lane per coefficient: the limbs left the extraction carried)
    return f28_pack(r);
}
template <int NT, bool TRIPLE>
H2V_DI F28 coop_engine(const Coop &c, const int tab_row_byte) {
    const F28Regs z = c.nq == 1 ? coop_accumulate<NT, TRIPLE, 1>(c, tab_row_byte)     // (wave-uniform: a kernel only ever takes one of these)
                      : c.nq == 2 ? coop_accumulate<NT, TRIPLE, 2>(c, tab_row_byte) : coop_accumulate<NT, TRIPLE, 4>(c, tab_row_byte);
    return f28_unpack(z.a, z.b, z.c, z.d);
}

// one product of two staged slots, reduced (< 2p): the twelve-lane kernel's line products
H2V_DN F28Regs coop_prod(const Coop c, const int xs, const int ys) {
    uint32_t x[14], y[14];
    coop_load28_pair(x, y, coop_slot(c, xs), coop_slot(c, ys));
    uint64_t acc[28];
#pragma unroll
    for (int i = 0; i < 14; i++)
#pragma unroll
        for (int j = 0; j < 14; j++) {
            if (i == 0 || j == 13) acc[i + j] = (uint64_t)x[i] * y[j];
            else acc[i + j] += (uint64_t)x[i] * y[j];
        }
    acc[27] = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) {
        const uint32_t m = ((uint32_t)acc[k] * FP_N0_28) & FP28_MASK;
#pragma unroll
        for (int j = 0; j < 14; j++) acc[k + j] += (uint64_t)m * FP_MOD28[j];
        acc[k + 1] += acc[k] >> 28;
    }
    F28 r;
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 13; k++) {
        carry += acc[14 + k];
        r.l[k] = (uint32_t)carry & FP28_MASK;
        carry >>= 28;
    }
    r.l[13] = (uint32_t)(carry + acc[27]);
    return f28_pack(r);
}
H2V_DI F28 coop_shfl_xor1(const F28 &a) {
    F28 r;
#pragma unroll
    for (int i = 0; i < 14; i++)   // quad_perm [1, 0, 3, 2]: a DPP move at VALU rate (the generic shuffle is an LDS round trip per limb)
    {
        r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.l[i], 0xB1, 0xf, 0xf, false);
        asm volatile("" : "+v"(r.l[i]));   // keep the move a move: folded into the consuming add / sub (v_subrev_u32_dpp vN, vN, vM) the
    }                                      // quad-cooperative mixed addition of h2v_curve28.hpp came out wrong in lane 0 (ROCm 7.2)
    return r;
}
// stage a distributed value (v <= 6) as the A operand (a_{k,part}, and -a_{k,1})
H2V_DI void coop_stage_a(const Coop &c, const F28 &a) {
    if (c.g < 12 && c.h == 0) {   // (narrow kernel: h = 0 in every lane)
        coop_store28(coop_slot(c, COOP_SLOT_A + c.g), a);
        if (c.g & 1) {
            F28 n;
            F28_NEG(n, a, 7, 1);
            f28_carry(n);
            coop_store28(coop_slot(c, COOP_SLOT_NA + (c.g >> 1)), n);
        }
    }
}
// stage a distributed value (v <= 6) as the B operand (b and xi*b)
H2V_DI void coop_stage_b(const Coop &c, const F28 &b) {
    const F28 pb = coop_shfl_xor1(b);
    if (c.g < 12 && (c.both || c.h == 1)) {
        coop_store28(coop_slot(c, COOP_SLOT_B + c.g), b);
        F28 xb, t;
        F28_NEG(t, pb, 7, 1);                  // 7p - b1
        if (c.g & 1) t = pb;                   // imaginary part: (xi b)_1 = b0 + b1 ; real part: (xi b)_0 = b0 - b1
        f28_add(xb, b, t);
        f28_carry(xb);
        coop_store28(coop_slot(c, COOP_SLOT_XB + c.g), xb);
    }
}
// c = a * b (all distributed)
H2V_DI F28 coop_mul(const Coop &c, const F28 &a, const F28 &b) {
    coop_stage_a(c, a);
    coop_stage_b(c, b);
    __syncthreads();
    const F28 r = coop_engine<COOP_N_MUL_TERMS, false>(c, COOP_TAB_MUL_B + c.g * 2 * COOP_N_MUL_TERMS);
    __syncthreads();
    return r;
}
// a^2 for a in the cyclotomic subgroup (Granger-Scott; formulas and table: tools/gen_coop_tables.py: csqr_table).
// Operands: A = a (6), NA = 7p - a_im (7), ND2 = 2 NA_2 (14, uncarried) from the half-0 lanes; from the half-1 lanes
// D = 2a (12, uncarried) and, per Fp2 coefficient, the sum S = a_re + a_im (12) and the difference M = a_re - a_im + 7p
// (13) (Re(x^2) = S M is one product instead of two); and the shared constants +-2/3: the engine returns
// 3 (Q_k -/+ (2/3) a_k) = 3 Q_k -/+ 2 a_k already reduced.  Four terms per coefficient, two per lane; a half-sum is
// below (3 * 2 * 168 / 2520 + 1) p = 1.4 p (largest product of operand bounds: ND2 x S = 14 x 12).
// Staging is ONE instruction stream for both half-waves (lanes of a wave that take different branches run them one
// after the other): both halves hold a and its Fp2 partner, every lane forms 7p - a, the half-0 lanes carry and store
// that (NA; odd g), the half-1 lanes carry and store partner + (a or 7p - a) (S / M), and the uncarried first store is
// a << h (A or D).
H2V_DI void coop_csqr_stage(const Coop &c, const F28 &a, const F28 &pa, const F28 &neg, const bool hi) {
    const bool odd = (c.g & 1) != 0;
    F28 v, u;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const uint32_t t = odd ? neg.l[i] : a.l[i];      // imaginary-part lane: M = re - im = pa + (7p - a); real: S = a + pa
        v.l[i] = hi ? pa.l[i] + t : neg.l[i];
        u.l[i] = a.l[i] << (hi ? 1 : 0);       // A = a (6, 1) | D = 2a (12, 2): limbs below 2^29 - see the headroom note at the engine
    }
    f28_carry(v);                              // NA (7, 1) | S (12, 1) / M (13, 1)
    if (c.g < 12) {
        coop_store28(coop_slot(c, hi ? COOP_SLOT_D + c.g : COOP_SLOT_A + c.g), u);
        if (hi || odd) coop_store28(coop_slot(c, hi ? COOP_SLOT_SM + c.g : COOP_SLOT_NA + (c.g >> 1)), v);
        if (!hi && c.g == 5) {                 // ND2 = 2 NA_2 = -2 a_21                    (14, 2)
            F28 nd;
            f28_mul_small<2>(nd, v);
            coop_store28(coop_slot(c, COOP_SLOT_ND2), nd);
        }
    }
}
H2V_DI F28 coop_csqr(const Coop &c, const F28 &a) {
    const F28 pa = coop_shfl_xor1(a);          // the other part of the same Fp2 coefficient
    F28 neg;
    F28_NEG(neg, a, 7, 1);                     // 7p - a                                   (7, 3)
    if (c.both) {                              // narrow kernel: the lane plays both roles, one after the other
        coop_csqr_stage(c, a, pa, neg, false);
        coop_csqr_stage(c, a, pa, neg, true);
    } else {
        coop_csqr_stage(c, a, pa, neg, c.h != 0);
    }
    __syncthreads();
    F28 r = coop_engine<COOP_N_CSQR_TERMS, true>(c, COOP_TAB_CSQR_B + c.g * 2 * COOP_N_CSQR_TERMS);
    __syncthreads();
    if (c.nq == 1) {
        // one lane per coefficient: r = Q_k reduced (< 1.2p); h_k = 3 Q_k - 2 g_k (even k) / + 2 g_k (odd k) below 17p, folded below 2p
        // (h2v_fp28.hpp: f28_fold; the engines with several lanes per coefficient take the constant product and tripled columns)
        const bool minus = ((c.g >> 1) & 1) == 0;
#pragma unroll
        for (int i = 0; i < 14; i++) {
            const uint32_t g2 = a.l[i] << 1;
            r.l[i] = 3u * r.l[i] + (minus ? F28_BIAS_13_2[i] - g2 : g2);
        }
        f28_carry(r);
        f28_fold(r);
    }
    return r;
}
// a^2 for a general a (the Miller loop's f^2): 8 terms per coefficient instead of the 12 of coop_mul(a, a)
// (tools/gen_coop_tables.py: sqr_table).  Operands: A = a (6), NA = 7p - a_im (7) from the half-0 lanes; from the
// half-1 lanes D = 2a (12), XD = xi D = (d0 - d1 + 13p, d0 + d1) (25) and, for k = 3..5, XA = xi a = (a0 - a1 + 7p,
// a0 + a1) (13).  A half-sum is below (4 * 7 * 25 / 2520 + 1) p = 1.28 p, so the result is below 3p like a product's.
H2V_DI F28 coop_sqr(const Coop &c, const F28 &a) {
    coop_stage_a(c, a);
    const F28 pa = coop_shfl_xor1(a);          // the other part of the same Fp2 coefficient
    if (c.g < 12 && (c.both || c.h == 1)) {
        F28 d2, pd2, t, xd;
        f28_mul_small<2>(d2, a);               // (12, 2): stored with limbs below 2^29
        coop_store28(coop_slot(c, COOP_SLOT_D + c.g), d2);
        f28_mul_small<2>(pd2, pa);             // (12, 2)
        F28_NEG(t, pd2, 13, 2);                // 13p - d1                                 (13, 4)
        if (c.g & 1) t = pd2;                  // imaginary part: (xi d)_1 = d0 + d1 ; real part: (xi d)_0 = d0 - d1
        f28_add(xd, d2, t);
        f28_carry(xd);
        coop_store28(coop_slot(c, COOP_SLOT_XD + c.g), xd);
        if (c.g >= 6) {
            F28 xa;
            F28_NEG(t, pa, 7, 1);
            if (c.g & 1) t = pa;
            f28_add(xa, a, t);
            f28_carry(xa);
            coop_store28(coop_slot(c, COOP_SLOT_XA + (c.g - 6)), xa);
        }
    }
    __syncthreads();
    const F28 r = coop_engine<COOP_N_SQR_TERMS, false>(c, COOP_TAB_SQR_B + c.g * 2 * COOP_N_SQR_TERMS);
    __syncthreads();
    return r;
}
H2V_DI F28 coop_conj(const Coop &c, const F28 &a) {  // w -> -w: odd powers change sign.  a: v <= 5
    F28 r = a;
    if (c.g < 12 && ((c.g >> 1) & 1)) {
        F28_NEG(r, a, 6, 1);
        f28_carry(r);
    }
    return r;
}
// a -> a^p: coefficient k becomes conj(a_k) * gamma^k.  Result v <= 5.
H2V_DI F28 coop_frob(const Coop &c, const F28 &a) {
    const F28 pa = coop_shfl_xor1(a);
    F28 r = a;
    if (c.g < 12) {
        const int k = c.g >> 1;
        Fp g0, g1;
#pragma unroll
        for (int i = 0; i < 12; i++) { g0.v[i] = FROB_GAMMA[k][0][i]; g1.v[i] = FROB_GAMMA[k][1][i]; }
        F28 h0, h1, x, y;
        f28_from_fp(h0, g0);
        f28_from_fp(h1, g1);
        if (c.g & 1) {   // imaginary part: a_k0*g1 - a_k1*g0
            f28_mul(x, pa, h1); f28_mul(y, a, h0); F28_SUB(r, x, y, 3, 1);
        } else {         // real part: a_k0*g0 + a_k1*g1
            f28_mul(x, a, h0); f28_mul(y, pa, h1); f28_add(r, x, y);
        }
        f28_carry(r);
    }
    return r;
}
// copy one line's 8 constant slots (512 B) from the plan into the wave-shared area
H2V_DI void coop_stage_line(const Coop &c, int shared_slot, const uint32_t *lines28, int line_idx, int lane) {
    // plan: 8 slots x 16 dwords per line; lane copies 2 dwords: slot = lane / 8, dwords 2*(lane % 8) ..
    const uint2 v = reinterpret_cast<const uint2 *>(lines28 + (size_t)line_idx * 8 * 16)[lane];
    uint32_t *dst = coop_lds + COOP_SHR_OFF + (shared_slot - COOP_SHARED_BASE + (lane >> 3)) * COOP_SLOT_DW + 2 * (lane & 7);
    dst[0] = v.x;
    dst[1] = v.y;
}
// f <- f * line (loop 1: el against s_g2's lines; loop 2: -er against G2's lines).  The spare lanes' products go
// to the OTHER loop's T slots (they belong to that loop's next line).
template <int LOOP>
H2V_DI F28 coop_line(const Coop &c, const F28 &f) {
    coop_stage_a(c, f);
    __syncthreads();
    const F28 r = coop_engine<COOP_N_LINE_TERMS, false>(c, (LOOP == 1 ? COOP_TAB_LINE1_B : COOP_TAB_LINE2_B) + c.g * 2 * COOP_N_LINE_TERMS);
    __syncthreads();
    if (c.g >= 12 && c.h == 0 && !c.twelve) coop_store28(coop_slot(c, (LOOP == 1 ? COOP_SLOT_T2 : COOP_SLOT_T1) + (c.g - 12)), r);
    return r;
}
// 1/f for a distributed f: N = f * conj(f) lies in Fp6 (even powers of w); lane 0 inverts it with the tower code
// on canonical limbs.
H2V_DN F28Regs coop_inv_raw(const Coop c, const F28Regs fr, bool &ok) {
    const F28 f = f28_unpack(fr.a, fr.b, fr.c, fr.d);
    const F28 fc = coop_conj(c, f);
    const F28 nrm28 = coop_mul(c, f, fc);
    Fp nrm;
    f28_to_fp(nrm, nrm28);
    // gather the even coefficients on lane 0 through the (now free) A slots, raw 12 x 32 limbs
    if (c.g < 12 && c.h == 0) {
        uint32_t *p = coop_slot(c, COOP_SLOT_A + c.g);
#pragma unroll
        for (int i = 0; i < 12; i++) p[i] = nrm.v[i];
    }
    __syncthreads();
    bool good = true;
    if (c.g == 0 && c.h == 0) {
        Fp6 n6, inv6;
        const int src[6] = {0, 1, 4, 5, 8, 9};  // (k=0: c0), (k=2: c1), (k=4: c2)
        Fp *dst[6] = {&n6.c0.c0, &n6.c0.c1, &n6.c1.c0, &n6.c1.c1, &n6.c2.c0, &n6.c2.c1};
        for (int q = 0; q < 6; q++) {
            const uint32_t *p = coop_slot(c, COOP_SLOT_A + src[q]);
#pragma unroll
            for (int i = 0; i < 12; i++) dst[q]->v[i] = p[i];
        }
        good = fp6_inv(inv6, n6);
        const Fp *res[6] = {&inv6.c0.c0, &inv6.c0.c1, &inv6.c1.c0, &inv6.c1.c1, &inv6.c2.c0, &inv6.c2.c1};
        for (int q = 0; q < 6; q++) {
            uint32_t *p = coop_slot(c, COOP_SLOT_A + src[q]);
#pragma unroll
            for (int i = 0; i < 12; i++) p[i] = res[q]->v[i];
        }
    }
    __syncthreads();
    Fp ninv;
    fp_set_zero(ninv);
    if (c.g < 12 && !((c.g >> 1) & 1)) {
        const uint32_t *p = coop_slot(c, COOP_SLOT_A + c.g);
#pragma unroll
        for (int i = 0; i < 12; i++) ninv.v[i] = p[i];
    }
    __syncthreads();
    ok = good;
    F28 ninv28;
    f28_from_fp(ninv28, ninv);
    return f28_pack(coop_mul(c, fc, ninv28));
}
H2V_DI F28 coop_inv(const Coop &c, const F28 &f, bool &ok) {
    const F28Regs z = coop_inv_raw(c, f28_pack(f), ok);
    return f28_unpack(z.a, z.b, z.c, z.d);
}

// dbg (optional): per proof 2 x 12 Fp (canonical, 12 dwords each): f after the Miller loop, f after the final
// exponentiation; flat order (k, part).
// The kernel interprets COOP_PROGRAM (coop_program.h: 34 steps since the Miller loop and the exponentiations by x are single
// steps; generated and simulated against the
// big-integer pairing by tools/gen_coop_program.py).  Fp12 variables live in a private array, so no vector state is
// live across the engine call (the first version kept them in VGPRs and spent 65 % of its wave-cycles waiting on
// the spills around every call).
template <bool WIDE, bool NARROW = false, bool TWELVE = false>
H2V_DI void pairing_coop_body(const H2vDevPlan &plan, uint32_t n, const uint32_t *__restrict__ pts, const uint8_t *__restrict__ valid, const uint8_t *__restrict__ valid_sub,
                              const uint32_t *__restrict__ er_jac, const uint32_t *__restrict__ el_jac /* folded el (recursion) or NULL */,
                              uint32_t *__restrict__ status, uint8_t *__restrict__ accept, uint32_t *__restrict__ dbg, const uint32_t bid,
                              bool *verdict_out = nullptr /* WIDE: the verdict, in every lane */) {
    static_assert(!(WIDE && NARROW), "one proof per wave, or four");
    static_assert(!TWELVE || NARROW, "the twelve-lane kernel is the narrow engine, packed");
    const int lane = threadIdx.x;
    const int grp12 = (lane * 43) >> 9;     // lane / 12 for lane < 64
    const int grp = WIDE ? 0 : TWELVE ? (grp12 < COOP_GROUPS_TWELVE ? grp12 : COOP_GROUPS_TWELVE - 1) : NARROW ? lane >> 4 : lane >> 5;
    Coop c;
    c.g = TWELVE ? (grp12 < COOP_GROUPS_TWELVE ? lane - 12 * grp12 : 12 + (lane - 12 * COOP_GROUPS_TWELVE)) : lane & 15;
    c.h = NARROW ? 0 : (lane >> 4) & 1;
    c.both = NARROW;
    c.twelve = TWELVE;
    c.nq = WIDE ? 4 : NARROW ? 1 : 2;
    c.q = WIDE ? c.h + 2 * (lane >> 5) : c.h;
    c.grp_off = COOP_GRP_OFF + grp * COOP_GROUP_DW;
    const int leader = TWELVE ? grp * 12 : NARROW ? grp * 16 : grp * 32;  // lane (g = 0, h = 0) of the group
    const bool is_leader = lane == leader;
    const uint32_t i = WIDE ? bid : bid * (TWELVE ? COOP_GROUPS_TWELVE : NARROW ? COOP_GROUPS_NARROW : COOP_GROUPS_PER_WAVE) + grp;
    const bool live = i < n;
    const uint32_t ii = live ? i : n - 1;  // dead groups shadow the last proof, never write
    const uint32_t slots = H2V_SLOTS(plan);

    // operand tables -> LDS
    {
        const uint32_t *t0 = reinterpret_cast<const uint32_t *>(&COOP_TAB_MUL[0][0]);
        const uint32_t *t1 = reinterpret_cast<const uint32_t *>(&COOP_TAB_LINE1[0][0]);
        const uint32_t *t2 = reinterpret_cast<const uint32_t *>(&COOP_TAB_LINE2[0][0]);
        const uint32_t *t3 = reinterpret_cast<const uint32_t *>(&COOP_TAB_CSQR[0][0]);
        constexpr int n0 = 16 * 2 * COOP_N_MUL_TERMS / 4, n1 = 16 * 2 * COOP_N_LINE_TERMS / 4, n3 = 16 * 2 * COOP_N_CSQR_TERMS / 4;
        for (int q = lane; q < n0; q += 64) coop_lds[COOP_TAB_OFF + q] = t0[q];
        for (int q = lane; q < n1; q += 64) { coop_lds[COOP_TAB_OFF + n0 + q] = t1[q]; coop_lds[COOP_TAB_OFF + n0 + n1 + q] = t2[q]; }
        for (int q = lane; q < n3; q += 64) coop_lds[COOP_TAB_OFF + n0 + 2 * n1 + q] = t3[q];
        const uint32_t *t4 = reinterpret_cast<const uint32_t *>(&COOP_TAB_SQR[0][0]);
        constexpr int n4 = 16 * 2 * COOP_N_SQR_TERMS / 4;
        for (int q = lane; q < n4; q += 64) coop_lds[COOP_TAB_OFF + n0 + 2 * n1 + n3 + q] = t4[q];
    }
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
        for (int k = 0; k < 12; k++) {
            el.x.v[k] = pp[k]; el.y.v[k] = pp[12 + k];
            ej.x.v[k] = er_jac[(size_t)ii * 36 + k]; ej.y.v[k] = er_jac[(size_t)ii * 36 + 12 + k]; ej.z.v[k] = er_jac[(size_t)ii * 36 + 24 + k];
        }
        if (st != 0) { g1a_set_inf(el); g1j_set_inf(ej); }  // rejected already: keep the arithmetic well-defined
        g1j_to_affine(er, ej);
        if (el_jac && st == 0) {
            G1J lj;
#pragma unroll
            for (int k = 0; k < 12; k++) { lj.x.v[k] = el_jac[(size_t)ii * 36 + k]; lj.y.v[k] = el_jac[(size_t)ii * 36 + 12 + k]; lj.z.v[k] = el_jac[(size_t)ii * 36 + 24 + k]; }
            g1j_to_affine(el, lj);
        }
        if (g1a_is_inf(el)) flags |= 1;
        if (g1a_is_inf(er)) flags |= 2;
        fp_neg(er.y, er.y);
        coop_store28(coop_slot(c, COOP_SLOT_PX1), el.x);
        coop_store28(coop_slot(c, COOP_SLOT_PY1), el.y);
        coop_store28(coop_slot(c, COOP_SLOT_PX2), er.x);
        coop_store28(coop_slot(c, COOP_SLOT_PY2), er.y);
        Fp z;
        fp_set_zero(z);
        coop_store28(coop_slot(c, COOP_SLOT_ZERO), z);
    }
    flags = __shfl(flags, leader);
    st = __shfl(st, leader);
    const bool skip1 = (flags & 1) != 0, skip2 = (flags & 2) != 0;
    bool inv_ok = true;

    if (lane < 14) {   // the two shared constants of the cyclotomic squaring
        coop_lds[COOP_SHR_OFF + (COOP_SLOT_C23P - COOP_SHARED_BASE) * COOP_SLOT_DW + lane] = FP_C23P28[lane];
        coop_lds[COOP_SHR_OFF + (COOP_SLOT_C23N - COOP_SHARED_BASE) * COOP_SLOT_DW + lane] = FP_C23N28[lane];
    }
    F28 vars[COOP_N_VARS];
    for (int pc = 0; pc < COOP_PROGRAM_LEN; pc++) {
        const uint32_t ins = COOP_PROGRAM[pc];
        const int op = ins & 0xff, d = (ins >> 8) & 0xff, a = (ins >> 16) & 0xff, b = ins >> 24;
        if (op == COOP_OP_END) break;
        switch (op) {
        case COOP_OP_MUL: {
            const F28 x = vars[a], y = vars[b];
            vars[d] = coop_mul(c, x, y);
        } break;
        case COOP_OP_CSQR: {   // b squarings in a row: the value stays in registers between them
            F28 x = vars[a];
#pragma unroll 1
            for (int rep = 0; rep < b; rep++) x = coop_csqr(c, x);
            vars[d] = x;
        } break;
        case COOP_OP_MSTEP: {
            // One Miller-loop step on F (kept in registers throughout): F = F^2, then d times (line of loop 1, line of
            // loop 2) from line index a.  Invariants (see coop_tables.h / gen_coop_tables.py): before LINE1(n) the
            // shared slots hold LN1(n) and T1(n); LINE1's spare lanes produce T2(n) from LN2(n); LINE2's spare lanes
            // produce T1(n+1) from LN1(n+1).
            F28 f = vars[COOP_VAR_F];
            f = coop_sqr(c, f);
#pragma unroll 1
            for (int ln = a; ln < a + d; ln++) {
                if (ln > 0) coop_stage_line(c, COOP_SLOT_LN2, plan.lines28_g2, ln, lane);
                __syncthreads();
                const F28 r1 = coop_line<1>(c, f);
                if (!skip1 && c.g < 12) f = r1;
                if (ln + 1 < H2V_MILLER_LINES) coop_stage_line(c, COOP_SLOT_LN1, plan.lines28_sg2, ln + 1, lane);
                __syncthreads();
                const F28 r2 = coop_line<2>(c, f);
                if (!skip2 && c.g < 12) f = r2;
            }
            vars[COOP_VAR_F] = f;
        } break;
        case COOP_OP_MILLER: {
            // The whole Miller loop as one step: F never leaves the registers (as 63 MSTEPs it went through the variable
            // file - scratch - 126 times).  Per bit of |x| below the leading one: F = F^2, then one (doubling) or two
            // (doubling + addition) rounds of { line of loop 1, line of loop 2 }; invariants as for MSTEP.
            F28 f = vars[COOP_VAR_F];
            int ln = 0;
#pragma unroll 1
            for (int bit = 62; bit >= 0; bit--) {
                f = coop_sqr(c, f);
                const int steps = ((BLS_X_ABS >> bit) & 1) ? 2 : 1;
#pragma unroll 1
                for (int s2 = 0; s2 < steps; s2++, ln++) {
                    if (TWELVE) {
                        // both lines' constants, then the round's eight products (-lambda) xP in one pass: lane g < 8 takes
                        // constant g & 3 (nl0, nl1, nxl0, nxl1) of loop 1 + (g >> 2)
                        coop_stage_line(c, COOP_SLOT_LN1, plan.lines28_sg2, ln, lane);
                        coop_stage_line(c, COOP_SLOT_LN2, plan.lines28_g2, ln, lane);
                        __syncthreads();
                        {
                            const int lp = (c.g >> 2) & 1, which = c.g & 3;
                            const F28Regs z = coop_prod(c, (lp ? COOP_SLOT_LN2 : COOP_SLOT_LN1) + which, lp ? COOP_SLOT_PX2 : COOP_SLOT_PX1);
                            if (c.g < 8) coop_store28(coop_slot(c, (lp ? COOP_SLOT_T2 : COOP_SLOT_T1) + which), f28_unpack(z.a, z.b, z.c, z.d));
                        }
                        const F28 r1 = coop_line<1>(c, f);      // (its staging barrier covers the T slots)
                        if (!skip1 && c.g < 12) f = r1;
                        const F28 r2 = coop_line<2>(c, f);
                        if (!skip2 && c.g < 12) f = r2;
                        continue;
                    }
                    if (ln > 0) coop_stage_line(c, COOP_SLOT_LN2, plan.lines28_g2, ln, lane);
                    __syncthreads();
                    const F28 r1 = coop_line<1>(c, f);
                    if (!skip1 && c.g < 12) f = r1;
                    if (ln + 1 < H2V_MILLER_LINES) coop_stage_line(c, COOP_SLOT_LN1, plan.lines28_sg2, ln + 1, lane);
                    __syncthreads();
                    const F28 r2 = coop_line<2>(c, f);
                    if (!skip2 && c.g < 12) f = r2;
                }
            }
            vars[COOP_VAR_F] = f;
        } break;
        case COOP_OP_EXPX: {   // d = a^x (x < 0: conjugate of a^|x|), the running power in registers throughout
            F28 x = vars[a];
#pragma unroll 1
            for (int bit = 62; bit >= 0; bit--) {
                x = coop_csqr(c, x);
                if ((BLS_X_ABS >> bit) & 1) {
                    const F28 y = vars[a];
                    x = coop_mul(c, x, y);
                }
            }
            vars[d] = coop_conj(c, x);
        } break;
        case COOP_OP_WARMUP: {
            if (TWELVE) break;       // (every round takes its own products)
            coop_stage_line(c, COOP_SLOT_LN1, plan.lines28_sg2, 0, lane);
            coop_stage_line(c, COOP_SLOT_LN2, plan.lines28_g2, 0, lane);
            __syncthreads();
            (void)coop_line<2>(c, vars[COOP_VAR_F]);  // only its spare lanes matter: T1 of line 0
        } break;
        case COOP_OP_CONJ: vars[d] = coop_conj(c, vars[a]); break;
        case COOP_OP_FROB: vars[d] = coop_frob(c, vars[a]); break;
        case COOP_OP_INV: {
            bool ok = true;
            vars[d] = coop_inv(c, vars[a], ok);
            inv_ok = ok;
        } break;
        case COOP_OP_MOV: vars[d] = vars[a]; break;
        case COOP_OP_SETONE: {
            F28 o;
            f28_set_zero(o);
            if (c.g == 0) f28_set_one(o);
            vars[d] = o;
        } break;
        case COOP_OP_DUMP: {
            if (dbg && live && c.g < 12 && c.h == 0) {
                Fp o, oc;
                f28_to_fp(oc, vars[a]);
                fp_from_mont(o, oc);
#pragma unroll
                for (int k = 0; k < 12; k++) dbg[((size_t)i * 24 + 12 * d + c.g) * 12 + k] = o.v[k];
            }
        } break;
        default: break;
        }
    }
    // == 1 ?
    Fp res;
    f28_to_fp(res, vars[COOP_VAR_F]);
    bool mine = true;
    if (c.g == 0) { Fp one; fp_set_one(one); mine = fp_eq(res, one); }
    else if (c.g < 12) mine = fp_is_zero(res);
    const unsigned long long bal = __ballot(mine);
    const bool is_one = WIDE ? bal == ~0ull : TWELVE ? ((bal >> leader) & 0xfffull) == 0xfffull : NARROW ? ((bal >> leader) & 0xffffull) == 0xffffull : ((bal >> leader) & 0xffffffffull) == 0xffffffffull;
    inv_ok = __shfl((int)inv_ok, leader) != 0;
    if (is_leader && live) {
        if (st == 0 && !(is_one && inv_ok)) st |= H2V_ST_PAIRING;
        status[i] = st;
        accept[i] = st == 0 ? 1 : 0;
    }
    if (verdict_out) *verdict_out = st == 0 && is_one && inv_ok;   // (st, is_one, inv_ok are the leader's in every lane)
}
extern "C" __global__ void __launch_bounds__(64, 2)
k_pairing_coop(H2vDevPlan plan, uint32_t n, const uint32_t *__restrict__ pts, const uint8_t *__restrict__ valid, const uint8_t *__restrict__ valid_sub,
               const uint32_t *__restrict__ er_jac, const uint32_t *__restrict__ el_jac /* folded el (recursion) or NULL */,
               uint32_t *__restrict__ status, uint8_t *__restrict__ accept, uint32_t *__restrict__ dbg,
               const uint32_t *__restrict__ skip /* RLC mode: return at once when *skip != 0; NULL = always run */) {
    if (skip && skip[0]) return;
    if (!skip) { pairing_coop_body<false>(plan, n, pts, valid, valid_sub, er_jac, el_jac, status, accept, dbg, blockIdx.x); return; }
    // conditional launch (fall-back of the RLC batch mode): a small grid walks the logical blocks
    // skip[1 + g]: group g (proofs 64 g .. 64 g + 63) passed its own check (k_pairing_rlc_groups): its verdicts are final
    const uint32_t n_blocks = (n + COOP_GROUPS_PER_WAVE - 1) / COOP_GROUPS_PER_WAVE;
    for (uint32_t bid = blockIdx.x; bid < n_blocks; bid += gridDim.x) {
        if (skip[1 + ((bid * COOP_GROUPS_PER_WAVE) >> 6)]) continue;     // (COOP_GROUPS_PER_WAVE divides 64: one group per block)
        pairing_coop_body<false>(plan, n, pts, valid, valid_sub, er_jac, el_jac, status, accept, dbg, bid);
        __syncthreads();
    }
}
// one proof per wave, four lanes per coefficient: for launches that cannot give every SIMD a wave anyway
extern "C" __global__ void __launch_bounds__(64, 2)
k_pairing_coop_wide(H2vDevPlan plan, uint32_t n, const uint32_t *__restrict__ pts, const uint8_t *__restrict__ valid, const uint8_t *__restrict__ valid_sub,
                    const uint32_t *__restrict__ er_jac, const uint32_t *__restrict__ el_jac, uint32_t *__restrict__ status,
                    uint8_t *__restrict__ accept, uint32_t *__restrict__ dbg) {
    pairing_coop_body<true>(plan, n, pts, valid, valid_sub, er_jac, el_jac, status, accept, dbg, blockIdx.x);
}
// four proofs per wave, one lane per coefficient: fewest instructions per pairing, longest chain (see Coop)
extern "C" __global__ void __launch_bounds__(64, 2)
k_pairing_coop_narrow(H2vDevPlan plan, uint32_t n, const uint32_t *__restrict__ pts, const uint8_t *__restrict__ valid, const uint8_t *__restrict__ valid_sub,
                      const uint32_t *__restrict__ er_jac, const uint32_t *__restrict__ el_jac, uint32_t *__restrict__ status,
                      uint8_t *__restrict__ accept, uint32_t *__restrict__ dbg) {
    pairing_coop_body<false, true>(plan, n, pts, valid, valid_sub, er_jac, el_jac, status, accept, dbg, blockIdx.x);
}
// five proofs per wave: the narrow engine without its idle lanes (see Coop::twelve)
extern "C" __global__ void __launch_bounds__(64, 2)
k_pairing_coop_twelve(H2vDevPlan plan, uint32_t n, const uint32_t *__restrict__ pts, const uint8_t *__restrict__ valid, const uint8_t *__restrict__ valid_sub,
                      const uint32_t *__restrict__ er_jac, const uint32_t *__restrict__ el_jac, uint32_t *__restrict__ status,
                      uint8_t *__restrict__ accept, uint32_t *__restrict__ dbg) {
    pairing_coop_body<false, true, true>(plan, n, pts, valid, valid_sub, er_jac, el_jac, status, accept, dbg, blockIdx.x);
}
// The single pairing of the RLC batch mode with its epilogue fused in.  flags[0]: batch check passed -> accept[i] = good_i for
// the whole batch, and the kernels queued behind return at once; failed -> accept[] is left to them.  flags[1 + g] for the
// n_groups groups of 64 proofs start as the batch verdict; after a failed batch check k_pairing_rlc_groups replaces them
// with the groups' own verdicts (where that stage runs), and the per-proof kernels skip the groups that passed.
extern "C" __global__ void __launch_bounds__(64, 2)
k_pairing_rlc(H2vDevPlan plan, const uint32_t *__restrict__ pts1, const uint8_t *__restrict__ valid1, const uint32_t *__restrict__ er_jac,
              const uint32_t *__restrict__ el_jac, uint32_t *__restrict__ status1, uint8_t *__restrict__ accept1,
              uint32_t n_batch, const uint8_t *__restrict__ good, uint8_t *__restrict__ accept, uint32_t *__restrict__ flags,
              uint32_t n_groups, uint32_t *__restrict__ fail_ctr, uint32_t *__restrict__ stats /* [0] groups seen, [1] groups that failed (cumulative) or NULL */,
              uint32_t groups_follow /* the group stage runs behind a failed check and counts by itself */) {
    bool ok = false;
    pairing_coop_body<true>(plan, 1u, pts1, valid1, nullptr, er_jac, el_jac, status1, accept1, nullptr, 0u, &ok);
    if (threadIdx.x == 0) flags[0] = ok ? 1u : 0u;
    if (threadIdx.x == 0 && stats && (ok || !groups_follow)) {       // what the workspace routes its next calls by (h2v_capi.hip: rlc_route)
        atomicAdd(stats, n_groups);
        if (!ok) atomicAdd(stats + 1, n_groups);
    }
    for (uint32_t g = threadIdx.x; g < n_groups; g += 64) flags[1 + g] = ok ? 1u : 0u;
    // (laned calls: one counter per call, shared by its chunks - how many batch checks of the call failed)
    if (threadIdx.x == 0 && !ok && fail_ctr) atomicAdd(fail_ctr, 1u);
    if (ok)
        for (uint32_t i = threadIdx.x; i < n_batch; i += 64) accept[i] = good[i];
}
// Fall-back, stage 1 (only after a failed batch check): one wave per GROUP of 64 proofs checks e(L_g, s_g2) == e(R_g, G2)
// for the group's own sums (the bucket MSMs of h2v_pippenger.hpp, one small problem per group and side).  A group that
// passes is final - accept[i] = good_i, flags[1 + g] = 1 - and only the groups that fail go on to the per-proof kernels:
// one rejecting proof in 4096 costs 64 per-proof MSMs and pairings, not 4096.  Soundness per group as for the batch (the
// same coefficients r_i: a failing proof survives its group's check with probability <= 2^-128).
extern "C" __global__ void __launch_bounds__(64, 2)
k_pairing_rlc_groups(H2vDevPlan plan, const uint32_t *__restrict__ pts_g, const uint8_t *__restrict__ valid_g, const uint32_t *__restrict__ er_g,
                     const uint32_t *__restrict__ el_g, uint32_t *__restrict__ status_g, uint8_t *__restrict__ accept_g, uint32_t n_groups,
                     uint32_t n_batch, const uint8_t *__restrict__ good, uint8_t *__restrict__ accept, uint32_t *__restrict__ flags,
                     uint32_t *__restrict__ stats) {
    if (flags[0]) return;
    const uint32_t g = blockIdx.x;
    bool ok = false;
    pairing_coop_body<true>(plan, n_groups, pts_g, valid_g, nullptr, er_g, el_g, status_g, accept_g, nullptr, g, &ok);
    if (threadIdx.x == 0) flags[1 + g] = ok ? 1u : 0u;
    if (threadIdx.x == 0 && stats) {
        atomicAdd(stats, 1u);
        if (!ok) atomicAdd(stats + 1, 1u);
    }
    if (ok) {
        const uint32_t i = g * 64 + threadIdx.x;
        if (i < n_batch) accept[i] = good[i];
    }
}
