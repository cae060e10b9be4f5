// Binary verifier-plan layout shared by the plan compiler (plan.py: Plan.to_bytes) and the HIP backend.
// The plan is this build's "VerifyingKey surface": the reference's InstantiationSpecificData
// (/root/reference/src/plutus_gen/extraction/data/circuit_types/instantiation_data.rs:26-41) plus the
// straight-line verifier program that `extract_circuit` + the emitters would have rendered as source text.
#pragma once
#include <stdint.h>

#define H2V_PLAN_MAGIC "H2VPLAN1"
#define H2V_PLAN_VERSION 4u
#define H2V_PLAN_HDR_WORDS 46
#define H2V_MILLER_LINES 68  // 63 doublings + 5 additions for |x| = 0xd201000000010000

// header words (uint32 little-endian) after the 8-byte magic
enum {
    H2V_HW_VERSION = 0, H2V_HW_PROOF_LEN, H2V_HW_N_PI, H2V_HW_N_CI, H2V_HW_N_REGS, H2V_HW_N_INSTR, H2V_HW_N_CONSTS,
    H2V_HW_N_POINTS, H2V_HW_N_VK_BASES, H2V_HW_N_TERMS, H2V_HW_N_TRACE, H2V_HW_PI_POINT, H2V_HW_N_SQUEEZES,
    H2V_HW_STREAM_LEN,
    H2V_HW_OFF_INSTR, H2V_HW_OFF_CONSTS, H2V_HW_OFF_POINTS, H2V_HW_OFF_VK_BASES, H2V_HW_OFF_TERMS, H2V_HW_OFF_LINES_SG2,
    H2V_HW_OFF_LINES_G2, H2V_HW_OFF_TRACE, H2V_HW_TOTAL_LEN, H2V_HW_OFF_LINES28_SG2, H2V_HW_OFF_LINES28_G2,
    // recursion (IVC; plan.py / ivc.py): flag, number of terms of the proof's own MSM, and the public-input positions
    // (x_hi, x_lo, y_hi, y_lo) of the two accumulator points
    H2V_HW_IVC, H2V_HW_N_MAIN_TERMS, H2V_HW_ACC_IDX0 /* .. +7 */,
    // lanes per proof of the transcript + combiner program (the instruction stream is a sequence of bundles of that many
    // records), and an optional second, wider schedule of the same program: lanes (0 = none), registers, records, offset
    H2V_HW_VM_LANES = H2V_HW_ACC_IDX0 + 8, H2V_HW_VM2_LANES, H2V_HW_VM2_N_REGS, H2V_HW_VM2_N_INSTR, H2V_HW_VM2_OFF_INSTR
};

// opcodes of the transcript + Fr-combiner program (8-byte records: op, pad, dst, a, b).  The program is a sequence of
// BUNDLES of `vm_lanes` records: record l of a bundle is executed by lane l of the proof's lanes (NOP = idle).  Records of
// one bundle are independent (none reads or writes a register another one writes); the transcript operations (ABSORB_*,
// READ_*, SQUEEZE) and END sit on lane 0 with the rest of their bundle idle.
enum {
    H2V_OP_END = 0, H2V_OP_ABSORB_REG, H2V_OP_ABSORB_CI, H2V_OP_LOAD_INSTANCE, H2V_OP_READ_POINT, H2V_OP_READ_SCALAR,
    H2V_OP_SQUEEZE, H2V_OP_CONST, H2V_OP_ADD, H2V_OP_SUB, H2V_OP_MUL, H2V_OP_NEG, H2V_OP_INV, H2V_OP_OUT_SCALAR,
    H2V_OP_ASSERT_ZERO,  // status |= H2V_ST_RECURSION unless reg a == 0 (verifying-key hash check of the IVC fold)
    H2V_OP_NOP,
    H2V_OP_COUNT
};
enum { H2V_TERM_PROOF_POINT = 0, H2V_TERM_VK_BASE = 1, H2V_TERM_COMMITTED_INSTANCE = 2, H2V_TERM_ACC_POINT = 3 };

// per-proof status bits produced on the device: H2V_ST_* of include/h2v.h (0 = nothing wrong so far)
#ifndef H2V_ST_BAD_SCALAR
#define H2V_ST_BAD_SCALAR 1u
#define H2V_ST_INVERSE_OF_ZERO 2u
#define H2V_ST_SHORT_PROOF 4u
#define H2V_ST_BAD_POINT 8u
#define H2V_ST_PAIRING 16u
#endif
#ifndef H2V_ST_RECURSION
#define H2V_ST_RECURSION 32u
#endif

struct H2vInstr {
    uint8_t op, pad;
    uint16_t dst, a, b;
};

// device view of a loaded plan (all pointers are device pointers)
struct H2vDevPlan {
    uint32_t proof_len, n_pi, n_ci, n_regs, n_instr, n_consts, n_points, n_vk_bases, n_terms, n_trace, pi_point;
    uint32_t vm_lanes;         // records per bundle of `instr` (the launcher swaps in the wide schedule's instr / n_instr / n_regs / vm_lanes)
    const H2vInstr *instr;
    const uint32_t *consts;    // n_consts * 8   (Fr, Montgomery)
    const uint32_t *points;    // n_points       (byte offset in the proof)
    const uint32_t *vk_bases;  // n_vk_bases * 24 (affine x||y, Montgomery; all-zero = infinity)
    const uint32_t *terms;     // n_terms * 2    (kind, index)
    const uint32_t *lines_sg2; // 68 * 48        (lambda.c0, lambda.c1, c.c0, c.c1)
    const uint32_t *lines_g2;
    const uint32_t *lines28_sg2; // 68 * 8 operand slots of 16 dwords (28-bit limbs): cooperative pairing engine
    const uint32_t *lines28_g2;
    const uint32_t *trace;     // n_trace * 2    (slot id, register)
    // recursion (IVC): terms [0, n_main_terms) are the proof's own MSM, term n_main_terms is acc_left, the rest
    // acc_right and its fixed bases; acc_idx = public-input positions of (x_hi, x_lo, y_hi, y_lo) x (left, right)
    uint32_t ivc, n_main_terms;
    uint32_t acc_idx[8];
    const uint32_t *fold_terms;  // 4 x (kind, index): el + c*acc_left, er + c*acc_right over the fold's own point buffer
    const uint32_t *vk_tab;      // n_vk_bases x 2 x 224: affine window tables [1..8]B and [1..8]phi(B), built at plan load
    // fixed-base MSM launches (non-recursive plans whose VK-base terms are the tail of the term list): all-window tables of
    // the VK bases [base][65][8][28]; terms [0, n_var) are per-proof points, [n_var, n_var + n_fix) VK bases
    const uint32_t *fix_tab;
    uint32_t n_var, n_fix;
    // window width of the all-window tables (4, 8 or 12 bits: signed digits, 2^(c-1) entries per window) and their number
    // of windows W; layout fix_tab[base][W][2^(c-1)][28] (affine x, y: 2 x 14 limbs of 28 bits)
    uint32_t fix_c, fix_W;
    // optional wide schedule of the program (0 lanes = none); host-side use only (launch_vm swaps it in)
    uint32_t wide_lanes, wide_n_regs, wide_n_instr;
    const H2vInstr *wide_instr;
};
// per-proof point slots: the proof's G1 elements, the committed instance, then (recursion) the two accumulator points
#define H2V_SLOTS(plan) ((plan).n_points + (plan).n_ci + 2u * (plan).ivc)
