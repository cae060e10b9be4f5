"""Verifier-plan compiler: VerifyingKey -> flat binary "plan" consumed by the HIP backend.

This is the MI355X-side counterpart of the reference's plan-time layer `extract_circuit`
(/root/reference/src/plutus_gen/extraction/mod.rs:31-232): where the reference walks the VK once and emits
Plinth/Aiken *source text* for the verifier, this compiler walks the same data once and emits

  * a straight-line, branch-free bytecode program over an Fr register file (transcript replay P1-P2 +
    Fr combiner P3-P6 of SURVEY.md §3.2) that every GPU lane executes in lock-step for its own proof,
  * the static proof-point table (byte offset of every G1 in the proof) for the decompression kernel,
  * the flattened final-MSM term table  er = sum_t s_t * B_t  (SURVEY.md App. A.3),
  * device-ready constants: Fr constant pool, VK bases (affine, Montgomery limbs), and the precomputed
    optimal-ate line tables of the two FIXED G2 pairing arguments s_g2 and G2.

Reference definitions followed (same order, same formulas):
  proof layout        src/plutus_gen/extraction/data/extraction_steps/proof.rs:13-143, pcs/kzg.rs:55-79
  inputs/absorbs      src/plutus_gen/emitters/aiken.rs:44-84
  instance eval       emitters/aiken.rs:200-223
  lagrange basis      aiken-verifier/aiken_halo2/lib/lagrange.ak:79-96
  gates/lookups       emitters/aiken.rs:248-330, extraction/data/languages/aiken.rs:18-29,122-182
  permutation         extraction/data/extraction_steps/permutation.rs:80-303, emitters/aiken.rs:332-441
  trashcans           emitters/aiken.rs:444-461
  hEval / vanishing   emitters/aiken.rs:463-575, aiken-verifier/templates/verification_h2.hbs:85-96,
                      extraction/data/extraction_steps/vanishing.rs:6-52
  queries / sets      extraction/mod.rs:120-226, extraction/pcs/mod.rs:36-109, emitters/aiken.rs:580-587
  multi-open          aiken-verifier/aiken_halo2/lib/halo2_kzg.ak:15-171, lagrange.ak:40-77

All field inversions of one proof depend only on the challenges x and x3, so they are gathered into ONE
Montgomery-trick batch with a single INV instruction.  The reference rejects (recip_eea divides by zero,
bls_utils.ak:151-154) iff any inverted value is zero; the product of the merged batch is zero under exactly
the same condition, so accept/reject is unchanged.  (Only deviation: the interpolation denominators are taken
as x^(m-1)*prod(w^a - w^b), which differs from lagrange.ak's value-comparison skip only when x == 0.)
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

from . import bls12_381 as bls
from .vk import VerifyingKey

R = bls.R

# ---- opcodes (mirrored in csrc/h2v_plan.h)
OP_END, OP_ABSORB_REG, OP_ABSORB_CI, OP_LOAD_INSTANCE, OP_READ_POINT, OP_READ_SCALAR, OP_SQUEEZE, OP_CONST, \
    OP_ADD, OP_SUB, OP_MUL, OP_NEG, OP_INV, OP_OUT_SCALAR, OP_ASSERT_ZERO, OP_NOP = range(16)
OP_NAMES = ["END", "ABSORB_REG", "ABSORB_CI", "LOAD_INSTANCE", "READ_POINT", "READ_SCALAR", "SQUEEZE", "CONST",
            "ADD", "SUB", "MUL", "NEG", "INV", "OUT_SCALAR", "ASSERT_ZERO", "NOP"]
# the transcript is one hash chain: these run on lane 0 of a proof's lanes, alone in their bundle, in program order
TRANSCRIPT_OPS = frozenset((OP_ABSORB_REG, OP_ABSORB_CI, OP_READ_POINT, OP_READ_SCALAR, OP_SQUEEZE))
VM_LANE_CHOICES = (1, 2, 4, 8, 16)
VM_LDS_BYTES = 160 * 1024 - 8192 - 1024   # register file budget of one block (h2v_capi.hip: vm_lds_fits)

# ---- MSM term kinds (ACC_POINT: the two accumulator points rebuilt from the public inputs, recursion only)
TERM_PROOF_POINT, TERM_VK_BASE, TERM_COMMITTED_INSTANCE, TERM_ACC_POINT = 0, 1, 2, 3

# ---- trace slots (same order as the oracle's trace struct; expressions start at TRACE_EXPR0)
TRACE_NAMES = ["theta", "beta", "gamma", "trash", "y", "x", "x1", "x2", "x3", "x4", "x_prev", "x_next", "x_last", "xn",
               "l_last", "l_0", "active_rows", "h_eval", "vanishing_s", "f_eval", "v",
               # only in plans compiled with legacy_no_trash_squeeze (the golden-proof replay of the GPU tests)
               "advice_eval_1", "advice_eval_2", "advice_eval_3"]
TRACE_EXPR0 = 32
MAX_TRACE_EXPR = 256

MILLER_LINES = 68
PLAN_MAGIC = b"H2VPLAN1"
PLAN_VERSION = 4
PLAN_HDR_WORDS = 46  # 8 + 4*46 = 192 bytes: keeps every 16-byte-padded section 16-byte aligned

ROT_LAST = "last"  # rotation key of x_last = w^-(bf+1) x


@dataclass
class Plan:
    vk_name: str
    proof_len: int
    n_pi: int
    n_ci: int
    n_regs: int
    instrs: List[Tuple[int, int, int, int]]  # (op, dst, a, b)
    consts: List[int]
    points: List[int]  # proof byte offset of each G1 read from the proof, in read order
    point_names: List[str]
    vk_bases: List[Optional[Tuple[int, int]]]  # affine points
    terms: List[Tuple[int, int]]  # (kind, index)
    term_names: List[str]
    pi_point: int
    lines_sg2: list
    lines_g2: list
    trace: List[Tuple[int, int]]  # (slot id, register)
    n_squeezes: int = 0
    stream_len: int = 0  # bytes hashed by the transcript
    # recursion (ivc.py): terms[:n_main_terms] is the proof's own MSM; then one term for acc_left, then acc_right
    # followed by its fixed bases.  acc_coords: public-input positions (x_hi, x_lo, y_hi, y_lo) x (left, right).
    n_main_terms: int = -1
    acc_coords: Optional[List[int]] = None
    # lanes per proof of the transcript + combiner kernel: `instrs` is a sequence of bundles of vm_lanes records, record
    # l of a bundle is executed by lane l (OP_NOP = idle); no record of a bundle reads or writes a register that another
    # record of the same bundle writes, so running the records one after the other (run_plan) gives the same result
    vm_lanes: int = 1
    # optional second schedule of the same program for launches that leave most of the chip idle (small batches,
    # large circuits): more lanes per proof, shorter chain, more waves.  (lanes, n_regs, instrs); the trace table
    # refers to the first schedule only.
    wide: Optional[Tuple[int, int, List[Tuple[int, int, int, int]]]] = None
    # host-side record of the commitment map the compiler derived (pcs/mod.rs:36-109), not part of the blob: per
    # commitment, in order, {commitment: (kind, idx), set: first-seen point-set index, sorted_set: position after the
    # cardinality sort (aiken.rs:580-587), pairs: [(rotation, eval key)...]} - what tests compare with ProofData.hs
    commitment_map: Optional[list] = None

    def __post_init__(self):
        if self.n_main_terms < 0:
            self.n_main_terms = len(self.terms)

    @property
    def is_recursive(self) -> bool:
        return self.acc_coords is not None

    @property
    def n_terms(self) -> int:
        return len(self.terms)

    def to_bytes(self) -> bytes:
        def pad8(b: bytearray):  # (16-byte alignment: the engine reads operand slots with 128-bit loads)
            while len(b) % 16:
                b.append(0)

        body = bytearray()
        offs = {}

        def section(name, data: bytes):
            pad8(body)
            offs[name] = len(body)
            body.extend(data)

        section("instr", b"".join(struct.pack("<BBHHH", op, 0, d, a, b) for op, d, a, b in self.instrs))
        if self.wide:
            section("instr_wide", b"".join(struct.pack("<BBHHH", op, 0, d, a, b) for op, d, a, b in self.wide[2]))
        section("consts", b"".join(bls.fr_mont_bytes(c) for c in self.consts))
        section("points", b"".join(struct.pack("<I", o) for o in self.points))
        vb = bytearray()
        for p in self.vk_bases:
            vb += bytes(96) if p is None else bls.fp_mont_bytes(p[0]) + bls.fp_mont_bytes(p[1])
        section("vk_bases", bytes(vb))
        section("terms", b"".join(struct.pack("<II", k, i) for k, i in self.terms))
        for name, tab in (("lines_sg2", self.lines_sg2), ("lines_g2", self.lines_g2)):
            assert len(tab) == MILLER_LINES
            section(name, b"".join(bls.f2_mont_bytes(lam) + bls.f2_mont_bytes(c) for lam, c in tab))
        section("trace", b"".join(struct.pack("<II", s, r) for s, r in self.trace))
        # the same line tables as operand slots of the cooperative pairing engine (8 x 64 B per line)
        for name, tab in (("lines28_sg2", self.lines_sg2), ("lines28_g2", self.lines_g2)):
            section(name, b"".join(bls.line_slots(lam, c) for lam, c in tab))
        pad8(body)
        hdr_len = 8 + 4 * PLAN_HDR_WORDS
        fields = [PLAN_VERSION, self.proof_len, self.n_pi, self.n_ci, self.n_regs, len(self.instrs), len(self.consts),
                  len(self.points), len(self.vk_bases), len(self.terms), len(self.trace), self.pi_point,
                  self.n_squeezes, self.stream_len]
        sect = [hdr_len + offs[k] for k in ("instr", "consts", "points", "vk_bases", "terms", "lines_sg2", "lines_g2",
                                            "trace")]
        fields += sect + [hdr_len + len(body)]
        fields += [hdr_len + offs["lines28_sg2"], hdr_len + offs["lines28_g2"]]
        fields += [1 if self.is_recursive else 0, self.n_main_terms] + list(self.acc_coords or [0] * 8)
        fields += [self.vm_lanes]
        fields += [self.wide[0], self.wide[1], len(self.wide[2]), hdr_len + offs["instr_wide"]] if self.wide else [0, 0, 0, 0]
        fields += [0] * (PLAN_HDR_WORDS - len(fields))
        return PLAN_MAGIC + struct.pack("<%dI" % PLAN_HDR_WORDS, *fields) + bytes(body)


class _Builder:
    """Straight-line program builder over virtual registers with hash-consing of pure operations."""

    def __init__(self):
        self.code: List[List[int]] = []  # [op, dst, a, b] on virtual registers
        self.n_virt = 0
        self.consts: List[int] = []
        self.const_idx: Dict[int, int] = {}
        self.cse: Dict[tuple, int] = {}
        self.const_reg: Dict[int, int] = {}
        self.value_of_const_reg: Dict[int, int] = {}
        # scheduling hint: virtual register -> a register whose bundle must come first although the instruction does not
        # read it (a re-load that would otherwise be hoisted to the top of the program and stay live all the way down)
        self.not_before: Dict[int, int] = {}

    def new(self) -> int:
        self.n_virt += 1
        return self.n_virt - 1

    def emit(self, op, dst=0, a=0, b=0):
        self.code.append([op, dst, a, b])

    def const(self, v: int) -> int:
        v %= R
        if v in self.const_reg:
            return self.const_reg[v]
        if v not in self.const_idx:
            self.const_idx[v] = len(self.consts)
            self.consts.append(v)
        r = self.new()
        self.emit(OP_CONST, r, self.const_idx[v], 0)
        self.const_reg[v] = r
        self.value_of_const_reg[r] = v
        return r

    def _pure(self, op, a, b=0, commutative=False):
        key = (op, min(a, b), max(a, b)) if commutative else (op, a, b)
        if key in self.cse:
            return self.cse[key]
        r = self.new()
        self.emit(op, r, a, b)
        self.cse[key] = r
        return r

    def add(self, a, b):
        return self._pure(OP_ADD, a, b, True)

    def sub(self, a, b):
        return self._pure(OP_SUB, a, b)

    def mul(self, a, b):
        return self._pure(OP_MUL, a, b, True)

    def neg(self, a):
        return self._pure(OP_NEG, a)

    def inv(self, a):
        return self._pure(OP_INV, a)

    def batch_inverse(self, regs: List[int]) -> List[int]:
        """Montgomery trick: len-1 + 1 INV + 2(len-1) multiplications; zero anywhere => INV of zero => reject."""
        if not regs:
            return []
        pre = [regs[0]]
        for r in regs[1:]:
            pre.append(self.mul(pre[-1], r))
        inv = self.inv(pre[-1])
        out = [0] * len(regs)
        for i in range(len(regs) - 1, 0, -1):
            out[i] = self.mul(inv, pre[i - 1])
            inv = self.mul(inv, regs[i])
        out[0] = inv
        return out


def _rot_value(vk: VerifyingKey, rot) -> int:
    """omega^rotation as a constant."""
    n = -(vk.blinding_factors + 1) if rot == ROT_LAST else rot
    return pow(vk.omega, n, R) if n >= 0 else pow(vk.omega_inv, -n, R)


def _rot_sort_key(rot):
    # RotationDescription derive(Ord): Last < Previous < Current < Next < Custom(n)   (rotation_description.rs:14-24)
    if rot == ROT_LAST:
        return (0, 0)
    if rot == -1:
        return (1, 0)
    if rot == 0:
        return (2, 0)
    if rot == 1:
        return (3, 0)
    return (4, rot)


def compile_plan(vk: VerifyingKey, lanes: Optional[int] = None, legacy_no_trash_squeeze: bool = False) -> Plan:
    """legacy_no_trash_squeeze (TEST ONLY): the proof layout before the `trash` challenge was added
    (extraction_steps/proof.rs:68) - the layout of the reference's only in-tree full proof, transcript.ak:241-382 -
    so that the GPU transcript kernel can be replayed against every challenge of that vector.  Only for keys without
    trashcan arguments; the first three advice evaluations are added to the trace set."""
    b = _Builder()
    L = len(vk.lookups)
    Cn = vk.n_perm_chunks
    n_trash = len(vk.trashcans)
    n_splits = vk.quotient_poly_degree
    n_ci = vk.n_committed_instances
    assert n_ci in (0, 1)
    assert len(vk.permutation_commitments) == len(vk.permutation_columns) > 0
    assert vk.cs_degree >= 3

    pos = 0
    points: List[int] = []
    point_names: List[str] = []
    n_squeezes = 0
    stream_len = 0

    def read_point(name) -> int:
        nonlocal pos, stream_len
        idx = len(points)
        points.append(pos)
        point_names.append(name)
        b.emit(OP_READ_POINT, 0, pos & 0xFFFF, pos >> 16)
        pos += 48
        stream_len += 49
        return idx

    def read_scalar() -> int:
        nonlocal pos, stream_len
        r = b.new()
        b.emit(OP_READ_SCALAR, r, pos & 0xFFFF, pos >> 16)
        pos += 32
        stream_len += 33
        return r

    def squeeze() -> int:
        nonlocal n_squeezes, stream_len
        r = b.new()
        b.emit(OP_SQUEEZE, r)
        n_squeezes += 1
        stream_len += 1
        return r

    def absorb(reg):
        nonlocal stream_len
        b.emit(OP_ABSORB_REG, 0, reg, 0)
        stream_len += 33

    one = b.const(1)
    zero = b.const(0)

    # ---- P1: transcript init + inputs (verification_h2.hbs:26-28, aiken.rs:44-84)
    absorb(b.const(vk.transcript_repr))
    if n_ci:
        b.emit(OP_ABSORB_CI)
        stream_len += 49
    absorb(b.const(vk.n_public_inputs))
    pis = []
    for k in range(vk.n_public_inputs):
        r = b.new()
        b.emit(OP_LOAD_INSTANCE, r, k, 0)
        absorb(r)
        pis.append(r)

    # ---- P2: proof read order (proof.rs:13-143)
    # advice commitments phase by phase, every phase followed by the squeezes of its challenges (proof.rs:22-46); the
    # challenges themselves are never used (no expression may name one), they only advance the transcript
    adv_phase = vk.advice_column_phase or [0] * vk.num_advice_columns
    chal_phase = vk.challenge_phase or []
    adv_pts: List[Optional[int]] = [None] * vk.num_advice_columns
    for phase in range((max(adv_phase) if adv_phase else 0) + 1):
        for i in range(vk.num_advice_columns):
            if adv_phase[i] == phase:
                adv_pts[i] = read_point("a%d" % (i + 1))
        for ph in chal_phase:
            if ph == phase:
                squeeze()
    theta = squeeze()
    lk_pin, lk_ptab = [], []
    for i in range(L):
        lk_pin.append(read_point("permuted_input_%d" % (i + 1)))
        lk_ptab.append(read_point("permuted_table_%d" % (i + 1)))
    beta = squeeze()
    gamma = squeeze()
    perm_pts = [read_point("permutations_committed_%s" % chr(ord("a") + i)) for i in range(Cn)]
    lk_prod = [read_point("lookup_commitment_%d" % (i + 1)) for i in range(L)]
    if legacy_no_trash_squeeze:
        assert n_trash == 0, "the legacy layout has no trashcan arguments"
        trash = b.const(0)
    else:
        trash = squeeze()
    trash_pts = [read_point("trashcan_commitment_%d" % (i + 1)) for i in range(n_trash)]
    vanish_rand_pt = read_point("vanishing_rand")
    y = squeeze()
    split_pts = [read_point("vanishing_split_%d" % (i + 1)) for i in range(n_splits)]
    x = squeeze()
    # xn_minus_one = x^(n-1), xn = xn_minus_one * x  (aiken.rs:124-136); n = 2^k
    acc = one
    for _ in range(vk.k):
        acc = b.mul(b.mul(acc, acc), x)
    xn_minus_one = acc
    xn = b.mul(xn_minus_one, x)

    # instance evaluations: committed columns are read now, public ones are computed after the batch inversion
    instance_eval: List[Optional[int]] = []
    for (col, _rot) in vk.instance_queries:
        if col < n_ci:
            instance_eval.append(read_scalar())
        else:
            instance_eval.append(None)
    advice_eval = [read_scalar() for _ in vk.advice_queries]
    fixed_eval = [read_scalar() for _ in vk.fixed_queries]
    random_eval = read_scalar()
    perm_common = [read_scalar() for _ in vk.permutation_commitments]
    perm_eval = []
    for i in range(Cn):
        z = [read_scalar(), read_scalar()]
        z.append(read_scalar() if i != Cn - 1 else None)
        perm_eval.append(z)
    lk_eval = []
    for i in range(L):
        # product, product_next, permuted_input, permuted_input_inv, permuted_table  (aiken.rs:180-192)
        lk_eval.append([read_scalar() for _ in range(5)])
    trash_eval = [read_scalar() for _ in range(n_trash)]

    # ---- queries -> commitment map -> point sets (mod.rs:120-226, pcs/mod.rs:36-109)
    queries = []  # (commitment key, eval register or callable, rotation)
    for qi, (col, rot) in enumerate(vk.advice_queries):
        queries.append((("advice", col), ("advice", qi), rot))
    for qi, (col, rot) in enumerate(vk.instance_queries):
        if col < n_ci:
            queries.append((("instance", col), ("instance", qi), rot))
    for i in range(Cn):
        queries.append((("perm", i), ("perm", i, 0), 0))
        queries.append((("perm", i), ("perm", i, 1), 1))
    for i in range(Cn - 2, -1, -1):
        queries.append((("perm", i), ("perm", i, 2), ROT_LAST))
    for i in range(L):
        queries.append((("lookup", i), ("lk", i, 0), 0))
        queries.append((("perm_input", i), ("lk", i, 2), 0))
        queries.append((("perm_table", i), ("lk", i, 4), 0))
        queries.append((("perm_input", i), ("lk", i, 3), -1))
        queries.append((("lookup", i), ("lk", i, 1), 1))
    for i in range(n_trash):
        queries.append((("trash", i), ("trash", i), 0))
    for qi, (col, rot) in enumerate(vk.fixed_queries):
        queries.append((("fixed", col), ("fixed", qi), rot))
    for i in range(len(vk.permutation_commitments)):
        queries.append((("common", i), ("common", i), 0))
    queries.append((("vanishing_g", 0), ("vanishing_s",), 0))
    queries.append((("vanishing_rand", 0), ("random",), 0))

    commitments: List[tuple] = []
    cmap: Dict[tuple, list] = {}
    for ck, ek, rot in queries:
        if ck not in cmap:
            cmap[ck] = []
            commitments.append(ck)
        cmap[ck].append((rot, ek))
    for ck in commitments:
        cmap[ck].sort(key=lambda pe: _rot_sort_key(pe[0]))  # stable, by point
    uniq_sets: List[tuple] = []
    set_of: Dict[tuple, int] = {}
    for ck in commitments:
        pts = tuple(p for p, _ in cmap[ck])
        if pts not in uniq_sets:
            uniq_sets.append(pts)
        set_of[ck] = uniq_sets.index(pts)
    sort_order = sorted(range(len(uniq_sets)), key=lambda i: (len(uniq_sets[i]), i))  # aiken.rs:580-587
    S = len(uniq_sets)

    # ---- PCS tail (pcs/kzg.rs:55-79)
    x1 = squeeze()
    x2 = squeeze()
    f_pt = read_point("f_commitment")
    x3 = squeeze()
    q_evals = [read_scalar() for _ in range(S)]
    x4 = squeeze()
    pi_pt = read_point("pi")
    proof_len = pos

    # ---- rotated evaluation points (verification_h2.hbs:32-37)
    def rotated(rot):
        w = _rot_value(vk, rot)
        return x if w == 1 else b.mul(b.const(w), x)

    x_last = rotated(ROT_LAST)

    # ---- the single batch inversion ------------------------------------------------------------------
    inv_in: List[int] = []
    # (a) lagrange basis at rotations -(bf+1)..0 : denominators x - w^i  (lagrange.ak:79-96)
    bf = vk.blinding_factors
    van_rot_w = [_rot_value(vk, i) for i in range(-(bf + 1), 1)]
    van_idx = []
    for w in van_rot_w:
        van_idx.append(len(inv_in))
        inv_in.append(b.sub(x, b.const(w)))
    # (b) public-input lagrange basis at rotations 0..n_pi (aiken.rs:207-222)
    need_pi_basis = any(e is None for e in instance_eval) and vk.n_public_inputs > 0
    pi_rot_w = [pow(vk.omega, i, R) for i in range(vk.n_public_inputs + 1)] if need_pi_basis else []
    pi_idx = []
    for w in pi_rot_w:
        pi_idx.append(len(inv_in))
        inv_in.append(b.sub(x, b.const(w)))
    # (c) xn - 1 (verification_h2.hbs:90)
    xn_m1 = b.sub(xn, one)
    xn_idx = len(inv_in)
    inv_in.append(xn_m1)
    # (d) interpolation denominators per sorted set: prod_{j != i} (p_i - p_j) = x^(m-1) * prod (w_i - w_j)
    xpow = {0: one, 1: x}

    def x_power(e):
        if e not in xpow:
            xpow[e] = b.mul(x_power(e - 1), x)
        return xpow[e]

    interp_idx = []
    for s in range(S):
        pts = uniq_sets[sort_order[s]]
        ws = [_rot_value(vk, p) for p in pts]
        idxs = []
        for i in range(len(pts)):
            c = 1
            for j in range(len(pts)):
                if j != i:
                    c = c * (ws[i] - ws[j]) % R
            idxs.append(len(inv_in))
            inv_in.append(b.mul(b.const(c), x_power(len(pts) - 1)) if len(pts) > 1 else one)
        interp_idx.append(idxs)
    # (e) f_eval denominators prod_j (x3 - p_j)  (halo2_kzg.ak:134-141)
    set_points = []
    fden_idx = []
    for s in range(S):
        pts = [rotated(p) for p in uniq_sets[sort_order[s]]]
        set_points.append(pts)
        d = one
        for p in pts:
            d = b.mul(d, b.sub(x3, p))
        fden_idx.append(len(inv_in))
        inv_in.append(d)
    inv_out = b.batch_inverse(inv_in)

    # ---- P3: lagrange basis, active rows (verification_h2.hbs:39-60)
    common = b.mul(xn_m1, b.const(vk.barycentric_weight))
    basis = [b.mul(b.mul(inv_out[van_idx[i]], common), b.const(van_rot_w[i])) for i in range(bf + 2)]
    l_last = basis[0]
    l_blind = basis[1:1 + bf]
    l_0 = basis[-1]
    sum_blind = zero
    for e in l_blind:
        sum_blind = b.add(e, sum_blind)
    active_rows = b.sub(one, b.add(l_last, sum_blind))
    # public-input column evaluation: inner_product(basis[0..n_pi), inputs)
    if need_pi_basis:
        pbasis = [b.mul(b.mul(inv_out[pi_idx[i]], common), b.const(pi_rot_w[i])) for i in range(vk.n_public_inputs)]
        acc = zero
        for i in range(vk.n_public_inputs):
            # the public input is loaded a second time, next to its use: the copy that was absorbed at the top of the
            # program would otherwise occupy a register through the whole proof read and the batch inversion (32 of them
            # for the sha256 wrapper: the difference between 16 and 8 proofs per block of the combiner kernel)
            r = b.new()
            b.emit(OP_LOAD_INSTANCE, r, i, 0)
            b.not_before[r] = pbasis[i]
            acc = b.add(b.mul(pbasis[i], r), acc)
        pub_eval = acc
    else:
        pub_eval = zero
    instance_eval = [e if e is not None else pub_eval for e in instance_eval]

    # ---- P4: combiner
    def ev(e) -> int:
        t = e[0]
        if t == "const":
            return b.const(e[1])
        if t == "fixed":
            return fixed_eval[e[1]]
        if t == "advice":
            return advice_eval[e[1]]
        if t == "neg":
            return b.neg(ev(e[1]))
        if t == "sum":
            return b.add(ev(e[1]), ev(e[2]))
        if t == "prod":
            return b.mul(ev(e[1]), ev(e[2]))
        if t == "scaled":
            return b.mul(ev(e[1]), b.const(e[2]))
        raise ValueError(t)

    def compress(exprs, ch):
        acc = zero
        for e in exprs:
            acc = b.add(b.mul(acc, ch), ev(e))
        return acc

    expressions: List[int] = [ev(g) for g in vk.gates]
    # permutation terms (permutation.rs:80-143)
    expressions.append(b.mul(l_0, b.sub(one, perm_eval[0][0])))
    zl = perm_eval[Cn - 1][0]
    expressions.append(b.mul(l_last, b.sub(b.mul(zl, zl), zl)))
    for i in range(1, Cn):
        expressions.append(b.mul(b.sub(perm_eval[i][0], perm_eval[i - 1][2]), l_0))
    # permutation sets (permutation.rs:145-303; aiken.rs:345-441)
    bx = b.mul(beta, x)

    def column_eval(ty, col):
        if ty == "advice":
            return advice_eval[vk.advice_queries.index((col, 0))]
        if ty == "fixed":
            return fixed_eval[vk.fixed_queries.index((col, 0))]
        return instance_eval[vk.instance_queries.index((col, 0))]

    not_blind = b.sub(one, b.add(l_last, sum_blind))
    for i in range(Cn):
        left = perm_eval[i][1]
        right = perm_eval[i][0]
        for idx in range(vk.chunk_len):
            col = i * vk.chunk_len + idx
            if col >= len(vk.permutation_columns):
                break
            e = column_eval(*vk.permutation_columns[col])
            left = b.mul(left, b.add(b.add(e, b.mul(beta, perm_common[col])), gamma))
            right = b.mul(right, b.add(b.add(e, b.mul(bx, b.const(pow(bls.DELTA, col, R)))), gamma))
        expressions.append(b.mul(b.sub(left, right), not_blind))
    # lookups (aiken.rs:264-330)
    for i, (ins, tabs) in enumerate(vk.lookups):
        tab = compress(tabs, theta)
        inp = compress(ins, theta)
        prod, prod_next, pin, pinv, ptab = lk_eval[i]
        expressions.append(b.mul(l_0, b.sub(one, prod)))
        expressions.append(b.mul(l_last, b.sub(b.mul(prod, prod), prod)))
        left = b.mul(b.mul(prod_next, b.add(pin, beta)), b.add(ptab, gamma))
        right = b.mul(b.mul(prod, b.add(inp, beta)), b.add(tab, gamma))
        expressions.append(b.mul(b.sub(left, right), active_rows))
        d = b.sub(pin, ptab)
        expressions.append(b.mul(l_0, d))
        expressions.append(b.mul(b.mul(d, b.sub(pin, pinv)), active_rows))
    # trashcans (aiken.rs:444-461)
    for i, (sel, cons) in enumerate(vk.trashcans):
        expressions.append(b.sub(compress(cons, trash), b.mul(b.sub(one, ev(sel)), trash_eval[i])))
    # hEval = Horner in y with acc0 = 0 (aiken.rs:557-563); vanishing_s = hEval / (xn - 1)
    h_eval = zero
    for e in expressions:
        h_eval = b.add(b.mul(h_eval, y), e)
    vanishing_s = b.mul(h_eval, inv_out[xn_idx])

    # ---- P6: multi-open scalars
    def eval_reg(ek):
        k = ek[0]
        if k == "advice":
            return advice_eval[ek[1]]
        if k == "instance":
            return instance_eval[ek[1]]
        if k == "fixed":
            return fixed_eval[ek[1]]
        if k == "perm":
            return perm_eval[ek[1]][ek[2]]
        if k == "lk":
            return lk_eval[ek[1]][ek[2]]
        if k == "trash":
            return trash_eval[ek[1]]
        if k == "common":
            return perm_common[ek[1]]
        if k == "vanishing_s":
            return vanishing_s
        if k == "random":
            return random_eval
        raise ValueError(ek)

    vk_bases: List[Optional[Tuple[int, int]]] = []
    vk_base_idx: Dict[tuple, int] = {}

    def vk_base(key, pt):
        if key not in vk_base_idx:
            vk_base_idx[key] = len(vk_bases)
            vk_bases.append(pt)
        return vk_base_idx[key]

    fixed_pts = [bls.g1_decompress(bytes.fromhex(h)) for h in vk.fixed_commitments]
    perm_cpts = [bls.g1_decompress(bytes.fromhex(h)) for h in vk.permutation_commitments]

    terms: List[Tuple[int, int]] = []
    term_names: List[str] = []
    term_scalar: List[int] = []

    def add_term(kind, index, scalar_reg, name):
        terms.append((kind, index))
        term_scalar.append(scalar_reg)
        term_names.append(name)

    q_eval_sets: List[List[int]] = []
    x4p = one
    for s in range(S):
        old = sort_order[s]
        m = len(uniq_sets[old])
        acc_evals: List[Optional[int]] = [None] * m
        x1p = one
        for ck in commitments:
            if set_of[ck] != old:
                continue
            coeff = b.mul(x4p, x1p)  # x4^s * x1^j
            kind = ck[0]
            if kind == "advice":
                add_term(TERM_PROOF_POINT, adv_pts[ck[1]], coeff, point_names[adv_pts[ck[1]]])
            elif kind == "instance":
                add_term(TERM_COMMITTED_INSTANCE, 0, coeff, "ci_1")
            elif kind == "perm":
                add_term(TERM_PROOF_POINT, perm_pts[ck[1]], coeff, point_names[perm_pts[ck[1]]])
            elif kind == "lookup":
                add_term(TERM_PROOF_POINT, lk_prod[ck[1]], coeff, point_names[lk_prod[ck[1]]])
            elif kind == "perm_input":
                add_term(TERM_PROOF_POINT, lk_pin[ck[1]], coeff, point_names[lk_pin[ck[1]]])
            elif kind == "perm_table":
                add_term(TERM_PROOF_POINT, lk_ptab[ck[1]], coeff, point_names[lk_ptab[ck[1]]])
            elif kind == "trash":
                add_term(TERM_PROOF_POINT, trash_pts[ck[1]], coeff, point_names[trash_pts[ck[1]]])
            elif kind == "fixed":
                add_term(TERM_VK_BASE, vk_base(ck, fixed_pts[ck[1]]), coeff, "f%d_commitment" % (ck[1] + 1))
            elif kind == "common":
                add_term(TERM_VK_BASE, vk_base(ck, perm_cpts[ck[1]]), coeff, "p%d_commitment" % (ck[1] + 1))
            elif kind == "vanishing_rand":
                add_term(TERM_PROOF_POINT, vanish_rand_pt, coeff, "vanishing_rand")
            elif kind == "vanishing_g":
                # vanishing_g = sum_i (x^(n-1))^i * H_{i+1}  (vanishing.rs:6-52) flattened into the MSM
                c = coeff
                for i in range(n_splits):
                    add_term(TERM_PROOF_POINT, split_pts[i], c, point_names[split_pts[i]])
                    c = b.mul(c, xn_minus_one)
            else:
                raise ValueError(ck)
            for j, (_p, ek) in enumerate(cmap[ck]):
                t = b.mul(eval_reg(ek), x1p)
                acc_evals[j] = t if acc_evals[j] is None else b.add(acc_evals[j], t)
            x1p = b.mul(x1p, x1)
        q_eval_sets.append([e for e in acc_evals])
        x4p = b.mul(x4p, x4)
    x4_S = x4p
    add_term(TERM_PROOF_POINT, f_pt, x4_S, "f_commitment")

    # f_eval (halo2_kzg.ak:121-159): r_eval by interpolation (lagrange.ak:40-77)
    f_eval = zero
    r_evals = []
    for s in range(S):
        pts = set_points[s]
        m = len(pts)
        r_eval = zero
        for i in range(m):
            num = one
            for j in range(m):
                if j != i:
                    num = b.mul(num, b.sub(x3, pts[j]))
            r_eval = b.add(r_eval, b.mul(q_eval_sets[s][i], b.mul(num, inv_out[interp_idx[s][i]])))
        r_evals.append(r_eval)
    for s in range(S - 1, -1, -1):
        e = b.mul(b.sub(q_evals[s], r_evals[s]), inv_out[fden_idx[s]])
        f_eval = b.add(b.mul(f_eval, x2), e)
    # v (halo2_kzg.ak:161-171)
    v = zero
    x4p = one
    for s in range(S + 1):
        v = b.add(v, b.mul(x4p, q_evals[s] if s < S else f_eval))
        x4p = b.mul(x4p, x4)
    # er = final_com + v*(-G1) + x3*pi ; el = pi
    add_term(TERM_VK_BASE, vk_base(("neg_g1", 0), bls.g1_neg(bls.G1_GEN)), v, "neg_g1_generator")
    add_term(TERM_PROOF_POINT, pi_pt, x3, "pi")

    # per-proof terms first, VK-base terms last (a stable partition; the sum does not care): the backend can then run the
    # two kinds as two contiguous ranges - merged ladders for the first, all-window fixed-base tables for the second
    order = sorted(range(len(terms)), key=lambda t: terms[t][0] == TERM_VK_BASE)
    terms[:] = [terms[t] for t in order]
    term_scalar[:] = [term_scalar[t] for t in order]
    term_names[:] = [term_names[t] for t in order]

    # ---- recursion: accumulator terms + verifying-key hash check (ivc.py; emitters/aiken.rs:648-757)
    n_main_terms = len(terms)
    acc_coords = None
    if vk.recursion_vks is not None:
        from . import ivc
        lay = ivc.layout(vk)
        b.emit(OP_ASSERT_ZERO, 0, b.sub(pis[lay["vk_hash"]], b.const(vk.transcript_repr)), 0)
        add_term(TERM_ACC_POINT, 0, pis[lay["left_scalar"]], "acc_left")
        add_term(TERM_ACC_POINT, 1, pis[lay["right_scalar"]], "acc_right")
        keys = [("neg_g1", 0)] + [("fixed", i) for i in range(len(fixed_pts))] + [("common", i) for i in range(len(perm_cpts))]
        names = ["neg_g1_generator"] + ["f%d_commitment" % (i + 1) for i in range(len(fixed_pts))] + \
                ["p%d_commitment" % (i + 1) for i in range(len(perm_cpts))]
        for inner in vk.recursion_vks:
            for i in range(len(inner["fixed_commitments"])):
                keys.append(("inner_f", inner["name"], i)); names.append("f%d_%s" % (i + 1, inner["name"]))
            for i in range(len(inner["permutation_commitments"])):
                keys.append(("inner_p", inner["name"], i)); names.append("p%d_%s" % (i + 1, inner["name"]))
        for key, name, h, k in zip(keys, names, ivc.fixed_bases(vk), lay["fixed_scalars"]):
            add_term(TERM_VK_BASE, vk_base(key, bls.g1_decompress(bytes.fromhex(h))), pis[k], name)
        acc_coords = [lay["left_x"][0], lay["left_x"][1], lay["left_y"][0], lay["left_y"][1],
                      lay["right_x"][0], lay["right_x"][1], lay["right_y"][0], lay["right_y"][1]]

    for t, reg in enumerate(term_scalar):
        b.emit(OP_OUT_SCALAR, t, reg, 0)
    b.emit(OP_END)

    # ---- trace registers
    named = {"theta": theta, "beta": beta, "gamma": gamma, "trash": trash, "y": y, "x": x, "x1": x1, "x2": x2,
             "x3": x3, "x4": x4, "x_last": x_last, "xn": xn, "l_last": l_last, "l_0": l_0,
             "active_rows": active_rows, "h_eval": h_eval, "vanishing_s": vanishing_s, "f_eval": f_eval, "v": v}
    if legacy_no_trash_squeeze:
        for i, r in enumerate(advice_eval[:3]):
            named["advice_eval_%d" % (i + 1)] = r
    trace_virt = [(TRACE_NAMES.index(n), r) for n, r in named.items()]
    for i, e in enumerate(expressions[:MAX_TRACE_EXPR]):
        trace_virt.append((TRACE_EXPR0 + i, e))

    # ---- register allocation (linear scan over the straight-line code)
    keep_alive = {r for _, r in trace_virt}
    (instrs, n_regs, mapping, vm_lanes), wide = _schedule_and_allocate(b.code, keep_alive, lanes, b.not_before)
    trace = [(slot, mapping[r]) for slot, r in trace_virt]

    s_g2 = bls.g2_decompress(bytes.fromhex(vk.s_g2))
    plan = Plan(
        vk_name=vk.name, proof_len=proof_len, n_pi=vk.n_public_inputs, n_ci=n_ci, n_regs=n_regs, instrs=instrs,
        consts=b.consts, points=points, point_names=point_names, vk_bases=vk_bases, terms=terms,
        term_names=term_names, pi_point=pi_pt, lines_sg2=bls.g2_line_table(s_g2),
        lines_g2=bls.g2_line_table(bls.G2_GEN), trace=trace, n_squeezes=n_squeezes, stream_len=stream_len,
        n_main_terms=n_main_terms, acc_coords=acc_coords, vm_lanes=vm_lanes, wide=wide,
        commitment_map=[{"commitment": ck, "set": set_of[ck], "sorted_set": sort_order.index(set_of[ck]),
                         "pairs": list(cmap[ck])} for ck in commitments],
    )
    return plan


def _uses(op, dst, a, b):
    if op in (OP_ADD, OP_SUB, OP_MUL):
        return (a, b)
    if op in (OP_NEG, OP_INV, OP_ABSORB_REG, OP_OUT_SCALAR, OP_ASSERT_ZERO):
        return (a,)
    return ()


def _defines(op):
    return op in (OP_LOAD_INSTANCE, OP_READ_SCALAR, OP_SQUEEZE, OP_CONST, OP_ADD, OP_SUB, OP_MUL, OP_NEG, OP_INV)


# Relative cost of one bundle on the device (one Fr multiplication = 1): used to compare schedules only.
_COST_MUL, _COST_INV, _COST_CHEAP, _COST_BUNDLE = 1.0, 14.0, 0.06, 0.05
_COST_TRANSCRIPT = {OP_ABSORB_REG: 1.6, OP_ABSORB_CI: 0.8, OP_READ_POINT: 0.8, OP_READ_SCALAR: 1.7, OP_SQUEEZE: 3.5}


def _schedule(code, lanes: int, pack_mul: bool, not_before=None):
    """List scheduling of the straight-line SSA program into bundles of `lanes` records.

    Every instruction goes to the earliest bundle after its operands' bundles that still has a free lane (records of a
    bundle must be independent: divergent lanes of a wave run in no particular order); transcript operations keep their
    program order and get a bundle of their own on lane 0.  pack_mul: a multiplication prefers the earliest bundle that
    already holds one (the bundle costs one multiplication however many lanes multiply).  Returns a list of bundles,
    each a list of `lanes` records [op, dst, a, b] (virtual registers) or None."""
    slot_of: Dict[int, int] = {}
    bundles: List[List[Optional[list]]] = []
    exclusive: List[bool] = []
    has_mul: List[bool] = []
    first_open = 0   # every bundle below this index is full or exclusive
    for ins in code:
        op, dst, a, c = ins
        if op == OP_END:
            continue
        e = 0
        for r in _uses(op, dst, a, c):
            e = max(e, slot_of[r] + 1)
        if not_before and op == OP_LOAD_INSTANCE and dst in not_before:
            e = max(e, slot_of[not_before[dst]])   # (the same bundle is fine: it does not read that register)
        if op in TRANSCRIPT_OPS:
            s = len(bundles)   # later than every operand and every earlier transcript operation
            bundles.append([list(ins)] + [None] * (lanes - 1))
            exclusive.append(True)
            has_mul.append(False)
        else:
            s = -1
            start = max(e, first_open)
            if op == OP_MUL and pack_mul:
                for k in range(start, len(bundles)):
                    if has_mul[k] and not exclusive[k] and None in bundles[k]:
                        s = k
                        break
            if s < 0:
                for k in range(start, len(bundles)):
                    if not exclusive[k] and None in bundles[k]:
                        s = k
                        break
            if s < 0:
                s = len(bundles)
                bundles.append([None] * lanes)
                exclusive.append(False)
                has_mul.append(False)
            bundles[s][bundles[s].index(None)] = list(ins)
            has_mul[s] = has_mul[s] or op == OP_MUL
            while first_open < len(bundles) and (exclusive[first_open] or None not in bundles[first_open]):
                first_open += 1
        if _defines(op):
            slot_of[dst] = s
    bundles.append([[OP_END, 0, 0, 0]] + [None] * (lanes - 1))
    return bundles


def _schedule_cost(bundles) -> float:
    total = 0.0
    for bun in bundles:
        ops = {r[0] for r in bun if r is not None}
        t = _COST_BUNDLE
        for op in sorted(ops):   # (a fixed order: the sum is floating point, and csrc/h2v_plancc.hpp must reproduce it bit for bit)
            if op == OP_MUL:
                t += _COST_MUL
            elif op == OP_INV:
                t += _COST_INV
            elif op in _COST_TRANSCRIPT:
                t += _COST_TRANSCRIPT[op]
            else:
                t += _COST_CHEAP
        total += t
    return total


def _allocate(bundles, keep_alive):
    """Physical registers for the scheduled program: a virtual register lives from the bundle that defines it to the
    last bundle that reads it (registers named in the trace table: to the end), and its physical register becomes free
    for bundles AFTER that one (never within it: the lanes of a bundle run in no particular order)."""
    last_use: Dict[int, int] = {}
    for s, bun in enumerate(bundles):
        for rec in bun:
            if rec is not None:
                for r in _uses(*rec):
                    last_use[r] = s
    end = len(bundles)
    for r in keep_alive:
        last_use[r] = end
    free: List[int] = []
    n_phys = 0
    mapping: Dict[int, int] = {}
    expiring: Dict[int, List[int]] = {}
    out = []
    for s, bun in enumerate(bundles):
        for p in expiring.pop(s, []):
            free.append(p)
        for rec in bun:
            if rec is None:
                out.append((OP_NOP, 0, 0, 0))
                continue
            op, dst, a, c = rec
            pa, pc = a, c
            if op in (OP_ADD, OP_SUB, OP_MUL):
                pa, pc = mapping[a], mapping[c]
            elif op in (OP_NEG, OP_INV, OP_ABSORB_REG, OP_OUT_SCALAR, OP_ASSERT_ZERO):
                pa = mapping[a]
            pd = dst
            if _defines(op):
                if free:
                    pd = free.pop()
                else:
                    pd = n_phys
                    n_phys += 1
                mapping[dst] = pd
                lu = max(last_use.get(dst, s), s)   # a dead value still needs a slot for the write
                if lu < end:
                    expiring.setdefault(lu + 1, []).append(pd)
            out.append((op, pd, pa, pc))
    assert n_phys < 65536
    return out, max(n_phys, 1), mapping


def check_bundles(instrs, lanes: int):
    """The invariants the device relies on (h2v_plan_load checks the same): transcript operations and END only on lane
    0 with the other lanes idle, and no record of a bundle touching a register that another record of it writes."""
    assert len(instrs) % lanes == 0
    for s in range(0, len(instrs), lanes):
        bun = instrs[s:s + lanes]
        if bun[0][0] in TRANSCRIPT_OPS or bun[0][0] == OP_END:
            assert all(r[0] == OP_NOP for r in bun[1:]), "transcript operation shares its bundle"
        written = [r[1] for r in bun if _defines(r[0])]
        assert len(set(written)) == len(written), "two lanes write one register"
        for k, r in enumerate(bun):
            assert k == 0 or (r[0] not in TRANSCRIPT_OPS and r[0] != OP_END), "transcript operation off lane 0"
            for u in _uses(*r):
                assert all(u != w or (j == k and False) for j, w in enumerate(r2[1] for r2 in bun if _defines(r2[0]))), \
                    "a lane reads a register written in the same bundle"


def _schedule_and_allocate(code, keep_alive, lanes: Optional[int], not_before=None):
    """Two schedules of the program (a block of the combiner kernel has 64 lanes, so 64 / L proofs share its LDS
    register file):
      * narrow: the smallest lane count L whose register file fits in LDS - the batch then needs the fewest waves, which
        is what counts when the other kernels of the phase keep every SIMD busy (simple_mul x 4096: L = 2);
      * wide (optional): the L with the smallest estimated run time, kept when it is at least 10 % faster; the launcher
        takes it when the launch would otherwise leave most of the chip idle (h2v_capi.hip: launch_vm).
    L = 1 is the fallback when nothing fits (register file in global memory).  `lanes` forces a single schedule."""
    cands = []
    for L in ((lanes,) if lanes else VM_LANE_CHOICES):
        best = None
        for pack in (False, True):
            bundles = _schedule(code, L, pack, not_before)
            instrs, n_regs, mapping = _allocate(bundles, keep_alive)
            cost = _schedule_cost(bundles)
            if best is None or cost < best[0]:
                best = (cost, instrs, n_regs, mapping, L)
        check_bundles(best[1], L)
        cands.append(best + (best[2] * 32 * (64 // L) <= VM_LDS_BYTES,))
    fitting = [c for c in cands if c[5]]
    if lanes or not fitting:
        c = cands[0]
        return (c[1], c[2], c[3], c[4]), None
    narrow = fitting[0]
    floor = min(c[0] for c in fitting)
    wide = next(c for c in fitting if c[0] <= 1.1 * floor)   # the fewest lanes within 10 % of the best estimate
    wide_out = (wide[4], wide[2], wide[1]) if wide[4] > narrow[4] and wide[0] < narrow[0] * 0.9 else None
    return (narrow[1], narrow[2], narrow[3], narrow[4]), wide_out


# ----------------------------------------------------------------------------- big-integer interpreter
class PlanReject(Exception):
    pass


def run_plan(plan: Plan, proof: bytes, instances: List[int], committed: Optional[bytes] = None,
             stop_before_point: Optional[int] = None, use_wide: bool = False):
    """Executes the plan's bytecode with Python integers (host-side tool: used by the synthetic-proof forger and by
    the CPU tests of the compiler; NOT the verification path).  Returns (term scalars, registers, status).
    status: None = ran to the end, or a reject reason string.  With stop_before_point=i the run stops just before
    READ_POINT of point slot i (used to forge the last proof element)."""
    import hashlib

    instrs, n_regs = (plan.wide[2], plan.wide[1]) if use_wide else (plan.instrs, plan.n_regs)
    regs = [0] * n_regs
    scalars = [0] * plan.n_terms
    acc = bytearray()
    status = None
    n_points_read = 0
    if len(proof) < plan.proof_len and stop_before_point is None:
        return scalars, regs, "short"
    for op, d, a, c in instrs:
        if op == OP_END:
            break
        if op == OP_NOP:
            continue
        if op == OP_ABSORB_REG:
            acc += b"\x01" + regs[a].to_bytes(32, "little")
        elif op == OP_ABSORB_CI:
            acc += b"\x01" + committed
        elif op == OP_LOAD_INSTANCE:
            if not 0 <= instances[a] < R:   # only the canonical encoding of a public input exists on the reference side
                status = status or "scalar"
            regs[d] = instances[a] % R
        elif op == OP_READ_POINT:
            if stop_before_point is not None and n_points_read == stop_before_point:
                return scalars, regs, "stopped"
            off = a | (c << 16)
            acc += b"\x01" + proof[off:off + 48]
            n_points_read += 1
        elif op == OP_READ_SCALAR:
            off = a | (c << 16)
            raw = proof[off:off + 32]
            val = int.from_bytes(raw, "little")
            if val >= R:
                status = status or "scalar"
            regs[d] = val % R
            acc += b"\x01" + raw
        elif op == OP_SQUEEZE:
            acc += b"\x00"
            h = hashlib.blake2b(bytes(acc), digest_size=32).digest()
            h2 = hashlib.blake2b(h, digest_size=32).digest()
            regs[d] = (int.from_bytes(h, "little") + int.from_bytes(h2, "little") * bls.R_2_256) % R
        elif op == OP_CONST:
            regs[d] = plan.consts[a]
        elif op == OP_ADD:
            regs[d] = (regs[a] + regs[c]) % R
        elif op == OP_SUB:
            regs[d] = (regs[a] - regs[c]) % R
        elif op == OP_MUL:
            regs[d] = regs[a] * regs[c] % R
        elif op == OP_NEG:
            regs[d] = (-regs[a]) % R
        elif op == OP_INV:
            if regs[a] == 0:
                status = status or "inverse"
                regs[d] = 0
            else:
                regs[d] = pow(regs[a], R - 2, R)
        elif op == OP_OUT_SCALAR:
            scalars[d] = regs[a]
        elif op == OP_ASSERT_ZERO:
            if regs[a] != 0:
                status = status or "recursion"
        else:
            raise ValueError("bad opcode %d" % op)
    return scalars, regs, status


def plan_stats(plan: Plan) -> dict:
    hist = {}
    for op, *_ in plan.instrs:
        hist[OP_NAMES[op]] = hist.get(OP_NAMES[op], 0) + 1
    return {"instrs": len(plan.instrs), "regs": plan.n_regs, "consts": len(plan.consts), "points": len(plan.points),
            "terms": plan.n_terms, "proof_len": plan.proof_len, "ops": hist, "squeezes": plan.n_squeezes,
            "stream_len": plan.stream_len}


def _main():
    """python -m plutus_halo2_verifier_gen_amd.plan vk.json plan.bin   (or a builtin circuit name instead of vk.json)"""
    import sys
    from . import vk as V

    if len(sys.argv) != 3:
        raise SystemExit(_main.__doc__)
    src, dst = sys.argv[1:]
    if src in V.BUILDERS:
        key, _ = V.BUILDERS[src]()
    else:
        with open(src) as f:
            key = VerifyingKey.from_json(f.read())
    plan = compile_plan(key)
    with open(dst, "wb") as f:
        f.write(plan.to_bytes())
    print(plan_stats(plan))


if __name__ == "__main__":
    _main()
