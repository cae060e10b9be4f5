"""VerifyingKey surface for the Halo2/KZG (BLS12-381) verification hot path.

Mirrors what the reference reads out of `midnight_proofs::plonk::VerifyingKey` + `ParamsVerifierKZG`
when it builds a verifier:
  * InstantiationSpecificData  /root/reference/src/plutus_gen/extraction/data/circuit_types/instantiation_data.rs:26-41
    (fixed/permutation commitments, omega, omega^-1, n^-1, s_g2, n, blinding_factors, transcript_repr,
    public-input count, committed-instance flags); rendered at aiken-verifier/templates/vk_constants.hbs:8-27
  * the ConstraintSystem pieces `extract_circuit` walks (src/plutus_gen/extraction/mod.rs:31-232): query lists,
    gate polynomials, lookup (input, table) expressions, trashcans, permutation columns, degree.

Expressions are nested tuples with the semantics of `Expression<Scalar>` as transpiled at
src/plutus_gen/extraction/data/languages/aiken.rs:122-182:
  ("const", int) | ("fixed", query_idx) | ("advice", query_idx) | ("neg", e) | ("sum", a, b) |
  ("prod", a, b) | ("scaled", e, int)          (query indices are 0-based here; Selector/Instance/Challenge
                                                 are unsupported by the reference and by this build)
"""
from __future__ import annotations

import json
import random
from dataclasses import dataclass, field, asdict
from typing import List, Optional, Tuple

from . import bls12_381 as bls

# Fr multiplicative generator 7, two-adicity 32: ROOT_OF_UNITY = 7^((r-1)/2^32); delta = 7^(2^32)
TWO_ADICITY = 32
ROOT_OF_UNITY = pow(7, (bls.R - 1) >> TWO_ADICITY, bls.R)

ROT_CUR, ROT_NEXT, ROT_PREV = 0, 1, -1


def domain_omega(k: int) -> int:
    w = ROOT_OF_UNITY
    for _ in range(TWO_ADICITY - k):
        w = w * w % bls.R
    return w


@dataclass
class VerifyingKey:
    name: str
    k: int
    blinding_factors: int
    cs_degree: int
    transcript_repr: int
    num_advice_columns: int
    num_fixed_columns: int
    advice_queries: List[Tuple[int, int]]  # (column, rotation)
    fixed_queries: List[Tuple[int, int]]
    instance_queries: List[Tuple[int, int]]
    gates: list  # list of polynomial expressions (flattened over gates)
    lookups: list  # list of (input_exprs, table_exprs)
    trashcans: list  # list of (selector_expr, constraint_exprs)
    permutation_columns: List[Tuple[str, int]]  # ("advice"|"fixed"|"instance", column index)
    fixed_commitments: List[str]  # hex of 48-byte compressed G1
    permutation_commitments: List[str]
    s_g2: str  # hex of 96-byte compressed G2
    n_public_inputs: int
    n_committed_instances: int = 0  # 0 or 1
    # IVC / recursion (instantiation_data.rs:40,117-135): None = no accumulator in the public inputs; a list (possibly
    # empty) of inner verifying keys {name, transcript_repr, fixed_commitments, permutation_commitments} otherwise.
    recursion_vks: Optional[list] = None
    # the JSON form of this object is frozen as docs/vk_schema.json ("h2v-vk/1"); a change of meaning bumps the number
    schema_version: int = 1
    # multi-phase circuits (cs.advice_column_phase() / cs.challenge_phase(), extraction_steps/proof.rs:22-46): the phase of
    # every advice column and of every challenge.  The proof carries the advice commitments phase by phase, each phase
    # followed by the squeezes of its challenges (the challenges only advance the transcript: an expression that used one
    # is a panic in the reference, languages/aiken.rs:150-156).  None = one phase, no challenges (optional fields of the
    # schema: a description without them means exactly that).
    advice_column_phase: Optional[List[int]] = None
    challenge_phase: Optional[List[int]] = None

    # ---- derived (instantiation_data.rs:84-103)
    @property
    def n(self) -> int:
        return 1 << self.k

    @property
    def omega(self) -> int:
        return domain_omega(self.k)

    @property
    def omega_inv(self) -> int:
        return bls.fr_inv(self.omega)

    @property
    def barycentric_weight(self) -> int:
        return bls.fr_inv(self.n)

    @property
    def chunk_len(self) -> int:
        return self.cs_degree - 2

    @property
    def n_perm_chunks(self) -> int:
        c = len(self.permutation_columns)
        return (c + self.chunk_len - 1) // self.chunk_len

    @property
    def quotient_poly_degree(self) -> int:
        return self.cs_degree - 1

    def to_json(self) -> str:
        return json.dumps(asdict(self))

    # ---- the instantiation-specific part, as the generated constant files carry it (wire.py: VKConstants)
    def constants(self):
        from .wire import VKConstants
        return VKConstants(fixed_commitments=list(self.fixed_commitments),
                           permutation_commitments=list(self.permutation_commitments), s_g2=self.s_g2,
                           omega=self.omega, omega_inv=self.omega_inv, barycentric_weight=self.barycentric_weight,
                           transcript_repr=self.transcript_repr, blinding_factors=self.blinding_factors,
                           recursion_vks=[dict(v) for v in self.recursion_vks] if self.recursion_vks else None)

    def with_constants(self, c) -> "VerifyingKey":
        """This circuit shape with the commitments / s_g2 / transcript representation of an exported
        `verifier_key.ak` or `VKConstants.hs` (wire.load_vk_constants).  The domain constants must agree with k."""
        from dataclasses import replace
        from .wire import WireError
        c.check(self.k)
        if c.omega != self.omega:
            raise WireError("omega differs from this build's 2^k-th root of unity")
        if c.blinding_factors != self.blinding_factors:
            raise WireError("blinding factors differ")
        if len(c.fixed_commitments) != len(self.fixed_commitments) or \
                len(c.permutation_commitments) != len(self.permutation_commitments):
            raise WireError("commitment counts differ from the circuit shape")
        rec = self.recursion_vks
        if c.recursion_vks is not None:   # the file carries the inner keys of a recursive circuit: they replace ours
            if rec is None or [v["name"] for v in rec] != [v["name"] for v in c.recursion_vks] or \
                    any(len(a["fixed_commitments"]) != len(b["fixed_commitments"]) or
                        len(a["permutation_commitments"]) != len(b["permutation_commitments"]) for a, b in zip(rec, c.recursion_vks)):
                raise WireError("inner verifying keys differ from the circuit shape")
            rec = [dict(v) for v in c.recursion_vks]
        return replace(self, fixed_commitments=list(c.fixed_commitments),
                       permutation_commitments=list(c.permutation_commitments), s_g2=c.s_g2,
                       transcript_repr=c.transcript_repr, recursion_vks=rec)

    @staticmethod
    def from_json(s: str) -> "VerifyingKey":
        d = json.loads(s)

        def tup(e):
            if isinstance(e, list):
                return tuple(tup(x) for x in e)
            return e

        d["advice_queries"] = [tuple(q) for q in d["advice_queries"]]
        d["fixed_queries"] = [tuple(q) for q in d["fixed_queries"]]
        d["instance_queries"] = [tuple(q) for q in d["instance_queries"]]
        d["gates"] = [tup(g) for g in d["gates"]]
        d["lookups"] = [([tup(e) for e in i], [tup(e) for e in t]) for i, t in d["lookups"]]
        d["trashcans"] = [(tup(s), [tup(e) for e in c]) for s, c in d["trashcans"]]
        d["permutation_columns"] = [tuple(c) for c in d["permutation_columns"]]
        known = {f for f in VerifyingKey.__dataclass_fields__}
        extra = sorted(set(d) - known)
        if extra:
            raise VKError("unknown field(s) in the verifying-key description: %s" % ", ".join(extra))
        key = VerifyingKey(**d)
        validate(key)
        return key


class VKError(ValueError):
    """The verifying-key description is malformed, or describes a circuit the reference cannot emit a verifier for."""


SCHEMA_VERSION = 1
_EXPR_ARITY = {"const": 1, "fixed": 1, "advice": 1, "neg": 1, "sum": 2, "prod": 2, "scaled": 2}
# Expression variants the reference panics on when it transpiles a gate / lookup / trashcan expression
# (extraction/data/languages/aiken.rs:134-156): they must never reach the plan compiler either.
_REFERENCE_PANICS = {"selector": "Selector not supported in custom gate", "instance": "Instance not supported",
                     "challenge": "Challenge not supported"}


def validate(vk: "VerifyingKey", strict_rotations: bool = False) -> None:
    """Checks a verifying-key description against docs/vk_schema.json and against what the reference itself accepts:
    raises VKError for a shape `extract_circuit` / the emitters would reject or panic on.
      * expression nodes: only Constant / Fixed / Advice / Negated / Sum / Product / Scaled; Selector, Instance and
        Challenge are a panic in the reference (languages/aiken.rs:134-156);
      * every query index an expression uses exists; every permutation column has the query at rotation 0 the
        permutation argument reads; committed-instance count 0 or 1 (extraction/mod.rs:41-55);
      * with strict_rotations: only the rotations the emitted Aiken verifier can name - prev, cur, next, last and the
        pre-computed x_rot_2 / x_rot_3 (rotation_description.rs:17-48, verification_h2.hbs:35-36).  This backend itself
        handles any rotation."""
    if vk.schema_version != SCHEMA_VERSION:
        raise VKError("unsupported schema_version %r (this build reads %d)" % (vk.schema_version, SCHEMA_VERSION))
    if not (1 <= vk.k <= TWO_ADICITY):
        raise VKError("k out of range")
    if vk.cs_degree < 3:
        raise VKError("cs_degree must be at least 3 (chunk length cs_degree - 2)")
    if vk.n_committed_instances not in (0, 1):
        raise VKError("the reference supports 0 or 1 committed instance columns (extraction/mod.rs:41-55)")
    if vk.advice_column_phase is not None:
        ph = vk.advice_column_phase
        if len(ph) != vk.num_advice_columns or any((not isinstance(x, int)) or x < 0 or x > 255 for x in ph):
            raise VKError("advice_column_phase: one phase (0..255) per advice column")
        if vk.num_advice_columns and set(range(max(ph) + 1)) - set(ph) - set(vk.challenge_phase or []):
            pass   # (an empty phase is legal: the reference iterates 0..=max_phase and simply emits nothing for it)
    if vk.challenge_phase is not None:
        if any((not isinstance(x, int)) or x < 0 or x > 255 for x in vk.challenge_phase):
            raise VKError("challenge_phase: one phase (0..255) per challenge")
        top = max(vk.advice_column_phase or [0]) if (vk.advice_column_phase or vk.num_advice_columns) else 0
        if any(x > top for x in vk.challenge_phase):
            raise VKError("a challenge of a phase beyond the last advice phase is never squeezed (proof.rs:24-29 iterates 0..=max advice phase)")
    if not (0 <= vk.transcript_repr < bls.R):
        raise VKError("transcript_repr is not a canonical scalar")
    for name, qs, ncols in (("advice", vk.advice_queries, vk.num_advice_columns), ("fixed", vk.fixed_queries, vk.num_fixed_columns),
                            ("instance", vk.instance_queries, vk.n_committed_instances + 1)):
        for col, rot in qs:
            if not (0 <= col < ncols):
                raise VKError("%s query names column %d of %d" % (name, col, ncols))
            if strict_rotations and rot not in (-1, 0, 1, 2, 3):
                raise VKError("rotation %d has no pre-computed point in the emitted verifier (verification_h2.hbs:32-37)" % rot)
        if len(set(qs)) != len(qs):
            raise VKError("duplicate %s query" % name)

    def walk(e, where):
        if not isinstance(e, (tuple, list)) or not e or not isinstance(e[0], str):
            raise VKError("%s: not an expression node: %r" % (where, e))
        tag = e[0]
        if tag in _REFERENCE_PANICS:
            raise VKError("%s: %s (the reference panics here: languages/aiken.rs:134-156)" % (where, _REFERENCE_PANICS[tag]))
        if tag not in _EXPR_ARITY:
            raise VKError("%s: unknown expression node %r" % (where, tag))
        if tag == "const":
            if len(e) != 2 or not isinstance(e[1], int) or not (0 <= e[1] < bls.R):
                raise VKError("%s: constant is not a canonical scalar" % where)
        elif tag in ("fixed", "advice"):
            n = len(vk.fixed_queries if tag == "fixed" else vk.advice_queries)
            if len(e) != 2 or not isinstance(e[1], int) or not (0 <= e[1] < n):
                raise VKError("%s: %s query index %r out of range (%d queries)" % (where, tag, e[1] if len(e) > 1 else None, n))
        elif tag == "neg":
            if len(e) != 2:
                raise VKError("%s: neg takes one operand" % where)
            walk(e[1], where)
        elif tag == "scaled":
            if len(e) != 3 or not isinstance(e[2], int) or not (0 <= e[2] < bls.R):
                raise VKError("%s: scaled takes an expression and a canonical scalar" % where)
            walk(e[1], where)
        else:
            if len(e) != 3:
                raise VKError("%s: %s takes two operands" % (where, tag))
            walk(e[1], where)
            walk(e[2], where)

    for i, g in enumerate(vk.gates):
        walk(g, "gate polynomial %d" % i)
    for i, lk in enumerate(vk.lookups):
        if len(lk) != 2 or len(lk[0]) != len(lk[1]) or not lk[0]:
            raise VKError("lookup %d: input and table expression lists must be non-empty and of equal length" % i)
        for e in list(lk[0]) + list(lk[1]):
            walk(e, "lookup %d" % i)
    for i, tc in enumerate(vk.trashcans):
        if len(tc) != 2:
            raise VKError("trashcan %d: (selector, constraint expressions)" % i)
        walk(tc[0], "trashcan %d selector" % i)
        for e in tc[1]:
            walk(e, "trashcan %d" % i)
    kinds = {"advice": vk.advice_queries, "fixed": vk.fixed_queries, "instance": vk.instance_queries}
    for ty, col in vk.permutation_columns:
        if ty not in kinds:
            raise VKError("permutation column of unknown type %r" % (ty,))
        if (col, 0) not in kinds[ty]:
            raise VKError("permutation column %s[%d] has no query at the current rotation" % (ty, col))
    for h in list(vk.fixed_commitments) + list(vk.permutation_commitments):
        if len(bytes.fromhex(h)) != 48:
            raise VKError("commitments are 48-byte compressed G1 points")
    if len(vk.fixed_commitments) != vk.num_fixed_columns:
        raise VKError("one fixed commitment per fixed column")
    if len(vk.permutation_commitments) != len(vk.permutation_columns):
        raise VKError("one permutation commitment per permutation column")
    if len(bytes.fromhex(vk.s_g2)) != 96:
        raise VKError("s_g2 is a 96-byte compressed G2 point")
    if vk.recursion_vks is not None:
        for inner in vk.recursion_vks:
            if set(inner) != {"name", "transcript_repr", "fixed_commitments", "permutation_commitments"}:
                raise VKError("inner verifying key: fields name / transcript_repr / fixed_commitments / permutation_commitments")


@dataclass
class Trapdoor:
    """Test-SRS secret + discrete logs of the VK commitments (synthetic workloads only).
    rec_dlogs: discrete logs of the inner verifying keys' commitments, in ivc.fixed_bases order."""
    s: int
    fixed_dlogs: List[int] = field(default_factory=list)
    perm_dlogs: List[int] = field(default_factory=list)
    rec_dlogs: List[int] = field(default_factory=list)


# ----------------------------------------------------------------------------- expression helpers
def const(c):
    return ("const", c % bls.R)


def fixed(q):
    return ("fixed", q)


def advice(q):
    return ("advice", q)


def neg(e):
    return ("neg", e)


def add(a, b):
    return ("sum", a, b)


def mul(a, b):
    return ("prod", a, b)


def sub(a, b):
    return ("sum", a, ("neg", b))


def scaled(e, c):
    return ("scaled", e, c % bls.R)


def expr_degree(e) -> int:
    t = e[0]
    if t == "const":
        return 0
    if t in ("fixed", "advice"):
        return 1
    if t in ("neg", "scaled"):
        return expr_degree(e[1])
    if t == "sum":
        return max(expr_degree(e[1]), expr_degree(e[2]))
    if t == "prod":
        return expr_degree(e[1]) + expr_degree(e[2])
    raise ValueError(t)


def expr_op_counts(e, acc=None):
    acc = acc if acc is not None else {"neg": 0, "add": 0, "mul": 0, "const": 0}
    t = e[0]
    if t == "const":
        acc["const"] += 1
    elif t == "neg":
        acc["neg"] += 1
        expr_op_counts(e[1], acc)
    elif t == "scaled":   # one multiplication by one constant (`from_int` in the reference's ScalarOps)
        acc["mul"] += 1
        acc["const"] += 1
        expr_op_counts(e[1], acc)
    elif t == "sum":
        acc["add"] += 1
        expr_op_counts(e[1], acc)
        expr_op_counts(e[2], acc)
    elif t == "prod":
        acc["mul"] += 1
        expr_op_counts(e[1], acc)
        expr_op_counts(e[2], acc)
    return acc


# ----------------------------------------------------------------------------- VK builders
def _commitments(rng: random.Random, count: int):
    dlogs = [rng.randrange(1, bls.R) for _ in range(count)]
    return dlogs, [bls.g1_compress(bls.g1_mul(bls.G1_GEN, d)).hex() for d in dlogs]


def _finish(name, rng, k, bf, degree, n_adv, n_fix, aq, fq, iq, gates, lookups, trash, perm_cols, n_pi, n_ci,
            transcript_repr=None):
    s = rng.randrange(2, bls.R)
    fd, fc = _commitments(rng, n_fix)
    pd, pc = _commitments(rng, len(perm_cols))
    vk = VerifyingKey(
        name=name, k=k, blinding_factors=bf, cs_degree=degree,
        transcript_repr=transcript_repr if transcript_repr is not None else rng.randrange(bls.R),
        num_advice_columns=n_adv, num_fixed_columns=n_fix,
        advice_queries=aq, fixed_queries=fq, instance_queries=iq,
        gates=gates, lookups=lookups, trashcans=trash, permutation_columns=perm_cols,
        fixed_commitments=fc, permutation_commitments=pc,
        s_g2=bls.g2_compress(bls.g2_mul(bls.G2_GEN, s)).hex(),
        n_public_inputs=n_pi, n_committed_instances=n_ci,
    )
    return vk, Trapdoor(s=s, fixed_dlogs=fd, perm_dlogs=pd)


def simple_mul_vk(seed: int = 0x48325631):
    """simple_mul circuit shape (/root/reference/src/circuits/simple_mul_circuit.rs:42-60, :144-159):
    2 advice columns (a0 queried at cur and next, a1 at cur), fixed = constant column + compressed selector,
    gate s_mul * (a0*a1 - a0_next), permutation over [constant, a0, a1] (degree 3 => chunk_len 1 => 3 chunks),
    instance column declared but never queried; 3 public inputs (examples/simple_mul.rs:68)."""
    rng = random.Random(seed)
    aq = [(0, 0), (1, 0), (0, 1)]
    fq = [(0, 0), (1, 0)]
    gate = mul(fixed(1), sub(mul(advice(0), advice(1)), advice(2)))
    return _finish(
        "simple_mul", rng, k=4, bf=5, degree=3, n_adv=2, n_fix=2, aq=aq, fq=fq, iq=[],
        gates=[gate], lookups=[], trash=[], perm_cols=[("fixed", 0), ("advice", 0), ("advice", 1)],
        n_pi=3, n_ci=0,
        transcript_repr=0x53772FDA8C4D27D16E6D1B3B0ED0F0C492414695F8050480AAEB9F0C1257BC6B,
    )


def lookup_table_vk(seed: int = 0x48325632):
    """lookup_table circuit shape (/root/reference/src/circuits/lookup_table_circuit.rs:42-78): 4 advice value
    columns with equality, fixed = tag, complex selector, 2 table columns; no gates; 4 lookup arguments
    [(tag, t_tag), (sel*val, t_val)] => lookup degree 5, chunk_len 3, 2 permutation chunks, 4 h-splits."""
    rng = random.Random(seed)
    aq = [(c, 0) for c in range(4)]
    fq = [(0, 0), (1, 0), (2, 0), (3, 0)]  # tag, selector, t_tag, t_val
    lookups = [([fixed(0), mul(fixed(1), advice(c))], [fixed(2), fixed(3)]) for c in range(4)]
    return _finish(
        "lookup_table", rng, k=6, bf=5, degree=5, n_adv=4, n_fix=4, aq=aq, fq=fq, iq=[],
        gates=[], lookups=lookups, trash=[], perm_cols=[("advice", c) for c in range(4)], n_pi=1, n_ci=0,
    )


def _random_expr(rng, n_adv_q, n_fix_q, n_mul, n_add, n_neg, max_degree):
    """Random expression with exactly the requested op counts (products limited by max_degree by pairing
    a fresh leaf with each multiplication above the degree cap: Scaled keeps degree)."""
    def leaf():
        c = rng.random()
        if c < 0.55:
            return advice(rng.randrange(n_adv_q))
        if c < 0.9 and n_fix_q:
            return fixed(rng.randrange(n_fix_q))
        return const(rng.randrange(bls.R))

    e = leaf()
    ops = ["m"] * n_mul + ["a"] * n_add + ["n"] * n_neg
    rng.shuffle(ops)
    for op in ops:
        if op == "n":
            e = neg(e)
        elif op == "a":
            e = add(e, leaf()) if rng.random() < 0.5 else add(leaf(), e)
        else:
            if expr_degree(e) + 1 <= max_degree:
                e = mul(e, leaf()) if rng.random() < 0.5 else mul(leaf(), e)
            else:
                e = scaled(e, rng.randrange(bls.R))
    return e


def _split_counts(rng, total, parts):
    cuts = sorted(rng.randrange(total + 1) for _ in range(parts - 1))
    prev = 0
    out = []
    for c in cuts + [total]:
        out.append(c - prev)
        prev = c
    return out


def _shaped_vk(name, seed, *, k, degree, n_adv, n_fix, n_cc, lookup_arg_exprs, gate_exprs, gate_ops,
               adv_rot_sets, n_pi, n_ci, bf=6, trash_exprs=()):
    """Shape-faithful synthetic VK for circuits whose real VK cannot be extracted offline
    (needs keygen_vk; SURVEY.md §7 'Hard parts').  adv_rot_sets: per advice column the rotations queried."""
    rng = random.Random(seed)
    aq = []
    for col, rots in enumerate(adv_rot_sets):
        for r in rots:
            aq.append((col, r))
    fq = [(c, 0) for c in range(n_fix)]
    iq = []
    if n_ci:
        iq.append((0, 0))
        iq.append((1, 0))
    else:
        iq.append((0, 0))
    muls = _split_counts(rng, gate_ops["mul"], gate_exprs)
    adds = _split_counts(rng, gate_ops["add"], gate_exprs)
    negs = _split_counts(rng, gate_ops["neg"], gate_exprs)
    gates = [_random_expr(rng, len(aq), len(fq), muls[i], adds[i], negs[i], degree) for i in range(gate_exprs)]
    lookups = []
    for n_e in lookup_arg_exprs:
        ins = [_random_expr(rng, len(aq), len(fq), 1, 1, 0, 2) for _ in range(n_e)]
        tabs = [fixed(rng.randrange(len(fq))) for _ in range(n_e)]
        lookups.append((ins, tabs))
    # permutation columns: advice columns first (all queried at cur), then fixed, then the instance columns
    perm_cols = []
    cur_adv = [c for c, rots in enumerate(adv_rot_sets) if 0 in rots]
    for c in cur_adv[: max(0, n_cc - 1)]:
        perm_cols.append(("advice", c))
    while len(perm_cols) < n_cc - 1:
        perm_cols.append(("fixed", len(perm_cols) % n_fix))
    perm_cols.append(("instance", 1 if n_ci else 0))
    # trashcans (selector expression, constraint expressions): emitters/aiken.rs:444-461
    trash = []
    for n_e in trash_exprs:
        trash.append((fixed(rng.randrange(len(fq))),
                      [_random_expr(rng, len(aq), len(fq), 1, 1, 0, 2) for _ in range(n_e)]))
    return _finish(name, rng, k=k, bf=bf, degree=degree, n_adv=n_adv, n_fix=n_fix, aq=aq, fq=fq, iq=iq,
                   gates=gates, lookups=lookups, trash=trash, perm_cols=perm_cols, n_pi=n_pi, n_ci=n_ci)


def atms_with_lookups_vk(seed: int = 0x48325633):
    """ATMS + lookup shape: 11 advice / 21 fixed evaluations, 5 gate polynomials, 1 lookup argument
    (/root/reference/aiken-verifier/templates/gates_test.hbs:27-59, src/circuits/atms_with_lookups_circuit.rs:36)."""
    adv = [[0, 1]] * 4 + [[0]] * 3  # 7 columns, 11 queries
    return _shaped_vk("atms_with_lookups", seed, k=14, degree=5, n_adv=7, n_fix=21, n_cc=8,
                      lookup_arg_exprs=[2], gate_exprs=5,
                      gate_ops={"mul": 60, "add": 50, "neg": 10}, adv_rot_sets=adv, n_pi=2, n_ci=0)


def _exact_gate_exprs(rng, n_adv_q, n_fix_q, n_exprs, n_mul, n_add, n_neg, n_const, max_factors=4):
    """`n_exprs` gate polynomials whose Expression trees hold EXACTLY n_mul multiplications (Product or Scaled nodes),
    n_add Sum nodes, n_neg Negated nodes and n_const constants in total - the figures the reference's cost model keeps
    per chip (stats/chips/types/scalar_ops.rs; a Scaled node is one multiplication by one `from_int` constant).
    Shape: every polynomial is a sum of terms, a term a product of 1..max_factors advice / fixed evaluations, optionally
    scaled by a constant, optionally negated - sums of low-degree products, as foreign-field and spread-table gates are
    (a chain of multiplications by fresh leaves would exceed the circuit degree long before 641 multiplications)."""
    n_terms = n_add + n_exprs
    assert n_const <= n_terms and n_neg <= n_terms
    extra = n_mul - n_const            # Product nodes = sum over terms of (factors - 1)
    assert 0 <= extra <= n_terms * (max_factors - 1), "op counts do not fit sums of products of this degree"
    factors = [1] * n_terms
    open_terms = list(range(n_terms))
    for _ in range(extra):
        t = open_terms[rng.randrange(len(open_terms))]
        factors[t] += 1
        if factors[t] == max_factors:
            open_terms.remove(t)
    scaled_terms = set(rng.sample(range(n_terms), n_const))
    neg_terms = set(rng.sample(range(n_terms), n_neg))

    def leaf():
        return advice(rng.randrange(n_adv_q)) if rng.random() < 0.6 else fixed(rng.randrange(n_fix_q))

    terms = []
    for t in range(n_terms):
        e = leaf()
        for _ in range(factors[t] - 1):
            e = mul(e, leaf())
        if t in scaled_terms:
            e = scaled(e, rng.randrange(1, bls.R))
        if t in neg_terms:
            e = neg(e)
        terms.append(e)
    rng.shuffle(terms)
    per = [1 + c for c in _split_counts(rng, n_add, n_exprs)]
    out, k = [], 0
    for n in per:
        e = terms[k]
        for t in terms[k + 1:k + n]:
            e = add(e, t)
        out.append(e)
        k += n
    assert k == n_terms
    return out


def chip_profile_vk(name, seed, profile, *, advice_rotations, lookup_input, n_adv_cc, k=17, bf=6, n_pi=1, n_ci=0,
                    instance_in_permutation=False):
    """A key with the shape the reference's chip profile records (docs/chip_profiles.json, written by
    src/plutus_gen/stats/profile.rs:54-162; the fixture tests/golden/reference_kats.json "chip_profiles" holds the
    numbers): advice / fixed columns, copy-constrained columns, lookup arguments, gate-expression count and op counts,
    and - through the rotation sets - the commitment map.  What the profile does not hold is given by the caller and
    cited there: which rotations each advice column is queried at and what one lookup input expression looks like.

    Conventions of the profile that this follows (tests/test_host_logic.py::test_chip_shapes_match_reference_profile):
      * pi = 1, ci = 0, and `copy_constraints` EXCLUDES the instance column (profile.rs:28-31): with
        instance_in_permutation=False the permutation runs over n_adv_cc advice columns and copy_constraints - n_adv_cc
        fixed columns only.  The example wrappers put the public-input column and the committed-instance column back
        (stats/estimate/build.rs:166-177);
      * lookup ARGUMENTS = (proof_commitments - advice - ceil(cc / (degree - 2)) - degree - 2) / 3, `lookups` = their
        expressions in total; the (degree - 1) + 1 vanishing commitments and the 2 of the multi-open argument are the rest;
      * gate_ops / lookup_ops are the per-expression counts plus one add and one mul per expression after the first of
        every gate / argument (ScalarExpression::batch_expressions, stats/chips/types/expression.rs:52-66)."""
    rng = random.Random(seed)
    degree = profile["degree"]
    n_adv, n_fix, cc = profile["advice_cols"], profile["fixed_cols"], profile["copy_constraints"]
    assert len(advice_rotations) == n_adv
    chunks = -(-cc // (degree - 2))
    n_lookup_args, rem = divmod(profile["proof_commitments"] - n_adv - chunks - degree - 2, 3)
    assert rem == 0 and n_lookup_args >= 0
    aq = [(c, r) for c, rots in enumerate(advice_rotations) for r in rots]
    fq = [(c, 0) for c in range(n_fix)]
    g = profile["gate_ops"]
    batching = profile["gate_expressions"] - profile["gates"]
    gates = _exact_gate_exprs(rng, len(aq), len(fq), profile["gate_expressions"], g["mul"] - batching,
                              g["add"] - batching, g["neg"], g["from_int"])
    lookups = []
    if n_lookup_args:
        per_arg, rem = divmod(profile["lookups"], n_lookup_args)
        assert rem == 0
        for _ in range(n_lookup_args):
            ins = [lookup_input(rng, len(aq), len(fq)) for _ in range(per_arg)]
            tabs = [fixed(rng.randrange(len(fq))) for _ in range(per_arg)]
            lookups.append((ins, tabs))
    cur_adv = [c for c, rots in enumerate(advice_rotations) if 0 in rots]
    perm_cols = [("advice", c) for c in cur_adv[:n_adv_cc]] + [("fixed", c) for c in range(cc - n_adv_cc)]
    assert len(perm_cols) == cc
    iq = []
    if n_ci:
        iq.append((0, 0))               # the committed column is opened at x (extraction/mod.rs:126-138)
    if instance_in_permutation:
        if n_ci:
            perm_cols.append(("instance", 0))
        perm_cols.append(("instance", n_ci))
        iq.append((n_ci, 0))
    return _finish(name, rng, k=k, bf=bf, degree=degree, n_adv=n_adv, n_fix=n_fix, aq=aq, fq=fq, iq=iq, gates=gates,
                   lookups=lookups, trash=[], perm_cols=perm_cols, n_pi=n_pi, n_ci=n_ci)


# docs/chip_profiles.json "sha256" / "secp256k1" as committed in the reference (the CPU suite checks these literals
# against the extracted fixture, so a drift of either side fails a test).
SHA256_PROFILE = {"degree": 5, "advice_cols": 8, "fixed_cols": 25, "copy_constraints": 7, "gates": 14,
                  "gate_expressions": 22, "lookups": 6, "proof_commitments": 24, "vk_commitments": 32, "evals": 64,
                  "gate_ops": {"neg": 22, "add": 173, "sub": 0, "mul": 156, "from_int": 116},
                  "lookup_ops": {"neg": 0, "add": 4, "sub": 0, "mul": 10, "from_int": 0}, "proof_size": 3200,
                  "commitment_map_sets": [[1, 36], [2, 3], [3, 2], [2, 2], [3, 8]]}
SECP256K1_PROFILE = {"degree": 5, "advice_cols": 9, "fixed_cols": 23, "copy_constraints": 10, "gates": 11,
                     "gate_expressions": 22, "lookups": 10, "proof_commitments": 23, "vk_commitments": 33, "evals": 64,
                     "gate_ops": {"neg": 67, "add": 476, "sub": 0, "mul": 641, "from_int": 403},
                     "lookup_ops": {"neg": 10, "add": 19, "sub": 0, "mul": 19, "from_int": 10}, "proof_size": 3152,
                     "commitment_map_sets": [[1, 37], [2, 2], [3, 3], [2, 2], [3, 7]]}


def sha256_vk(seed: int = 0x48325634, chip_alone: bool = False):
    """sha256 chip shape, pinned on /root/reference/docs/chip_profiles.json "sha256" (SHA256_PROFILE): 8 advice columns,
    ALL queried at {prev, cur, next} (stats/chips/primitives/hash/sha256.rs:9-22; six of them copy-constrained), 25 fixed
    at {cur}, 7 copy-constrained columns (6 advice + native's fixed_values, chips/primitives/native.rs:34), 2 lookup
    arguments of 3 expressions `q_lookup * advice` (sha256.rs:95-110), 22 gate expressions in 14 gates with 148 mul /
    165 add / 22 neg / 116 constants of their own, degree 5 => commitment-map sets
    {cur}: 36, {cur,next}: 3, {cur,next,last}: 2, {prev,cur}: 2, {prev,cur,next}: 8 and 57 MSM terms.
    chip_alone=True is the profile's own setting (pi = 1, ci = 0, instance column outside the permutation).  The
    default is the example's wrapper (examples/sha256.rs:42,131-137: 32 public inputs, one committed instance column
    = identity): both instance columns join the permutation (build.rs:166-177: 9 columns, still 3 chunks) and the
    committed column is opened at x: {cur} 36 -> 39, 60 MSM terms, 3 more evaluations.
    The profile's `evals` = 64 / `proof_size` = 3200 count one evaluation per advice / fixed COLUMN (profile.rs:112);
    a proof carries one per QUERY (24 + 25 here, as stats/estimate/build.rs:186-188 counts them): 80 scalars, 3712 B."""
    adv = [[-1, 0, 1]] * 8
    return chip_profile_vk("sha256", seed, SHA256_PROFILE, advice_rotations=adv, n_adv_cc=6,
                           lookup_input=lambda rng, na, nf: mul(fixed(rng.randrange(nf)), advice(rng.randrange(na))),
                           n_pi=1 if chip_alone else 32, n_ci=0 if chip_alone else 1,
                           instance_in_permutation=not chip_alone)


def secp256k1_vk(seed: int = 0x48325635, chip_alone: bool = False):
    """secp256k1 foreign-field chip shape, pinned on docs/chip_profiles.json "secp256k1" (SECP256K1_PROFILE): 9 advice
    columns - seven at {prev, cur, next}, one at {prev, cur}, one at {cur}: the only assignment that gives the profile's
    commitment map {cur}: 37 = 23 fixed + 10 sigma + 2 vanishing + 1 permuted table + 1 advice, {cur,next}: 2 = z_0 + the
    lookup product, {cur,next,last}: 3 = z_1..z_3, {prev,cur}: 2 = permuted input + 1 advice, {prev,cur,next}: 7 -,
    23 fixed, 10 copy-constrained columns (9 advice + fixed_values), 1 lookup argument of 10 expressions (one negation,
    one addition, one multiplication, one constant each), 22 gate expressions in 11 gates with 630 mul / 465 add /
    67 neg / 403 constants of their own, degree 5 => 57 MSM terms.
    Default = wrapper with 4 public inputs and a committed instance column (12 permutation columns, still 4 chunks;
    60 terms); chip_alone=True = the profile's setting.  Proof: 79 scalars / 3632 B against the profile's per-column
    count of 64 / 3152 B (see sha256_vk)."""
    adv = [[-1, 0, 1]] * 7 + [[-1, 0]] + [[0]]
    return chip_profile_vk("secp256k1", seed, SECP256K1_PROFILE, advice_rotations=adv, n_adv_cc=9,
                           lookup_input=lambda rng, na, nf: scaled(sub(advice(rng.randrange(na)), advice(rng.randrange(na))),
                                                                   rng.randrange(1, bls.R)),
                           n_pi=1 if chip_alone else 4, n_ci=0 if chip_alone else 1,
                           instance_in_permutation=not chip_alone)


def trashcan_mix_vk(seed: int = 0x48325637):
    """Small chip mix with two trashcan arguments, a lookup and a committed instance: every optional argument kind of
    the verifier in one key (trash squeeze + commitments proof.rs:68-75, identities emitters/aiken.rs:444-461)."""
    # rotations beyond prev / cur / next become Custom(n) (rotation_description.rs:17-48; x_rot_2, x_rot_3 of
    # verification_h2.hbs:35-36)
    adv = [[0, 1, -1, 2]] * 1 + [[0, 3]] * 1 + [[0, 1]] * 1 + [[0]] * 2
    return _shaped_vk("trashcan_mix", seed, k=9, degree=4, n_adv=5, n_fix=7, n_cc=5, lookup_arg_exprs=[2],
                      gate_exprs=3, gate_ops={"mul": 12, "add": 10, "neg": 2}, adv_rot_sets=adv, n_pi=5, n_ci=1,
                      trash_exprs=(2, 1))


def phased_vk(seed: int = 0x48325638):
    """A multi-phase circuit: advice columns in three phases and challenges squeezed after phases 0 and 1
    (extraction_steps/proof.rs:22-46: per phase, that phase's advice commitments, then that phase's challenges) - otherwise
    a small chip mix with a lookup and a trashcan."""
    adv = [[0, 1]] * 2 + [[0]] * 3
    vk, td = _shaped_vk("phased", seed, k=8, degree=5, n_adv=5, n_fix=6, n_cc=4, lookup_arg_exprs=[2], gate_exprs=3,
                        gate_ops={"mul": 14, "add": 11, "neg": 2}, adv_rot_sets=adv, n_pi=2, n_ci=0, trash_exprs=(2,))
    vk.advice_column_phase = [0, 1, 0, 2, 1]
    vk.challenge_phase = [0, 1, 1]
    return vk, td


def ivc_vk(seed: int = 0x48325636):
    """IVC-shaped circuit (examples/ivc.rs, src/circuits/ivc_circuit.rs): a small chip mix whose public inputs carry
    the verifying-key hash, the collapsed accumulator of the previous step and the fixed-base scalars
    (emitters/aiken.rs:648-757, docs/algorithms.html "Recursion (IVC)"), with one inner verifying key."""
    adv = [[0, 1]] * 2 + [[0]] * 3
    rng = random.Random(seed ^ 0x1)
    inner_f_d, inner_f = _commitments(rng, 2)
    inner_p_d, inner_p = _commitments(rng, 2)
    inner = [{"name": "inner", "transcript_repr": rng.randrange(bls.R), "fixed_commitments": inner_f,
              "permutation_commitments": inner_p}]
    n_fix, n_cc = 6, 4
    f_len = 1 + n_fix + n_cc + 4          # -G1, this key's commitments, the inner key's
    n_pi = 2 + f_len + 10 + 1             # two key hashes, fixed-base scalars, serialised accumulator, one app input
    vk, td = _shaped_vk("ivc", seed, k=10, degree=5, n_adv=5, n_fix=n_fix, n_cc=n_cc, lookup_arg_exprs=[2],
                        gate_exprs=4, gate_ops={"mul": 24, "add": 20, "neg": 4}, adv_rot_sets=adv, n_pi=n_pi, n_ci=0)
    vk.recursion_vks = inner
    td.rec_dlogs = inner_f_d + inner_p_d
    return vk, td


BUILDERS = {
    "simple_mul": simple_mul_vk,
    "lookup_table": lookup_table_vk,
    "atms_with_lookups": atms_with_lookups_vk,
    "sha256": sha256_vk,
    "secp256k1": secp256k1_vk,
    "ivc": ivc_vk,
    "trashcan_mix": trashcan_mix_vk,
    "phased": phased_vk,
}
