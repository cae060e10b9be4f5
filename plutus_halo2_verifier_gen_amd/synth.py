"""Synthetic proof batches for tests and benchmarks (no prover exists in this environment).

SURVEY.md §7 "Synthetic proofs without a prover": the build controls its own test-SRS secret s (s_g2 = s*G2), so
any plan-shaped transcript can be made to ACCEPT: choose every proof commitment as a known multiple of G1, every
evaluation uniformly at random, replay the transcript to obtain the final MSM scalars s_t, and set the last proof
element  pi = ((sum_{t != pi} s_t * dlog(B_t)) / (s - x3)) * G1 .  pi is the last thing read and never hashed
before a squeeze (/root/reference/src/plutus_gen/extraction/pcs/kzg.rs:76-78), so the scalars do not depend on
it.  Then  er = final_com - v*G1 + x3*pi = s*pi  and  e(pi, s*G2) == e(er, G2).

Negative cases follow the reference's recipe (flip a byte of the first scalar:
/root/reference/examples/simple_mul.rs:87-95) plus encodings the Rust reader rejects.

Host-side tooling only (big-integer Python); the verification path never touches this module.
"""
from __future__ import annotations

import os
import random
from concurrent.futures import ProcessPoolExecutor
from dataclasses import dataclass
from typing import List, Optional

from . import bls12_381 as bls
from .plan import Plan, TERM_COMMITTED_INSTANCE, TERM_PROOF_POINT, TERM_VK_BASE, compile_plan, run_plan
from .vk import Trapdoor, VerifyingKey

P, R = bls.P, bls.R


# ----------------------------------------------------------------------------- fast fixed-base G1 arithmetic
class _FixedBase:
    """8-bit-window table of multiples of G1 (affine), built once per process."""

    def __init__(self):
        self.win = 8
        n_win = 32
        tables = []
        base = bls.G1_GEN
        for _ in range(n_win):
            row = [None]
            acc = None
            for _ in range(255):
                acc = bls.g1_add(acc, base)
                row.append(acc)
            tables.append(row)
            base = bls.g1_add(acc, base)  # 256 * base
        self.tables = tables

    def mul(self, k: int):
        k %= R
        X, Y, Z = 1, 1, 0
        for w in range(32):
            d = (k >> (8 * w)) & 0xFF
            if d:
                x2, y2 = self.tables[w][d]
                X, Y, Z = bls._jac_add_affine(X, Y, Z, x2, y2)
        if Z == 0:
            return None
        zi = bls.fp_inv(Z)
        zi2 = zi * zi % P
        return (X * zi2 % P, Y * zi2 * zi % P)


_FB: Optional[_FixedBase] = None


def fixed_base() -> _FixedBase:
    global _FB
    if _FB is None:
        _FB = _FixedBase()
    return _FB


def _batch_affine_add(pairs):
    """[(P, Q)] -> [P + Q] with ONE field inversion (P != +-Q, neither infinity: random points)."""
    dens = [(q[0] - p[0]) % P for p, q in pairs]
    pre = []
    acc = 1
    for d in dens:
        acc = acc * d % P
        pre.append(acc)
    inv = bls.fp_inv(acc)
    out = [None] * len(pairs)
    for i in range(len(pairs) - 1, -1, -1):
        di = inv * (pre[i - 1] if i else 1) % P
        inv = inv * dens[i] % P
        (x1, y1), (x2, y2) = pairs[i]
        lam = (y2 - y1) * di % P
        x3 = (lam * lam - x1 - x2) % P
        out[i] = (x3, (lam * (x1 - x3) - y1) % P)
    return out


@dataclass
class Batch:
    """A batch in the layout the C-ABI takes (include/h2v.h: h2v_batch)."""
    n: int
    proofs: bytes
    proof_off: List[int]
    instances: bytes  # n * n_pi * 32 B little-endian
    committed: Optional[bytes]  # n * 48 B or None
    expected: List[int]  # 1 accept / 0 reject

    def proof(self, i) -> bytes:
        return self.proofs[self.proof_off[i]:self.proof_off[i + 1]]

    def instance_ints(self, i, n_pi):
        b = self.instances[32 * n_pi * i:32 * n_pi * (i + 1)]
        return [int.from_bytes(b[32 * k:32 * k + 32], "little") for k in range(n_pi)]

    def ci(self, i):
        return None if self.committed is None else self.committed[48 * i:48 * i + 48]


def _forge_range(args):
    vk_json, td, lo, hi, seed, ci_identity = args
    vk = VerifyingKey.from_json(vk_json)
    plan = compile_plan(vk)
    return _forge_with_plan(vk, td, plan, lo, hi, seed, ci_identity)


def _forge_with_plan(vk: VerifyingKey, td: Trapdoor, plan: Plan, lo: int, hi: int, seed: int, ci_identity: bool):
    fb = fixed_base()
    n_pts = len(plan.points)
    pi_pt = plan.pi_point
    assert pi_pt == n_pts - 1
    # per-slot arithmetic progression of commitments: C_j(i) = (c_j + i*d_j) * G, advanced with batched affine adds
    slot_rng = random.Random((seed << 8) ^ 0x51)
    c0 = [slot_rng.randrange(1, R) for _ in range(n_pts)]
    dd = [slot_rng.randrange(1, R) for _ in range(n_pts)]
    deltas = [fb.mul(d) for d in dd]
    cur = [fb.mul((c0[j] + lo * dd[j]) % R) for j in range(n_pts)]
    vk_dlogs = []
    # dlog of every VK base in plan order: fixed / permutation commitments via the trapdoor, -G1 -> -1
    fixed_pts = {bls.g1_decompress(bytes.fromhex(h), False): d for h, d in zip(vk.fixed_commitments, td.fixed_dlogs)}
    perm_pts = {bls.g1_decompress(bytes.fromhex(h), False): d for h, d in zip(vk.permutation_commitments, td.perm_dlogs)}
    rec_pts = {}
    if vk.recursion_vks is not None:
        from . import ivc
        n_own = 1 + len(vk.fixed_commitments) + len(vk.permutation_commitments)
        rec_pts = {bls.g1_decompress(bytes.fromhex(h), False): d for h, d in zip(ivc.fixed_bases(vk)[n_own:], td.rec_dlogs)}
    for pt in plan.vk_bases:
        if pt == bls.g1_neg(bls.G1_GEN):
            vk_dlogs.append(R - 1)
        elif pt in fixed_pts:
            vk_dlogs.append(fixed_pts[pt])
        elif pt in perm_pts:
            vk_dlogs.append(perm_pts[pt])
        else:
            vk_dlogs.append(rec_pts[pt])
    proofs, insts, cis = [], [], []
    for i in range(lo, hi):
        rng = random.Random((seed << 20) ^ i)
        dl = [(c0[j] + i * dd[j]) % R for j in range(n_pts)]
        buf = bytearray(plan.proof_len)
        for j in range(n_pts - 1):
            buf[plan.points[j]:plan.points[j] + 48] = bls.g1_compress(cur[j])
        pt_bytes = set()
        for j in range(n_pts):
            pt_bytes.update(range(plan.points[j], plan.points[j] + 48))
        # every other 32-byte record is a scalar: uniform Fr
        off = 0
        while off < plan.proof_len:
            if off in pt_bytes:
                off += 48
            else:
                buf[off:off + 32] = rng.randrange(R).to_bytes(32, "little")
                off += 32
        buf[plan.points[pi_pt]:plan.points[pi_pt] + 48] = bls.g1_compress(bls.G1_GEN)  # placeholder
        if vk.name == "simple_mul" and i == 0:
            inst = [42] * vk.n_public_inputs  # the literal inputs of examples/simple_mul.rs:68
        else:
            inst = [rng.randrange(R) for _ in range(vk.n_public_inputs)]
        if vk.recursion_vks is not None:
            ivc.make_accumulator(vk, td, rng, inst)  # a valid accumulator: the fold keeps the pairing equation true
        ci_bytes, ci_dlog = None, 0
        if vk.n_committed_instances:
            if ci_identity:
                ci_bytes = bls.g1_compress(None)  # identity, as in examples/sha256.rs:133
            else:
                ci_dlog = rng.randrange(1, R)
                ci_bytes = bls.g1_compress(fb.mul(ci_dlog))
        while True:
            scalars, _regs, status = run_plan(plan, bytes(buf), inst, ci_bytes)
            assert status is None, status
            total = 0
            x3 = None
            for t, (kind, idx) in enumerate(plan.terms[:plan.n_main_terms]):
                if kind == TERM_PROOF_POINT and idx == pi_pt:
                    x3 = scalars[t]
                    continue
                if kind == TERM_PROOF_POINT:
                    d = dl[idx]
                elif kind == TERM_VK_BASE:
                    d = vk_dlogs[idx]
                else:
                    d = ci_dlog
                total = (total + scalars[t] * d) % R
            if (td.s - x3) % R != 0:
                break
            buf[plan.proof_len - 80] ^= 1  # astronomically unlikely; perturb a scalar and retry
        p = total * bls.fr_inv(td.s - x3) % R
        buf[plan.points[pi_pt]:plan.points[pi_pt] + 48] = bls.g1_compress(fb.mul(p))
        proofs.append(bytes(buf))
        insts.append(b"".join(v.to_bytes(32, "little") for v in inst))
        if ci_bytes is not None:
            cis.append(ci_bytes)
        if i + 1 < hi:
            cur = _batch_affine_add(list(zip(cur, deltas)))
    return proofs, insts, cis


def forge_batch(vk: VerifyingKey, td: Trapdoor, n: int, seed: int = 1, workers: Optional[int] = None,
                plan: Optional[Plan] = None, ci_identity: bool = True) -> Batch:
    """n accepting proofs for `vk` (all valid)."""
    if workers is None:
        workers = min(os.cpu_count() or 1, 16)
    if n < 64 or workers <= 1:
        plan = plan or compile_plan(vk)
        parts = [_forge_with_plan(vk, td, plan, 0, n, seed, ci_identity)]
    else:
        vk_json = vk.to_json()
        chunks = [(vk_json, td, n * w // workers, n * (w + 1) // workers, seed, ci_identity) for w in range(workers)]
        with ProcessPoolExecutor(max_workers=workers) as ex:
            parts = list(ex.map(_forge_range, chunks))
    proofs = [p for part in parts for p in part[0]]
    insts = [p for part in parts for p in part[1]]
    cis = [p for part in parts for p in part[2]]
    off = [0]
    for p in proofs:
        off.append(off[-1] + len(p))
    return Batch(n=n, proofs=b"".join(proofs), proof_off=off, instances=b"".join(insts),
                 committed=b"".join(cis) if cis else None, expected=[1] * n)


# ----------------------------------------------------------------------------- negative cases
CORRUPTIONS = ("flip_first_scalar", "flip_last_scalar", "bad_point_flag", "point_not_on_curve", "point_not_in_subgroup",
               "noncanonical_scalar", "noncanonical_instance", "wrong_public_input", "wrong_pi", "truncated",
               "infinity_commitment",
               # recursion (IVC) only; None for plans without an accumulator
               "acc_limb", "acc_scalar", "acc_fixed_scalar", "acc_sign", "acc_vk_hash")


def _first_scalar_offset(plan: Plan) -> int:
    pts = set(plan.points)
    off = 0
    while off in pts:
        off += 48
    return off


def corrupt(plan: Plan, proof: bytes, inst: bytes, kind: str, rng: random.Random):
    """Returns (proof, instances) for one rejecting variant of an accepting proof."""
    buf = bytearray(proof)
    inst = bytearray(inst)
    if kind == "flip_first_scalar":
        # examples/simple_mul.rs:87-95: byte 48*G+2 where G = number of leading G1 elements
        o = _first_scalar_offset(plan) + 2
        buf[o] ^= 0xFF
        if int.from_bytes(buf[o - 2:o + 30], "little") >= R:
            buf[o] ^= 0xFF
            buf[o] ^= 0x01
    elif kind == "flip_last_scalar":
        o = plan.points[plan.pi_point] - 32
        buf[o] ^= 0x01
    elif kind == "bad_point_flag":
        j = rng.randrange(len(plan.points))
        buf[plan.points[j]] &= 0x7F
    elif kind == "point_not_on_curve":
        j = rng.randrange(len(plan.points))
        while True:
            x = rng.randrange(P)
            if bls.fp_sqrt(x * x * x + 4) is None:
                break
        raw = bytearray(x.to_bytes(48, "big"))
        raw[0] |= 0x80
        buf[plan.points[j]:plan.points[j] + 48] = raw
    elif kind == "point_not_in_subgroup":
        j = rng.randrange(len(plan.points))
        while True:
            x = rng.randrange(P)
            yy = bls.fp_sqrt(x * x * x + 4)
            if yy is not None and not bls.g1_in_subgroup((x, yy)):
                break
        buf[plan.points[j]:plan.points[j] + 48] = bls.g1_compress((x, yy))
    elif kind == "noncanonical_scalar":
        o = _first_scalar_offset(plan)
        v = int.from_bytes(buf[o:o + 32], "little")
        if v + R >= 1 << 256:
            v = 5
        buf[o:o + 32] = (v + R).to_bytes(32, "little")
    elif kind == "noncanonical_instance":
        # the same public input as v + r: one field element, a second 32-byte encoding the reference cannot express
        if len(inst) == 0:
            return None
        k = rng.randrange(len(inst) // 32)
        v = int.from_bytes(inst[32 * k:32 * k + 32], "little")
        if v + R >= 1 << 256:
            return None
        inst[32 * k:32 * k + 32] = (v + R).to_bytes(32, "little")
    elif kind == "wrong_public_input":
        if len(inst) == 0:
            return None
        v = (int.from_bytes(inst[0:32], "little") + 1) % R
        inst[0:32] = v.to_bytes(32, "little")
    elif kind == "wrong_pi":
        o = plan.points[plan.pi_point]
        buf[o:o + 48] = bls.g1_compress(fixed_base().mul(rng.randrange(1, R)))
    elif kind == "truncated":
        buf = buf[:-1]
    elif kind == "infinity_commitment":
        buf[plan.points[0]:plan.points[0] + 48] = bls.g1_compress(None)
    elif kind.startswith("acc_"):
        if not plan.is_recursive:
            return None
        lx_hi, lx_lo, ly_hi, ly_lo, rx_hi, rx_lo, ry_hi, ry_lo = plan.acc_coords
        geti = lambda k: int.from_bytes(inst[32 * k:32 * k + 32], "little")

        def seti(k, v):
            inst[32 * k:32 * k + 32] = (v % R).to_bytes(32, "little")

        # scalar positions follow the coordinates (ivc.layout): left scalar after left y, right scalar after right y
        if kind == "acc_limb":          # another x: off the curve, outside the subgroup, or simply a different point
            seti(lx_lo, geti(lx_lo) + 1 + rng.randrange(1 << 32))
        elif kind == "acc_scalar":
            seti(ly_hi + 1, geti(ly_hi + 1) + 1)
        elif kind == "acc_fixed_scalar":
            seti(ry_hi + 2 + rng.randrange(3), rng.randrange(R))
        elif kind == "acc_sign":        # y -> p - y: the same x with the other parity flag = the negated point
            from . import ivc
            y = ivc.coord(geti(ry_hi), geti(ry_lo))
            hi, lo = ivc.split_coord(bls.P - y)
            seti(ry_hi, hi)
            seti(ry_lo, lo)
        elif kind == "acc_vk_hash":
            seti(0, geti(0) ^ 1)
        else:
            raise ValueError(kind)
    else:
        raise ValueError(kind)
    return bytes(buf), bytes(inst)


def with_rejects(plan: Plan, batch: Batch, n_pi: int, fraction: float = 0.01, seed: int = 7, kinds=None) -> Batch:
    """Copy of `batch` where about `fraction` of the proofs are corrupted (expected = 0 there)."""
    rng = random.Random(seed)
    kinds = list(kinds or ["flip_first_scalar"])
    proofs, insts, expected = [], [], list(batch.expected)
    for i in range(batch.n):
        p = batch.proof(i)
        ins = batch.instances[32 * n_pi * i:32 * n_pi * (i + 1)]
        if rng.random() < fraction:
            k = kinds[rng.randrange(len(kinds))]
            res = corrupt(plan, p, ins, k, rng)
            if res is not None:
                p, ins = res
                expected[i] = 0
        proofs.append(p)
        insts.append(ins)
    off = [0]
    for p in proofs:
        off.append(off[-1] + len(p))
    return Batch(n=batch.n, proofs=b"".join(proofs), proof_off=off, instances=b"".join(insts),
                 committed=batch.committed, expected=expected)
