"""Recursion (IVC) accumulator fold: the extra phase the reference's verifier runs between the final MSM and the
pairing when the circuit carries a collapsed KZG accumulator in its public inputs.

Restated from /root/reference/src/plutus_gen/emitters/aiken.rs:648-757 (the RECURSION_ACCUMULATOR block of
aiken-verifier/templates/verification_h2.hbs:123) and docs/algorithms.html "Recursion (IVC)":

    N = number of public inputs i_1..i_N,  F = 1 + #fixed + #perm + sum over inner keys (#fixed + #perm)
    expect transcript_rep == i_1                                   (the inner keys' lines bind, they do not check)
    acc_left  = [i_{N-F-5}] g1_from_coords(1 + i_{N-F-8} B + i_{N-F-9},  1 + i_{N-F-6} B + i_{N-F-7})   (mod p)
    acc_right = [i_{N-F}]   g1_from_coords(1 + i_{N-F-3} B + i_{N-F-4},  1 + i_{N-F-1} B + i_{N-F-2})
    acc_fixed = sum_k [i_{N-F+1+k}] base_k ,  bases = -G1, f1.., p1.., inner keys' f.., p..
    acc_right_final = acc_right + acc_fixed
    c  = LE(blake2b_256(compress(el) || compress(er) || compress(acc_left) || compress(acc_right_final))) mod r
    el <- el + [c] acc_left ;  er <- er + [c] acc_right_final          B = 2^224 = (2^56)^4

g1_from_coords(x, y) (aiken_halo2/lib/bls_utils.ak:32-49) takes only the SIGN of y: it builds the compressed
encoding of x with the parity flag set iff y > p - y and decompresses it (on-curve and subgroup checked by the
builtin; a failure aborts the script = reject).

This module is the big-integer model used by the proof forger and the host tests; the device path is the plan section
and kernels described in DESIGN.md section 10.
"""
from __future__ import annotations

import hashlib
from typing import List, Optional, Tuple

from . import bls12_381 as bls

P, R = bls.P, bls.R
B224 = 1 << 224
SERIALIZED_ACC = 10   # 2 points x 2 coordinates x 2 chunks + 2 scalars (aiken.rs:657)


class Reject(Exception):
    pass


def fixed_bases(vk) -> List[str]:
    """Compressed hex of the accumulator's fixed bases in the emitter's order (aiken.rs:659-694)."""
    out = [bls.g1_compress(bls.g1_neg(bls.G1_GEN)).hex()]
    out += list(vk.fixed_commitments) + list(vk.permutation_commitments)
    for inner in vk.recursion_vks or []:
        out += list(inner["fixed_commitments"]) + list(inner["permutation_commitments"])
    return out


def layout(vk) -> dict:
    """0-based public-input positions of every accumulator field (the emitter's i_k are 1-based)."""
    n, f = vk.n_public_inputs, len(fixed_bases(vk))
    nb_vks = 1 + len(vk.recursion_vks or [])
    if n < nb_vks + f + SERIALIZED_ACC:
        raise ValueError("not enough public inputs to support recursion (aiken.rs:702)")
    i = lambda k: k - 1
    return {
        "F": f, "vk_hash": i(1),
        "left_x": (i(n - f - 8), i(n - f - 9)), "left_y": (i(n - f - 6), i(n - f - 7)), "left_scalar": i(n - f - 5),
        "right_x": (i(n - f - 3), i(n - f - 4)), "right_y": (i(n - f - 1), i(n - f - 2)), "right_scalar": i(n - f),
        "fixed_scalars": [i(n - f + 1 + k) for k in range(f)],
    }


def coord(hi: int, lo: int) -> int:
    return (1 + hi * B224 + lo) % P


def split_coord(value: int) -> Tuple[int, int]:
    """(hi, lo) with coord(hi, lo) == value: what a prover puts into the public inputs."""
    t = (value - 1) % P
    return t >> 224, t & (B224 - 1)


def g1_from_coords(x: int, y: int):
    flags = 0xA0 if y > P - y else 0x80
    raw = bytearray(x.to_bytes(48, "big"))
    raw[0] |= flags
    try:
        return bls.g1_decompress(bytes(raw), True)
    except ValueError as e:
        raise Reject("accumulator point: %s" % e)


def fold(vk, instances: List[int], el, er):
    """-> (el', er') or raises Reject.  Points are affine tuples / None for infinity."""
    lay = layout(vk)
    if instances[lay["vk_hash"]] != vk.transcript_repr:
        raise Reject("verifying-key hash mismatch")
    pt = lambda xs, ys: g1_from_coords(coord(instances[xs[0]], instances[xs[1]]), coord(instances[ys[0]], instances[ys[1]]))
    acc_left = bls.g1_mul(pt(lay["left_x"], lay["left_y"]), instances[lay["left_scalar"]])
    acc_right = bls.g1_mul(pt(lay["right_x"], lay["right_y"]), instances[lay["right_scalar"]])
    acc_fixed = None
    for h, k in zip(fixed_bases(vk), lay["fixed_scalars"]):
        acc_fixed = bls.g1_add(acc_fixed, bls.g1_mul(bls.g1_decompress(bytes.fromhex(h), False), instances[k]))
    acc_right_final = bls.g1_add(acc_right, acc_fixed)
    digest = hashlib.blake2b(b"".join(bls.g1_compress(q) for q in (el, er, acc_left, acc_right_final)), digest_size=32).digest()
    c = int.from_bytes(digest, "little") % R
    return bls.g1_add(el, bls.g1_mul(acc_left, c)), bls.g1_add(er, bls.g1_mul(acc_right_final, c)), c


def make_accumulator(vk, td, rng, instances: List[int]) -> None:
    """Overwrites the accumulator fields of `instances` with a VALID accumulator for the test SRS:
    e(acc_left, s_g2) == e(acc_right_final, G2), i.e. dlog(acc_right_final) = s * dlog(acc_left)."""
    lay = layout(vk)
    instances[lay["vk_hash"]] = vk.transcript_repr
    dlogs = [R - 1] + list(td.fixed_dlogs) + list(td.perm_dlogs) + list(td.rec_dlogs)
    assert len(dlogs) == lay["F"]
    fixed = sum(instances[k] * d for k, d in zip(lay["fixed_scalars"], dlogs)) % R
    a, il, ir = rng.randrange(1, R), rng.randrange(1, R), rng.randrange(1, R)
    b = (td.s * il * a - fixed) * pow(ir, -1, R) % R
    if b == 0:
        b, ir = 1, ir  # (unreachable for random inputs; keeps the point finite)
    for (xs, ys, sk), d, sc in ((("left_x", "left_y", "left_scalar"), a, il), (("right_x", "right_y", "right_scalar"), b, ir)):
        x, y = bls.g1_mul(bls.G1_GEN, d)
        (instances[lay[xs][0]], instances[lay[xs][1]]) = split_coord(x)
        (instances[lay[ys][0]], instances[lay[ys][1]]) = split_coord(y)
        instances[lay[sk]] = sc
