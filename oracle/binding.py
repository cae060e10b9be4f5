"""ORACLE (test infrastructure) - ctypes binding of oracle/_build/libh2v_oracle.so and the vk-description
serializer the C oracle parses.  Importable ONLY from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing in the product package imports this module.

The serializer takes the plain dict form of a VerifyingKey (json.loads(vk.to_json())) so that this file has no
dependency on the product package either.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libh2v_oracle.so")

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ORC_MAX_EXPR = 256
# compressed -G1 (aiken-verifier/templates/vk_constants.hbs:18)
NEG_G1_COMPRESSED = bytes.fromhex("b7f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
STATUS = {0: "accept", 1: "pairing", 2: "point", 3: "scalar", 4: "short", 5: "inverse", 6: "recursion"}


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    # make is a no-op when the library is newer than its sources
    subprocess.check_call(["make", "-s", "-C", HERE], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return LIB_PATH


class Trace(C.Structure):
    _fields_ = (
        [("status", C.c_int32), ("n_expressions", C.c_uint32)]
        + [(n, C.c_uint8 * 32) for n in (
            "theta", "beta", "gamma", "trash", "y", "x", "x1", "x2", "x3", "x4", "x_prev", "x_next", "x_last", "xn",
            "l_last", "l_0", "active_rows", "h_eval", "vanishing_s", "f_eval", "v")]
        + [(n, C.c_uint8 * 96) for n in ("vanishing_g", "el", "er")]
        + [("expressions", (C.c_uint8 * 32) * ORC_MAX_EXPR)]
    )

    def scalar(self, name) -> int:
        return int.from_bytes(bytes(getattr(self, name)), "little")

    def point(self, name):
        b = bytes(getattr(self, name))
        if b == bytes(96):
            return None
        return (int.from_bytes(b[:48], "big"), int.from_bytes(b[48:], "big"))

    def expression(self, i) -> int:
        return int.from_bytes(bytes(self.expressions[i]), "little")


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_vk_parse.restype = C.c_void_p
        L.orc_vk_parse.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_vk_free.argtypes = [C.c_void_p]
        L.orc_vk_num_point_sets.argtypes = [C.c_void_p]
        L.orc_vk_num_msm_terms.argtypes = [C.c_void_p]
        L.orc_vk_proof_len.argtypes = [C.c_void_p]
        L.orc_vk_proof_len.restype = C.c_size_t
        L.orc_verify.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.POINTER(Trace)]
        L.orc_verify_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_void_p, C.c_char_p, C.c_char_p,
                                       C.c_void_p, C.c_int]
        L.orc_vk_commitment_map.restype = C.c_long
        L.orc_vk_commitment_map.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_lookup_argument.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t, C.c_char_p,
                                          C.c_size_t, C.c_char_p, C.c_void_p]
        L.orc_transcript_script.restype = C.c_long
        L.orc_transcript_script.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_void_p,
                                            C.c_size_t]
        _lib = L
    return _lib


# ----------------------------------------------------------------------------- vk description blob
_TAGS = {"const": 0, "fixed": 1, "advice": 2, "neg": 3, "sum": 4, "prod": 5, "scaled": 6}


def _fr(x: int) -> bytes:
    return (x % R).to_bytes(32, "little")


def expr_blob(e) -> bytes:
    t = e[0]
    if t == "const":
        return bytes([0]) + _fr(e[1])
    if t in ("fixed", "advice"):
        return bytes([_TAGS[t]]) + struct.pack("<I", e[1])
    if t == "neg":
        return bytes([3]) + expr_blob(e[1])
    if t in ("sum", "prod"):
        return bytes([_TAGS[t]]) + expr_blob(e[1]) + expr_blob(e[2])
    if t == "scaled":
        return bytes([6]) + expr_blob(e[1]) + _fr(e[2])
    raise ValueError(t)


def vk_desc(vk: dict, omega: int, omega_inv: int, barycentric_weight: int) -> bytes:
    """vk: dict form of VerifyingKey; omega / omega^-1 / n^-1 are VK data in the reference
    (instantiation_data.rs:84-90) and are therefore passed in, not derived here."""
    out = bytearray(b"ORCVK001")
    out += struct.pack("<III", vk["k"], vk["blinding_factors"], vk["cs_degree"])
    out += _fr(vk["transcript_repr"]) + _fr(omega) + _fr(omega_inv) + _fr(barycentric_weight)
    out += struct.pack("<II", vk["num_advice_columns"], vk["num_fixed_columns"])
    for key in ("advice_queries", "fixed_queries", "instance_queries"):
        out += struct.pack("<I", len(vk[key]))
        for col, rot in vk[key]:
            out += struct.pack("<ii", col, rot)

    def exprs(lst):
        b = struct.pack("<I", len(lst))
        for e in lst:
            b += expr_blob(e)
        return b

    out += exprs(vk["gates"])
    out += struct.pack("<I", len(vk["lookups"]))
    for ins, tabs in vk["lookups"]:
        out += exprs(ins) + exprs(tabs)
    out += struct.pack("<I", len(vk["trashcans"]))
    for sel, cons in vk["trashcans"]:
        out += expr_blob(sel) + exprs(cons)
    out += struct.pack("<I", len(vk["permutation_columns"]))
    for ty, idx in vk["permutation_columns"]:
        out += bytes([{"advice": 0, "fixed": 1, "instance": 2}[ty]]) + struct.pack("<I", idx)
    for key in ("fixed_commitments", "permutation_commitments"):
        out += struct.pack("<I", len(vk[key]))
        for h in vk[key]:
            out += bytes.fromhex(h)
    out += bytes.fromhex(vk["s_g2"])
    out += struct.pack("<II", vk["n_public_inputs"], vk["n_committed_instances"])
    if vk.get("recursion_vks") is not None:
        # fixed bases of the accumulator in the emitter's order (emitters/aiken.rs:659-694): -G1, f.., p.., inner keys
        bases = [NEG_G1_COMPRESSED] + [bytes.fromhex(h) for h in vk["fixed_commitments"] + vk["permutation_commitments"]]
        for inner in vk["recursion_vks"]:
            bases += [bytes.fromhex(h) for h in inner["fixed_commitments"] + inner["permutation_commitments"]]
        out += struct.pack("<II", 1, len(bases)) + b"".join(bases)
    phases, chal = vk.get("advice_column_phase"), vk.get("challenge_phase")
    if phases is not None or chal:
        if vk.get("recursion_vks") is None:
            out += struct.pack("<II", 0, 0)      # (the phase section follows the recursion section)
        phases = phases if phases is not None else [0] * vk["num_advice_columns"]
        out += struct.pack("<I", len(phases)) + bytes(phases)
        out += struct.pack("<I", len(chal or [])) + bytes(chal or [])
    return bytes(out)


class OracleVK:
    def __init__(self, desc: bytes):
        self._h = lib().orc_vk_parse(desc, len(desc))
        if not self._h:
            raise ValueError("oracle: vk description rejected")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_vk_free(self._h)
            self._h = None

    @property
    def n_point_sets(self) -> int:
        return lib().orc_vk_num_point_sets(self._h)

    @property
    def n_msm_terms(self) -> int:
        return lib().orc_vk_num_msm_terms(self._h)

    @property
    def proof_len(self) -> int:
        return lib().orc_vk_proof_len(self._h)

    def verify(self, proof: bytes, instances, committed: bytes | None = None, trace: bool = False):
        # the 32 bytes as the caller would hand them over: a value in [r, 2^256) stays non-canonical (and is rejected)
        inst = b"".join(int(v).to_bytes(32, "little") if 0 <= int(v) < (1 << 256) else _fr(v) for v in instances)
        tr = Trace() if trace else None
        ok = lib().orc_verify(self._h, proof, len(proof), inst, committed, C.byref(tr) if trace else None)
        return (bool(ok), tr) if trace else bool(ok)

    CK_NAMES = ["instance", "advice", "fixed", "perm", "lookup", "perm_input", "perm_table", "common", "vanishing_g",
                "vanishing_rand", "trash"]
    EK_NAMES = ["instance", "advice", "fixed", "perm", "lookup", "lookup_next", "perm_input", "perm_input_inv",
                "perm_table", "common", "vanishing_s", "random", "trash"]
    ROT_NAMES = ["last", "prev", "cur", "next", "custom"]

    def commitment_map(self):
        """The commitment map build_sets derived (the structure orc_verify walks): a list of dicts
        {commitment: (kind, idx), set: first-seen point-set index, sorted_set: its position after the cardinality
        sort, pairs: [(rotation name, (eval kind, idx, sub))...]}."""
        buf = (C.c_int32 * 65536)()
        n = lib().orc_vk_commitment_map(self._h, buf, 65536)
        assert n >= 0
        out, w = [], 0
        while w < n:
            ck, cidx, st, srt, npts = buf[w:w + 5]
            w += 5
            pairs = []
            for _ in range(npts):
                rk, rn, ek, eidx, esub = buf[w:w + 5]
                w += 5
                rot = self.ROT_NAMES[rk] if rk < 4 else ("custom", rn)
                pairs.append((rot, (self.EK_NAMES[ek], eidx, esub)))
            out.append({"commitment": (self.CK_NAMES[ck], cidx), "set": st, "sorted_set": srt, "pairs": pairs})
        return out

    def verify_batch(self, proofs: bytes, offsets, instances: bytes, committed: bytes | None, threads: int = 1):
        n = len(offsets) - 1
        off = (C.c_uint64 * (n + 1))(*offsets)
        acc = (C.c_uint8 * n)()
        lib().orc_verify_batch(self._h, n, proofs, off, instances, committed, acc, threads)
        return bytes(acc)


# ----------------------------------------------------------------------------- primitive wrappers (golden tests)
def blake2b256(data: bytes) -> bytes:
    out = C.create_string_buffer(32)
    lib().orc_blake2b256(data, len(data), out)
    return out.raw


def transcript_script(proof: bytes, script):
    """script: list of ("common_scalar", int) | ("common_point", bytes48) | ("read_scalar",) | ("read_point",) |
    ("squeeze",).  Returns the list of read / squeezed values (ints for scalars, bytes for points)."""
    ops = bytearray()
    args = bytearray()
    kinds = []
    for st in script:
        if st[0] == "common_scalar":
            ops.append(0); args += _fr(st[1])
        elif st[0] == "common_point":
            ops.append(1); args += st[1]
        elif st[0] == "read_scalar":
            ops.append(2); kinds.append(32)
        elif st[0] == "read_point":
            ops.append(3); kinds.append(48)
        elif st[0] == "squeeze":
            ops.append(4); kinds.append(32)
        else:
            raise ValueError(st)
    cap = sum(kinds) + 1
    out = C.create_string_buffer(cap)
    n = lib().orc_transcript_script(proof, len(proof), bytes(ops), len(ops), bytes(args), out, cap)
    if n < 0:
        raise ValueError("transcript script failed: %d" % n)
    res, pos = [], 0
    for k in kinds:
        chunk = out.raw[pos:pos + k]
        res.append(int.from_bytes(chunk, "little") if k == 32 else chunk)
        pos += k
    return res


def fr_inv(a: int):
    out = C.create_string_buffer(32)
    if not lib().orc_fr_inv(_fr(a), out):
        return None
    return int.from_bytes(out.raw, "little")


def rotate_omegas(omega: int, omega_inv: int, lo: int, hi: int):
    n = hi - lo + 1
    out = C.create_string_buffer(32 * n)
    lib().orc_rotate_omegas(_fr(omega), _fr(omega_inv), lo, hi, out)
    return [int.from_bytes(out.raw[32 * i:32 * i + 32], "little") for i in range(n)]


def lagrange_basis(x, xn, w, rotations):
    n = len(rotations)
    out = C.create_string_buffer(32 * n)
    ok = lib().orc_lagrange_basis(_fr(x), _fr(xn), _fr(w), b"".join(_fr(r) for r in rotations), n, out)
    if not ok:
        return None
    return [int.from_bytes(out.raw[32 * i:32 * i + 32], "little") for i in range(n)]


def lagrange_evaluation(points, x):
    n = len(points)
    out = C.create_string_buffer(32)
    ok = lib().orc_lagrange_evaluation(b"".join(_fr(p[0]) for p in points), b"".join(_fr(p[1]) for p in points),
                                       n, _fr(x), out)
    return int.from_bytes(out.raw, "little") if ok else None


def multiopen_scalars(point_sets, commitments_evals, x1, x2, x3, x4, q_evals):
    """point_sets: list of lists of points; commitments_evals[s] = list (per commitment) of eval lists."""
    S = len(point_sets)
    sizes = (C.c_uint32 * S)(*[len(p) for p in point_sets])
    ncom = (C.c_uint32 * S)(*[len(c) for c in commitments_evals])
    pts = b"".join(_fr(p) for ps in point_sets for p in ps)
    evs = b"".join(_fr(e) for cs in commitments_evals for es in cs for e in es)
    tot = sum(len(p) for p in point_sets)
    qs = C.create_string_buffer(32 * tot)
    fe = C.create_string_buffer(32)
    v = C.create_string_buffer(32)
    ok = lib().orc_multiopen_scalars(S, sizes, pts, ncom, evs, _fr(x1), _fr(x2), _fr(x3), _fr(x4),
                                     b"".join(_fr(q) for q in q_evals), qs, fe, v)
    if not ok:
        return None
    flat = [int.from_bytes(qs.raw[32 * i:32 * i + 32], "little") for i in range(tot)]
    sets, pos = [], 0
    for p in point_sets:
        sets.append(flat[pos:pos + len(p)])
        pos += len(p)
    return sets, int.from_bytes(fe.raw, "little"), int.from_bytes(v.raw, "little")


def _xy(pt) -> bytes:
    return bytes(96) if pt is None else pt[0].to_bytes(48, "big") + pt[1].to_bytes(48, "big")


def _unxy(b: bytes):
    return None if b == bytes(96) else (int.from_bytes(b[:48], "big"), int.from_bytes(b[48:], "big"))


def g1_decompress(b: bytes):
    """returns (ok, point)"""
    out = C.create_string_buffer(96)
    ok = lib().orc_g1_decompress(b, out)
    return (bool(ok), _unxy(out.raw) if ok else None)


def g1_compress(pt) -> bytes:
    out = C.create_string_buffer(48)
    lib().orc_g1_compress(_xy(pt), out)
    return out.raw


def g1_in_subgroup(pt, naive=False) -> int:
    return lib().orc_g1_in_subgroup(_xy(pt), 1 if naive else 0)


def g1_msm(scalars, points):
    out = C.create_string_buffer(96)
    lib().orc_g1_msm(len(scalars), b"".join(_fr(s) for s in scalars), b"".join(_xy(p) for p in points), out)
    return _unxy(out.raw)


def pairing_check(p1, q1c: bytes, p2, q2c: bytes) -> int:
    return lib().orc_pairing_check(_xy(p1), q1c, _xy(p2), q2c)


def g2_generator_compressed() -> bytes:
    out = C.create_string_buffer(96)
    lib().orc_g2_generator_compressed(out)
    return out.raw


def lookup_argument(inputs, table, advice, fixed, theta, beta, gamma, l_0, l_last, active_rows, evals5):
    """The lookup block orc_verify runs, on explicit values: returns the five identities."""
    blob = b"".join(expr_blob(e) for e in list(inputs) + list(table))
    scal = b"".join(_fr(v) for v in [theta, beta, gamma, l_0, l_last, active_rows] + list(evals5))
    out = C.create_string_buffer(160)
    ok = lib().orc_lookup_argument(blob, len(blob), len(inputs), len(table), b"".join(_fr(a) for a in advice), len(advice),
                                   b"".join(_fr(f) for f in fixed), len(fixed), scal, out)
    return [int.from_bytes(out.raw[32 * i:32 * i + 32], "little") for i in range(5)] if ok else None


def eval_expr(e, advice, fixed):
    blob = expr_blob(e)
    out = C.create_string_buffer(32)
    ok = lib().orc_eval_expr(blob, len(blob), b"".join(_fr(a) for a in advice), len(advice),
                             b"".join(_fr(f) for f in fixed), len(fixed), out)
    return int.from_bytes(out.raw, "little") if ok else None
