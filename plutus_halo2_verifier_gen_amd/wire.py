"""Wire formats of the artefacts the reference exports next to a proof, so that the
backend consumes the reference's files directly.

Formats (each restated from the reference lines cited; nothing here is on the per-proof hot path):

* proof, hex          one line of lowercase hex, no newline         src/plutus_gen/proof_serialization.rs:11-25
* proof, JSON         serde_json of a byte slice = `[12,255,...]`   proof_serialization.rs:27-41
* public inputs       one 64-digit BIG-endian hex scalar per line   proof_serialization.rs:43-53
* committed instance  affine x, newline, affine y (96 hex digits each, big-endian, no trailing newline);
                      an empty file when the circuit has none      proof_serialization.rs:55-72
* VK constants        the generated `verifier_key.ak` (emitters/aiken.rs:1049-1210 + aiken-verifier/templates/
                      vk_constants.hbs) and `VKConstants.hs` (emitters/plinth.rs:915-1090 +
                      plinth-verifier/templates/vk_constants.hbs): fixed / permutation commitments, s_g2, omega,
                      omega^-1, barycentric weight, transcript representation, blinding factors.

The C-ABI takes scalars as 32-byte little-endian and G1 as 48-byte zcash-compressed (include/h2v.h); the loaders
return those.

* KZG parameters      `kzg_params/kzg_params_{k}` (src/kzg_params.rs:15-18, written with `SerdeFormat::RawBytesUnchecked`,
                      :50-57 / :61-81).  UNPINNED: the byte layout is defined inside the un-vendored `midnight-proofs`
                      crate and the reference holds no sample file.  `read_kzg_params` therefore trusts no layout: it
                      takes the verifier's two elements (g2, s_g2 - `ParamsKZG::verifier_params()`) from the END of the
                      file, tries the encodings a BLS12-381 G2 element can have there, and accepts one only if the
                      element before last decodes to the standard G2 generator and the last one to a point of the G2
                      subgroup.  A file in a layout it does not know is refused, never mis-read.  `s_g2` normally comes
                      from the VK constants files, which carry it compressed; the two are compared when both are given.
"""
from __future__ import annotations

import json
import re
from dataclasses import dataclass, field
from typing import IO, List, Optional, Sequence, Tuple

from . import bls12_381 as bls

R, P = bls.R, bls.P


class WireError(ValueError):
    pass


# ----------------------------------------------------------------------------- proof
def export_proof(path: str, proof: bytes) -> None:
    """proof_serialization.rs:11-25 (`export_proof`)."""
    with open(path, "w") as f:
        f.write(bytes(proof).hex())


def serialize_proof(path: str, proof: bytes) -> None:
    """proof_serialization.rs:27-41 (`serialize_proof`): serde_json of `&[u8]` is an array of integers, no spaces."""
    with open(path, "w") as f:
        f.write(json.dumps(list(bytes(proof)), separators=(",", ":")))


def parse_proof(text: str) -> bytes:
    """Accepts either export: hex text or the JSON byte array."""
    t = text.strip()
    if t.startswith("["):
        try:
            vals = json.loads(t)
        except json.JSONDecodeError as e:
            raise WireError("proof JSON: %s" % e)
        if not all(isinstance(v, int) and 0 <= v <= 255 for v in vals):
            raise WireError("proof JSON: entries must be bytes")
        return bytes(vals)
    try:
        return bytes.fromhex(t)
    except ValueError as e:
        raise WireError("proof hex: %s" % e)


def load_proof(path: str) -> bytes:
    with open(path) as f:
        return parse_proof(f.read())


# ----------------------------------------------------------------------------- public inputs
def export_public_inputs(instance: Sequence[int], out: IO[str]) -> None:
    """proof_serialization.rs:43-53: `to_bytes_be()` hex + newline per scalar."""
    for v in instance:
        if not 0 <= v < R:
            raise WireError("public input out of range")
        out.write("%064x\n" % v)


def parse_public_inputs(text: str) -> List[int]:
    vals = []
    for ln, line in enumerate(text.splitlines(), 1):
        s = line.strip()
        if not s:
            continue
        if len(s) != 64 or not re.fullmatch(r"[0-9a-fA-F]{64}", s):
            raise WireError("public inputs line %d: expected 64 hex digits" % ln)
        v = int(s, 16)
        if v >= R:
            raise WireError("public inputs line %d: not a canonical scalar" % ln)
        vals.append(v)
    return vals


def load_public_inputs(path: str) -> List[int]:
    with open(path) as f:
        return parse_public_inputs(f.read())


def instances_to_abi(values: Sequence[int]) -> bytes:
    """32-byte little-endian each: the layout of h2v_batch.instances (include/h2v.h)."""
    return b"".join(int(v).to_bytes(32, "little") for v in values)


# ----------------------------------------------------------------------------- committed instance
def export_committed_inputs(point: Optional[Tuple[int, int]], out: IO[str]) -> None:
    """proof_serialization.rs:55-72: nothing for None; x, newline, y (no trailing newline) otherwise."""
    if point is None:
        return
    x, y = point
    out.write("%096x\n" % x)
    out.write("%096x" % y)


def parse_committed_inputs(text: str) -> Optional[Tuple[int, int]]:
    lines = [s.strip() for s in text.splitlines() if s.strip()]
    if not lines:
        return None
    if len(lines) != 2 or any(not re.fullmatch(r"[0-9a-fA-F]{96}", s) for s in lines):
        raise WireError("committed instance: expected two lines of 96 hex digits")
    x, y = int(lines[0], 16), int(lines[1], 16)
    if x >= P or y >= P:
        raise WireError("committed instance: coordinate not reduced")
    if (x, y) != (0, 0) and (y * y - x * x * x - 4) % P != 0:
        raise WireError("committed instance: not on the curve")
    return (x, y)


def load_committed_inputs(path: str) -> Optional[Tuple[int, int]]:
    with open(path) as f:
        return parse_committed_inputs(f.read())


def committed_to_abi(point: Optional[Tuple[int, int]]) -> Optional[bytes]:
    """48-byte compressed G1 (h2v_batch.committed).  The exporter writes the identity as (0, 0)."""
    if point is None:
        return None
    return bls.g1_compress(None if point == (0, 0) else point)


# ----------------------------------------------------------------------------- VK constants (generated files)
@dataclass
class VKConstants:
    """`InstantiationSpecificData` as far as the generated constant files carry it
    (extraction/data/.../instantiation_data.rs:26-41)."""
    fixed_commitments: List[str] = field(default_factory=list)        # 48-byte compressed, hex
    permutation_commitments: List[str] = field(default_factory=list)
    s_g2: str = ""                                                    # 96-byte compressed, hex
    omega: int = 0
    omega_inv: int = 0
    barycentric_weight: int = 0
    transcript_repr: int = 0
    blinding_factors: int = 0
    # inner verifying keys of a recursive (IVC) circuit, as the {{{RECURSION_CONSTANTS}}} block of vk_constants.hbs carries
    # them (emitters/aiken.rs:1089-1134): None when the file has no such block, else a list of
    # {name, transcript_repr, fixed_commitments, permutation_commitments} in file order
    recursion_vks: Optional[List[dict]] = None

    def check(self, k: Optional[int] = None) -> None:
        """Internal consistency the reference guarantees by construction."""
        if self.omega * self.omega_inv % R != 1:
            raise WireError("omega * omega_inv != 1")
        inner = [h for v in (self.recursion_vks or []) for h in v["fixed_commitments"] + v["permutation_commitments"]]
        for h in self.fixed_commitments + self.permutation_commitments + inner:
            bls.g1_decompress(bytes.fromhex(h), True)
        bls.g2_decompress(bytes.fromhex(self.s_g2))
        if k is not None:
            if pow(self.omega, 1 << k, R) != 1 or pow(self.omega, 1 << (k - 1), R) == 1:
                raise WireError("omega is not a primitive 2^k-th root of unity")
            if self.barycentric_weight * (1 << k) % R != 1:
                raise WireError("barycentric weight != 1/n")


def _compress_xy(x: int, y: int) -> str:
    return bls.g1_compress(None if (x, y) == (0, 0) else (x, y)).hex()


def render_vk_constants_aiken(c: VKConstants) -> str:
    """The constant definitions of a generated `verifier_key.ak` (line formats of emitters/aiken.rs:1049-1160 and the
    template's scalar lines).  Only the definitions: imports and the budget-check tests carry no data."""
    o = []
    for i, h in enumerate(c.permutation_commitments):
        o.append('pub const p%d_commitment: ByteArray = #"%s"' % (i + 1, h))
    o.append("")
    for i, h in enumerate(c.fixed_commitments):
        o.append('pub const f%d_commitment: ByteArray = #"%s"' % (i + 1, h))
    o.append("")
    o.append('pub const g2_const: G2Element = decompress_g2( #"%s" )' % c.s_g2)
    o.append("pub const omega: State<Scalar> = from_int( 0x%064x )" % c.omega)
    o.append("pub const omega_inv: State<Scalar> = from_int( 0x%064x )" % c.omega_inv)
    o.append("pub const barycentric_weight: State<Scalar> = from_int( 0x%064x )" % c.barycentric_weight)
    o.append("pub const transcript_rep: State<Scalar> = from_int( 0x%064x )" % c.transcript_repr)
    o.append("pub const blinding_factors: Int = %d" % c.blinding_factors)
    for v in c.recursion_vks or []:      # emitters/aiken.rs:1089-1134: transcript_rep_<name>, f<i>_<name>, p<i>_<name>
        o.append("")
        o.append("pub const transcript_rep_%s = 0x%064x" % (v["name"], v["transcript_repr"]))
        for i, h in enumerate(v["fixed_commitments"]):
            o.append('pub const f%d_%s: ByteArray = #"%s"' % (i + 1, v["name"], h))
        for i, h in enumerate(v["permutation_commitments"]):
            o.append('pub const p%d_%s: ByteArray = #"%s"' % (i + 1, v["name"], h))
    return "\n".join(o) + "\n"


def parse_vk_constants_aiken(text: str) -> VKConstants:
    c = VKConstants()

    def numbered(prefix):
        found = {int(m.group(1)): m.group(2).lower()
                 for m in re.finditer(r'pub const %s(\d+)_commitment\s*:\s*ByteArray\s*=\s*#"([0-9a-fA-F]{96})"' % prefix, text)}
        if sorted(found) != list(range(1, len(found) + 1)):
            raise WireError("%s commitments are not numbered 1..n" % prefix)
        return [found[i] for i in range(1, len(found) + 1)]

    c.fixed_commitments = numbered("f")
    c.permutation_commitments = numbered("p")
    m = re.search(r'g2_const\s*:\s*G2Element\s*=\s*decompress_g2\(\s*#"([0-9a-fA-F]{192})"\s*\)', text)
    if not m:
        raise WireError("g2_const not found")
    c.s_g2 = m.group(1).lower()

    def scalar(name):
        m = re.search(r"pub const %s\s*:\s*State<Scalar>\s*=\s*from_int\(\s*0x([0-9a-fA-F]+)\s*\)" % name, text)
        if not m:
            raise WireError("%s not found" % name)
        return int(m.group(1), 16)

    c.omega, c.omega_inv = scalar("omega"), scalar("omega_inv")
    c.barycentric_weight, c.transcript_repr = scalar("barycentric_weight"), scalar("transcript_rep")
    m = re.search(r"pub const blinding_factors\s*:\s*Int\s*=\s*(\d+)", text)
    if not m:
        raise WireError("blinding_factors not found")
    c.blinding_factors = int(m.group(1))
    c.recursion_vks = _parse_recursion_constants_aiken(text)
    return c


def _parse_recursion_constants_aiken(text: str) -> Optional[List[dict]]:
    """The {{{RECURSION_CONSTANTS}}} block (emitters/aiken.rs:1089-1134): per inner key `pub const transcript_rep_<name> =
    <Scalar Debug>` followed by `pub const f<i>_<name>: ByteArray = #"<48 bytes>"` and `p<i>_<name>` lines.  The scalar is
    printed with Rust's `{:?}`, which for the field type is `0x` + 64 hex digits; a decimal integer is accepted too.
    The key's own constants use the fixed suffix `_commitment` and the name `transcript_rep`: not matched here."""
    names = []
    for m in re.finditer(r"pub const transcript_rep_(\w+)\s*=\s*(0x[0-9a-fA-F]+|\d+)", text):
        names.append((m.group(1), int(m.group(2), 0)))
    if not names:
        return None
    out = []
    for name, rep in names:
        if name == "commitment":
            raise WireError("an inner verifying key cannot be called 'commitment'")

        def numbered(prefix, name=name):
            found = {int(m.group(1)): m.group(2).lower()
                     for m in re.finditer(r'pub const %s(\d+)_%s\s*:\s*ByteArray\s*=\s*#"([0-9a-fA-F]{96})"' % (prefix, re.escape(name)), text)}
            if sorted(found) != list(range(1, len(found) + 1)):
                raise WireError("%s commitments of inner key %s are not numbered 1..n" % (prefix, name))
            return [found[i] for i in range(1, len(found) + 1)]

        if rep >= R:
            raise WireError("transcript_rep_%s is not a canonical scalar" % name)
        out.append({"name": name, "transcript_repr": rep, "fixed_commitments": numbered("f"), "permutation_commitments": numbered("p")})
    return out


def render_vk_constants_plinth(c: VKConstants) -> str:
    """The data-carrying definitions of a generated `VKConstants.hs` (emitters/plinth.rs:917-975, 1046-1090)."""
    def pairs(hexes):
        rows = []
        for h in hexes:
            pt = bls.g1_decompress(bytes.fromhex(h), False)
            x, y = (0, 0) if pt is None else pt
            rows.append("    (0x%096x, 0x%096x)" % (x, y))
        return ",\n".join(rows)

    o = ["f_commitments_val_pairs :: [(Integer, Integer)]", "f_commitments_val_pairs =", "  [",
         pairs(c.fixed_commitments), "  ]", "",
         "p_commitment_val_pairs :: [(Integer, Integer)]", "p_commitment_val_pairs =", "  [",
         pairs(c.permutation_commitments), "  ]", "",
         "s_g2_val_bbs :: BuiltinByteString", "s_g2_val_bbs =", "  stringToBuiltinByteStringHex", '    "%s"' % c.s_g2, ""]
    for name, v in (("omega_val", c.omega), ("omegaInv_val", c.omega_inv),
                    ("barycentricWeight_val", c.barycentric_weight), ("transcriptRepr", c.transcript_repr)):
        o += ["%s :: Scalar" % name, "%s =" % name, "  mkScalar", "    (0x%064x `modulo` bls12_381_field_prime)" % v, ""]
    o += ["blinding_factors :: Integer", "blinding_factors = %d" % c.blinding_factors]
    return "\n".join(o) + "\n"


def parse_vk_constants_plinth(text: str) -> VKConstants:
    c = VKConstants()

    def pair_list(name):
        m = re.search(r"^%s\s*=\s*\[(.*?)\]" % name, text, re.S | re.M)
        if not m:
            raise WireError("%s not found" % name)
        out = []
        for x, y in re.findall(r"\(\s*0x([0-9a-fA-F]+)\s*,\s*0x([0-9a-fA-F]+)\s*\)", m.group(1)):
            xi, yi = int(x, 16), int(y, 16)
            if xi >= P or yi >= P:
                raise WireError("%s: coordinate not reduced" % name)
            if (xi, yi) != (0, 0) and (yi * yi - xi * xi * xi - 4) % P != 0:
                raise WireError("%s: point not on the curve" % name)
            out.append(_compress_xy(xi, yi))
        return out

    c.fixed_commitments = pair_list("f_commitments_val_pairs")
    c.permutation_commitments = pair_list("p_commitment_val_pairs")
    m = re.search(r'stringToBuiltinByteStringHex\s*"([0-9a-fA-F]{192})"', text)
    if not m:
        raise WireError("s_g2 not found")
    c.s_g2 = m.group(1).lower()

    def scalar(name):
        m = re.search(r"^%s\s*=\s*mkScalar\s*\(\s*0x([0-9a-fA-F]+)\s*`modulo`" % name, text, re.S | re.M)
        if not m:
            raise WireError("%s not found" % name)
        return int(m.group(1), 16) % R

    c.omega, c.omega_inv = scalar("omega_val"), scalar("omegaInv_val")
    c.barycentric_weight, c.transcript_repr = scalar("barycentricWeight_val"), scalar("transcriptRepr")
    m = re.search(r"^blinding_factors\s*=\s*(\d+)", text, re.M)
    if not m:
        raise WireError("blinding_factors not found")
    c.blinding_factors = int(m.group(1))
    return c


def parse_vk_constants(text: str) -> VKConstants:
    return parse_vk_constants_aiken(text) if "pub const" in text else parse_vk_constants_plinth(text)


def load_vk_constants(path: str) -> VKConstants:
    with open(path) as f:
        return parse_vk_constants(f.read())


# ----------------------------------------------------------------------------- KZG parameters (verifier part)
@dataclass
class KZGVerifierParams:
    """What `ParamsKZG::verifier_params()` hands to `Guard::verify` (examples/simple_mul.rs:101-104): s_g2 (and g2)."""
    k: int                      # the u32 the file starts with (log2 of the number of G1 powers)
    s_g2: str                   # hex of the 96-byte compressed point - the form vk.py / the VK constants use
    encoding: str               # which G2 encoding the file's tail was in
    g1_element_bytes: Optional[int]  # (file length - 4 - the two G2 elements) / (2 * 2^k) if that is a whole number, else None


_R384 = 1 << 384


def _g2_from_fp_quad(vals):
    x, y = (vals[0], vals[1]), (vals[2], vals[3])
    return (x, y)


def _g2_tail_decoders():
    """(name, element size, bytes -> affine point or None).  Every decoder raises ValueError on a malformed element."""
    def fp_le(b, mont):
        v = int.from_bytes(b, "little")
        if v >= P:
            raise ValueError("coordinate not below p")
        return v * pow(_R384, -1, P) % P if mont else v

    def raw_affine(mont):
        def dec(b):
            return _g2_from_fp_quad([fp_le(b[48 * i:48 * i + 48], mont) for i in range(4)])     # x.c0 x.c1 y.c0 y.c1
        return dec

    def raw_projective(mont):
        def dec(b):
            c = [fp_le(b[48 * i:48 * i + 48], mont) for i in range(6)]                          # X Y Z, each c0 c1
            z = (c[4], c[5])
            if z == (0, 0):
                return None
            zi = bls.f2_inv(z)
            return (bls.f2_mul((c[0], c[1]), zi), bls.f2_mul((c[2], c[3]), zi))
        return dec

    def zcash_uncompressed(b):
        if b[0] & 0xE0:
            raise ValueError("flag bits set in an uncompressed element")
        v = [int.from_bytes(b[48 * i:48 * i + 48], "big") for i in range(4)]                   # x.c1 x.c0 y.c1 y.c0
        if any(t >= P for t in v):
            raise ValueError("coordinate not below p")
        return ((v[1], v[0]), (v[3], v[2]))

    return [("raw affine, Montgomery limbs, little-endian", 192, raw_affine(True)),
            ("raw affine, canonical, little-endian", 192, raw_affine(False)),
            ("zcash uncompressed", 192, zcash_uncompressed),
            ("zcash compressed", 96, lambda b: bls.g2_decompress(bytes(b), check_subgroup=False)),
            ("raw projective, Montgomery limbs, little-endian", 288, raw_projective(True)),
            ("raw projective, canonical, little-endian", 288, raw_projective(False))]


def parse_kzg_params(blob: bytes) -> KZGVerifierParams:
    """The verifier's part of a KZG parameter file (see the module docstring: self-validating, layout unpinned)."""
    blob = bytes(blob)
    if len(blob) < 4 + 2 * 96:
        raise WireError("KZG params: file too short")
    k = int.from_bytes(blob[:4], "little")
    if not 1 <= k <= 30:
        raise WireError("KZG params: implausible k = %d in the header" % k)
    for name, size, dec in _g2_tail_decoders():
        if len(blob) < 4 + 2 * size:
            continue
        try:
            g2 = dec(blob[-2 * size:-size])
            if g2 != bls.G2_GEN:
                continue
            s_g2 = dec(blob[-size:])
        except ValueError:
            continue
        if s_g2 is None or not bls.g2_is_on_curve(s_g2) or not bls.g2_in_subgroup(s_g2):
            raise WireError("KZG params (%s): g2 is the generator but s_g2 is not a point of the G2 subgroup" % name)
        body = len(blob) - 4 - 2 * size
        per = body // (2 << k) if body % (2 << k) == 0 else None
        return KZGVerifierParams(k=k, s_g2=bls.g2_compress(s_g2).hex(), encoding=name, g1_element_bytes=per)
    raise WireError("KZG params: the element before last is not the G2 generator in any known encoding "
                    "(layout of this file is not one this reader knows; nothing was assumed)")


def load_kzg_params(path: str) -> KZGVerifierParams:
    """Reads only the header and the tail (the G1 powers - 2 x 2^k elements - are the prover's)."""
    import os
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        head = f.read(4)
        tail_len = min(size - 4, 2 * 288)
        f.seek(size - tail_len)
        tail = f.read(tail_len)
    # the length-derived field needs the true file length: pad the middle virtually
    params = parse_kzg_params(head + tail)
    for _name, sz, _d in _g2_tail_decoders():
        if _name == params.encoding:
            body = size - 4 - 2 * sz
            params.g1_element_bytes = body // (2 << params.k) if body >= 0 and body % (2 << params.k) == 0 else None
    return params
