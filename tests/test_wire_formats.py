"""Wire formats of the reference's exported artefacts: proof hex / JSON, public-input lines,
committed-instance x/y, generated VK constant files.  CPU only; the GPU leg is in test_gpu_parity.py."""
import io
import json
import random

import pytest

from plutus_halo2_verifier_gen_amd import bls12_381 as bls
from plutus_halo2_verifier_gen_amd import vk as V
from plutus_halo2_verifier_gen_amd import wire

R = bls.R


def test_proof_hex_and_json_roundtrip(tmp_path, kats):
    proof = bytes.fromhex(kats["simple_mul_full"]["proof"])
    h, j = tmp_path / "serialized_proof.hex", tmp_path / "serialized_proof.json"
    wire.export_proof(str(h), proof)
    wire.serialize_proof(str(j), proof)
    # proof_serialization.rs:12 hex::encode = lowercase, no newline; :28 serde_json of &[u8] = integer array
    assert h.read_text() == proof.hex()
    assert json.loads(j.read_text()) == list(proof) and " " not in j.read_text()
    assert wire.load_proof(str(h)) == wire.load_proof(str(j)) == proof
    with pytest.raises(wire.WireError):
        wire.parse_proof("zz")
    with pytest.raises(wire.WireError):
        wire.parse_proof("[1,2,256]")


def test_public_inputs_are_big_endian_lines():
    vals = [42, 0, R - 1, 1 << 200]
    out = io.StringIO()
    wire.export_public_inputs(vals, out)
    lines = out.getvalue().split("\n")
    assert lines[-1] == "" and len(lines) == 5
    assert lines[0] == "00" * 31 + "2a"                       # to_bytes_be (proof_serialization.rs:45)
    assert wire.parse_public_inputs(out.getvalue()) == vals
    assert wire.instances_to_abi([42])[:2] == b"\x2a\x00"     # the C-ABI side is little-endian
    with pytest.raises(wire.WireError):
        wire.parse_public_inputs("%064x\n" % R)               # non-canonical
    with pytest.raises(wire.WireError):
        wire.parse_public_inputs("2a\n")


def test_committed_instance_xy():
    pt = bls.g1_mul(bls.G1_GEN, 12345)
    out = io.StringIO()
    wire.export_committed_inputs(pt, out)
    text = out.getvalue()
    assert text.count("\n") == 1 and not text.endswith("\n")  # x, newline, y, no trailing newline (:60-68)
    assert wire.parse_committed_inputs(text) == pt
    assert wire.committed_to_abi(pt) == bls.g1_compress(pt)
    none = io.StringIO()
    wire.export_committed_inputs(None, none)
    assert none.getvalue() == "" and wire.parse_committed_inputs("") is None
    assert wire.committed_to_abi((0, 0)) == bls.g1_compress(None)
    with pytest.raises(wire.WireError):
        wire.parse_committed_inputs("%096x\n%096x" % (pt[0], pt[1] ^ 1))


@pytest.mark.parametrize("flavour", ["aiken", "plinth"])
def test_vk_constants_files_roundtrip(flavour):
    vk, _ = V.lookup_table_vk()
    c = vk.constants()
    c.check(vk.k)
    render = wire.render_vk_constants_aiken if flavour == "aiken" else wire.render_vk_constants_plinth
    text = render(c)
    back = wire.parse_vk_constants(text)
    assert back == c
    # a different instantiation of the same circuit shape: constants replace, structure stays
    rng = random.Random(4)
    other = wire.VKConstants(**{**c.__dict__})
    other.fixed_commitments = [bls.g1_compress(bls.g1_mul(bls.G1_GEN, rng.randrange(1, R))).hex() for _ in c.fixed_commitments]
    other.transcript_repr = rng.randrange(R)
    vk2 = vk.with_constants(wire.parse_vk_constants(render(other)))
    assert vk2.fixed_commitments == other.fixed_commitments and vk2.gates == vk.gates
    assert vk2.transcript_repr == other.transcript_repr
    bad = wire.VKConstants(**{**c.__dict__})
    bad.omega = (c.omega * c.omega) % R
    with pytest.raises(wire.WireError):
        vk.with_constants(bad)


def test_vk_constants_golden_line_formats(kats):
    """The reference's own constants (vk_constants.hbs / transcript.ak) in the generated-file syntax."""
    neg_g1 = kats["neg_g1_generator"]
    c = wire.VKConstants(fixed_commitments=[neg_g1.lower()], permutation_commitments=[],
                         s_g2=bls.g2_compress(bls.G2_GEN).hex(), omega=V.domain_omega(4),
                         omega_inv=bls.fr_inv(V.domain_omega(4)), barycentric_weight=bls.fr_inv(16),
                         transcript_repr=7, blinding_factors=5)
    text = wire.render_vk_constants_aiken(c)
    assert 'pub const f1_commitment: ByteArray = #"%s"' % neg_g1.lower() in text     # emitters/aiken.rs:1060-1064
    assert "pub const blinding_factors: Int = 5" in text
    hs = wire.render_vk_constants_plinth(c)
    x, y = bls.g1_neg(bls.G1_GEN)
    assert "(0x%096x, 0x%096x)" % (x, y) in hs                                         # emitters/plinth.rs:923-927
    assert wire.parse_vk_constants(hs) == c


def test_assemble_batch_layout():
    from plutus_halo2_verifier_gen_amd import verify_files
    vk, _ = V.simple_mul_vk()
    proofs = [b"\x01" * 10, b"\x02" * 7]
    buf, off, inst, ci = verify_files.assemble(vk, proofs, [[1, 2, 3]], [])
    assert off == [0, 10, 17] and buf == proofs[0] + proofs[1] and ci is None
    assert inst == wire.instances_to_abi([1, 2, 3]) * 2
    with pytest.raises(wire.WireError):
        verify_files.assemble(vk, proofs, [[1, 2]], [])


def test_recursion_constants_roundtrip():
    """The {{{RECURSION_CONSTANTS}}} block of vk_constants.hbs (emitters/aiken.rs:1089-1134): inner keys of a recursive
    circuit are rendered as transcript_rep_<name> / f<i>_<name> / p<i>_<name> and read back; with_constants swaps them."""
    vk, _ = V.ivc_vk()
    c = vk.constants()
    assert c.recursion_vks and c.recursion_vks[0]["name"] == "inner"
    text = wire.render_vk_constants_aiken(c)
    assert "pub const transcript_rep_inner = 0x%064x" % c.recursion_vks[0]["transcript_repr"] in text
    assert 'pub const f1_inner: ByteArray = #"%s"' % c.recursion_vks[0]["fixed_commitments"][0] in text
    back = wire.parse_vk_constants(text)
    assert back == c
    back.check(vk.k)
    # the key's own constants are not mistaken for an inner key, and a plain file has no recursion block
    plain, _ = V.simple_mul_vk()
    assert wire.parse_vk_constants(wire.render_vk_constants_aiken(plain.constants())).recursion_vks is None
    # another instantiation of the inner key travels through with_constants
    rng = random.Random(8)
    other = wire.parse_vk_constants(text)
    other.recursion_vks[0]["transcript_repr"] = rng.randrange(R)
    vk2 = vk.with_constants(other)
    assert vk2.recursion_vks[0]["transcript_repr"] == other.recursion_vks[0]["transcript_repr"] and vk2.gates == vk.gates
    other.recursion_vks[0]["name"] = "elsewhere"
    with pytest.raises(wire.WireError):
        vk.with_constants(other)


def test_vk_json_schema_and_validator():
    """docs/vk_schema.json ("h2v-vk/1") is the frozen JSON form of VerifyingKey: same fields, every builder's key fits it
    (structurally checked here without a JSON-schema library), and vk.validate rejects what the reference itself would
    panic on or refuse (languages/aiken.rs:134-156, extraction/mod.rs:41-55)."""
    import json
    import os
    schema = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "docs", "vk_schema.json")))
    fields = set(V.VerifyingKey.__dataclass_fields__)
    assert set(schema["properties"]) == fields
    assert set(schema["required"]) <= fields and schema["properties"]["schema_version"]["const"] == V.SCHEMA_VERSION
    tags = {alt["items"][0]["const"] for alt in schema["definitions"]["expr"]["oneOf"]}
    assert tags == set(V._EXPR_ARITY)
    for name, build in V.BUILDERS.items():
        key, _ = build()
        d = json.loads(key.to_json())
        assert set(schema["required"]) <= set(d) <= fields, name
        assert V.VerifyingKey.from_json(key.to_json()) == key
    key, _ = V.lookup_table_vk()
    d = json.loads(key.to_json())

    def broken(**kw):
        return json.dumps({**d, **kw})

    for bad, why in [
        (broken(gates=[["selector", 0]]), "Selector"),
        (broken(gates=[["sum", ["instance", 0], ["const", 1]]]), "Instance"),
        (broken(gates=[["prod", ["challenge", 0], ["advice", 0]]]), "Challenge"),
        (broken(gates=[["advice", 99]]), "out of range"),
        (broken(gates=[["const", R]]), "canonical"),
        (broken(gates=[["pow", ["advice", 0], 2]]), "unknown expression"),
        (broken(n_committed_instances=2), "committed instance"),
        (broken(schema_version=2), "schema_version"),
        (broken(surprise=1), "unknown field"),
        (broken(fixed_commitments=d["fixed_commitments"][:-1]), "fixed commitment"),
        (broken(lookups=[[[["advice", 0]], []]]), "equal length"),
    ]:
        with pytest.raises(V.VKError, match=why):
            V.VerifyingKey.from_json(bad)
    # rotations the emitted Aiken verifier cannot name are an error only in strict mode (this backend handles them)
    far = V.VerifyingKey.from_json(broken(advice_queries=d["advice_queries"] + [[0, 7]]))
    V.validate(far)
    with pytest.raises(V.VKError, match="rotation"):
        V.validate(far, strict_rotations=True)


def _kzg_params_file(k, s, encoding, g1_bytes=96):
    """A parameter file as `ParamsKZG::write_custom` is understood to lay it out (k, 2^k G1 powers, 2^k Lagrange-basis
    G1 elements, g2, s_g2 - src/kzg_params.rs:61-81 calls it; the layout itself is in the un-vendored crate): the G1 part
    is filler here (the verifier never reads it), the two G2 elements are real."""
    R384 = 1 << 384

    def fp_le(v, mont):
        return ((v * R384) % bls.P if mont else v).to_bytes(48, "little")

    def enc(pt):
        (x0, x1), (y0, y1) = pt
        if encoding == "mont":
            return b"".join(fp_le(v, True) for v in (x0, x1, y0, y1))
        if encoding == "canon":
            return b"".join(fp_le(v, False) for v in (x0, x1, y0, y1))
        if encoding == "zcash":
            return b"".join(v.to_bytes(48, "big") for v in (x1, x0, y1, y0))
        if encoding == "compressed":
            return bls.g2_compress(pt)
        if encoding == "proj":
            z = (7, 11)
            X, Y = bls.f2_mul(pt[0], z), bls.f2_mul(pt[1], z)
            return b"".join(fp_le(v, True) for v in (X[0], X[1], Y[0], Y[1], z[0], z[1]))
        raise AssertionError(encoding)

    rng = random.Random(k)
    body = bytes(rng.getrandbits(8) for _ in range(2 * (1 << k) * g1_bytes))
    return k.to_bytes(4, "little") + body + enc(bls.G2_GEN) + enc(bls.g2_mul(bls.G2_GEN, s))


@pytest.mark.parametrize("encoding", ["mont", "canon", "zcash", "compressed", "proj"])
def test_kzg_params_reader_is_self_validating(encoding, tmp_path):
    """src/kzg_params.rs:50-57 reads `kzg_params_{k}` with RawBytesUnchecked; no sample of that layout exists in the
    reference, so the reader accepts a file only when the element before last IS the G2 generator (UNPINNED format)."""
    s = 0x1234567890ABCDEF1234567
    blob = _kzg_params_file(4, s, encoding)
    want = bls.g2_compress(bls.g2_mul(bls.G2_GEN, s)).hex()
    got = wire.parse_kzg_params(blob)
    assert (got.k, got.s_g2) == (4, want)
    assert got.g1_element_bytes == 96
    path = tmp_path / "kzg_params_4"
    path.write_bytes(blob)
    from_file = wire.load_kzg_params(str(path))
    assert (from_file.k, from_file.s_g2, from_file.encoding, from_file.g1_element_bytes) == (4, want, got.encoding, 96)
    # a damaged generator: refused, not mis-read
    size = {"compressed": 96, "proj": 288}.get(encoding, 192)
    bad = bytearray(blob)
    bad[-2 * size + 20] ^= 1
    with pytest.raises(wire.WireError, match="generator"):
        wire.parse_kzg_params(bytes(bad))
    # g2 fine, s_g2 off the curve / out of the subgroup
    bad = bytearray(blob)
    bad[-size + 60] ^= 1
    with pytest.raises(wire.WireError):
        wire.parse_kzg_params(bytes(bad))
    with pytest.raises(wire.WireError, match="short"):
        wire.parse_kzg_params(blob[:100])
    with pytest.raises(wire.WireError, match="implausible"):
        wire.parse_kzg_params(b"\xff\xff\xff\xff" + blob[4:])
