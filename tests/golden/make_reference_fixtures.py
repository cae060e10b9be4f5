#!/usr/bin/env python3
"""Extracts the reference's own known-answer vectors for the verification hot path into
tests/golden/reference_kats.json (DATA ONLY: hex inputs and expected outputs; no reference source text).

Run in the build container (needs /root/reference, which does not travel to the GPU box):
    python tests/golden/make_reference_fixtures.py

Sources:
  aiken-verifier/aiken_halo2/lib/transcript.ak:108-382      transcript / (de)serialisation KATs + full simple_mul proof
  aiken-verifier/aiken_halo2/lib/lagrange.ak:133-187        Lagrange basis KAT
  aiken-verifier/aiken_halo2/lib/omega_rotations.ak:48-81   omega rotations KAT
  plinth-verifier/plutus-halo2/test/Lagrange.hs:33-84       interpolation KATs
  plinth-verifier/plutus-halo2/test/ProofData.hs:33-215 -> test/Halo2MultiOpenMSM.hs:25-43   multi-open scalars
  aiken-verifier/templates/gates_test.hbs:9-79              lookup-identity KAT (inputs -> expected)
  aiken-verifier/templates/verification_h2.hbs:24, transcript.ak:99, vk_constants.hbs:18   constants
  aiken-verifier/aiken_halo2/lib/bls_utils.ak:119-128       g1_from_coords(x, y') == generator (sign-of-y semantics)
  docs/chip_profiles.json                                   per-chip shape numbers (columns, copy constraints, lookups,
                                                            gate op counts, proof / VK commitments, evals, commitment-map
                                                            set sizes, proof bytes) that BASELINE configs[3] / [4] are built from
"""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")


def read(p):
    with open(os.path.join(REF, p)) as f:
        return f.read()


def test_block(src, name):
    m = re.search(r"test %s\(\)[^{]*\{(.*?)\n\}" % re.escape(name), src, re.S)
    return m.group(1)


def hexes(block):
    return [h.lower() for h in re.findall(r"0x([0-9a-fA-F]+)", block)]


def bytestrings(block):
    return [h.lower() for h in re.findall(r'#"([0-9a-fA-F]*)"', block)]


def main():
    kats = {}
    tr = read("aiken-verifier/aiken_halo2/lib/transcript.ak")
    b = test_block(tr, "transcript_representation_test")
    kats["transcript_repr_first_challenge"] = {"repr": hexes(b)[0], "challenge": hexes(b)[1]}
    kats["point_generator"] = bytestrings(test_block(tr, "point_deserialization_generator"))[0]
    kats["point_neg_generator"] = bytestrings(test_block(tr, "point_deserialization_negated_generator"))[0]
    kats["point_42g"] = bytestrings(test_block(tr, "point_deserialization"))[0]
    kats["scalar_field_prime_bytes"] = bytestrings(test_block(tr, "scalar_deserialization_field_prime"))[0]
    b = test_block(tr, "overflow_scalar_deserialization")
    kats["scalar_other"] = {"bytes": bytestrings(b)[0], "value": hexes(b)[0]}
    kats["absorb_scalar_42_challenge"] = hexes(test_block(tr, "adding_scalar_to_transcript"))[0]
    kats["absorb_point_42g_challenge"] = hexes(test_block(tr, "adding_g1_to_transcript"))[0]
    b = test_block(tr, "squeeze_challenge_calculations")
    kats["mixed"] = {"proof": bytestrings(b)[0], "scalar": hexes(b)[0], "challenge": hexes(b)[1]}
    b = test_block(tr, "full_proof_deserialization_for_simple_mul_circuit")
    hs = hexes(b)
    kats["simple_mul_full"] = {
        "transcript_repr": hs[0], "proof": bytestrings(b)[0],
        "gamma": hs[1], "y": hs[2], "x": hs[3], "advice_eval_1": hs[4], "advice_eval_2": hs[5], "advice_eval_3": hs[6],
        "x1": hs[7], "x2": hs[8], "x3": hs[9], "x4": hs[10], "pi": bytestrings(b)[1],
    }
    m = re.search(r"0x([0-9a-f]+),\n\s*\),\n\s*\)\n\s*\(\n\s*squeeze_scalar", tr)
    kats["const_R_2_256"] = m.group(1)

    lg = read("aiken-verifier/aiken_halo2/lib/lagrange.ak")
    hs = hexes(test_block(lg, "calculate_lagrange_basis"))
    # x, xn, w, 6 rotations (+ literal from_int(1)), then 7 results
    kats["lagrange_basis"] = {"x": hs[0], "xn": hs[1], "barycentric_weight": hs[2], "rotations": hs[3:9] + ["1"],
                              "result": hs[9:16]}
    om = read("aiken-verifier/aiken_halo2/lib/omega_rotations.ak")
    hs = hexes(test_block(om, "calculate_rotations"))
    kats["rotations"] = {"omega": hs[0], "omega_inv": hs[1], "from": -6, "to": 0, "result": hs[2:8] + ["1"]}

    lh = read("plinth-verifier/plutus-halo2/test/Lagrange.hs")
    kats["interpolation"] = [
        {"points": [[1, 1]], "x": 2, "expected": "1"},
        {"points": [[1, 1], [2, 2]], "x": 4, "expected": "4"},
    ]
    m = re.search(r"biggerNumber = do(.*?)nonLinearCase ::", lh, re.S).group(1)
    nums = [int(n) for n in re.findall(r"mkScalar (\d+)\b", m)]
    kats["interpolation"].append({"points": [[nums[0], nums[1]], [nums[2], nums[3]]], "x": nums[4],
                                  "expected": hexes(m)[0]})
    m = re.search(r"nonLinearCase = do(.*)", lh, re.S).group(1)
    nums = [int(n) for n in re.findall(r"mkScalar (\d+)\b", m)]
    kats["interpolation"].append({"points": [[nums[i], nums[i + 1]] for i in range(0, 8, 2)], "x": nums[8],
                                  "expected": hexes(m)[0]})

    pd = read("plinth-verifier/plutus-halo2/test/ProofData.hs")
    vals = {}
    for name, h in re.findall(r"^(\w+) = mkScalar 0x([0-9a-fA-F]+)", pd, re.M):
        vals[name] = h.lower()
    pts = {}
    for name, x, y in re.findall(r"^(\w+) =\n\s+\(constructG1Point \. bimap mkFp mkFp\)\n\s+\( 0x([0-9a-fA-F]+)\n\s+, 0x([0-9a-fA-F]+)", pd, re.M):
        pts[name] = [x.lower(), y.lower()]
    cm = re.search(r"commitmentMap =\n(.*?)\n\n", pd, re.S).group(1)
    entries = []
    for name, idx, points, evs in re.findall(r"\((\w+), (\d+), \[([^\]]*)\], \[([^\]]*)\]\)", cm):
        entries.append({"commitment": name, "set": int(idx), "points": [p.strip() for p in points.split(",")],
                        "evals": [e.strip() for e in evs.split(",")]})
    mo = read("plinth-verifier/plutus-halo2/test/Halo2MultiOpenMSM.hs")
    hs = hexes(mo)
    kats["multiopen"] = {"scalars": vals, "points": pts, "commitment_map": entries,
                         "point_sets": [["x_current", "x_next"], ["x_current"], ["x_current", "x_next", "x_last"]],
                         "expected_v": hs[0], "expected_f_eval": hs[1],
                         "expected_q_eval_sets": [hs[2:4], hs[4:5], hs[5:8]]}

    gt = read("aiken-verifier/templates/gates_test.hbs")
    lets = dict((n, h.lower()) for n, h in re.findall(r"let (\w+) = from_int\(0x([0-9a-fA-F]+)\)", gt))
    exps = dict((n, h.lower()) for n, h in re.findall(r"expect (\w+) == from_int\(0x([0-9a-fA-F]+)\)", gt))
    kats["lookup_identities"] = {"inputs": lets, "expected": exps}

    vh = read("aiken-verifier/templates/verification_h2.hbs")
    kats["const_delta"] = re.search(r"scalarDelta = from_int\(0x([0-9a-f]+)\)", vh).group(1)
    vc = read("aiken-verifier/templates/vk_constants.hbs")
    kats["neg_g1_generator"] = re.search(r'neg_g1_generator[^#]*#"([0-9a-fA-F]+)"', vc).group(1).lower()
    bt = read("plinth-verifier/plutus-halo2/src/Plutus/Crypto/BlsTypes.hs")
    kats["field_primes_decimal"] = re.findall(r"(\d{70,})", bt)[:2]
    bu = read("aiken-verifier/aiken_halo2/lib/bls_utils.ak")
    kats["fp_prime"] = re.search(r"fp_prime: Int =\n\s*0x([0-9a-f]+)", bu).group(1)
    # bls_utils.ak:119-128 coord_generator: g1_from_coords(x, y) == generator although y is NOT the generator's y - the
    # function takes only the SIGN of y (the recursion fold rebuilds the accumulator points this way)
    cg = hexes(test_block(bu, "coord_generator"))
    kats["g1_from_coords_generator"] = {"x": cg[0], "y": cg[1]}
    # docs/chip_profiles.json (written by src/plutus_gen/stats/profile.rs:54-162 at pi = 1, ci = 0): the numbers only.
    # commitment_map is kept as [rotation-set size, #commitments] pairs in the file's order, which is the BTreeMap order
    # of RotationSet (stats/chips/types/rotation_set.rs:2-11: derive(Ord) over first, prev, curr, next, next2, next3, last)
    prof = json.loads(read("docs/chip_profiles.json"))
    keep = ("degree", "advice_cols", "fixed_cols", "copy_constraints", "nb_lookup_tables", "gates", "gate_expressions",
            "lookups", "trash", "proof_commitments", "vk_commitments", "evals", "gate_ops", "lookup_ops", "trash_ops",
            "proof_size", "vk_size")
    kats["chip_profiles"] = {}
    for chip in ("sha256", "secp256k1"):
        p = prof[chip]
        d = {k: p[k] for k in keep}
        d["commitment_map_sets"] = [[s[0], len(s)] for s in p["commitment_map"]]
        assert all(len(set(s)) == 1 for s in p["commitment_map"])
        kats["chip_profiles"][chip] = d
    with open(OUT, "w") as f:
        json.dump(kats, f, indent=1, sort_keys=True)
    print("wrote", OUT, len(json.dumps(kats)), "bytes")


if __name__ == "__main__":
    main()
