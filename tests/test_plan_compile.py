"""h2v_plan_compile: the C++ plan compiler behind the C-ABI (csrc/h2v_plancc.hpp, the counterpart of the reference's
extract_circuit, /root/reference/src/plutus_gen/extraction/mod.rs:31-232) against plan.py - byte-identical blobs for every
built-in circuit and for a fuzzed family of shapes, and the same refusals.  Host-only: runs without a GPU."""
import json
import os
import random

import pytest

from plutus_halo2_verifier_gen_amd import backend, bls12_381 as bls, plan as PL, vk as V



@pytest.mark.parametrize("name", sorted(V.BUILDERS))
def test_cpp_compiler_matches_plan_py_on_the_builtin_circuits(name):
    vk, _ = V.BUILDERS[name]()
    want = PL.compile_plan(vk).to_bytes()
    got = backend.plan_compile(vk.to_json())
    assert got == want, (name, len(got), len(want))
    if name in ("sha256", "secp256k1"):      # the chip-alone variants of the profile-pinned shapes too
        vk2, _ = V.BUILDERS[name](chip_alone=True)
        assert backend.plan_compile(vk2.to_json()) == PL.compile_plan(vk2).to_bytes()


def _random_shape(rng, i):
    degree = rng.choice([3, 4, 5, 6])
    n_adv = rng.randrange(1, 7)
    n_fix = rng.randrange(1, 9)
    rots = [[0], [0, 1], [0, -1], [0, 1, -1], [0, 2], [0, 1, 3], [0, -2]]
    adv = [rng.choice(rots) for _ in range(n_adv)]
    n_ci = rng.choice([0, 0, 1])
    n_pi = rng.choice([0, 1, 2, 5]) if n_ci == 0 else rng.choice([1, 3])
    n_cc = rng.randrange(1, min(n_adv + n_fix, 9) + 1)
    n_lk = rng.choice([0, 0, 1, 2]) if degree >= 5 else 0
    trash = tuple(rng.randrange(1, 3) for _ in range(rng.choice([0, 0, 1, 2])))
    g = rng.randrange(0, 4)
    vk, td = V._shaped_vk("fuzz%d" % i, rng.randrange(1 << 30), k=rng.randrange(3, 18), degree=degree, n_adv=n_adv, n_fix=n_fix,
                          n_cc=n_cc, lookup_arg_exprs=[rng.randrange(1, 4) for _ in range(n_lk)], gate_exprs=g,
                          gate_ops={"mul": rng.randrange(g, 6 * g + 1), "add": rng.randrange(g, 5 * g + 1), "neg": rng.randrange(0, g + 1)} if g else {"mul": 0, "add": 0, "neg": 0},
                          adv_rot_sets=adv, n_pi=max(n_pi, 1) if n_ci == 0 and False else n_pi, n_ci=n_ci, bf=rng.randrange(3, 8), trash_exprs=trash)
    if rng.random() < 0.4:
        vk.advice_column_phase = [rng.randrange(0, 3) for _ in range(n_adv)]
        top = max(vk.advice_column_phase)
        vk.challenge_phase = [rng.randrange(0, top + 1) for _ in range(rng.randrange(0, 4))]
    return vk


def test_cpp_compiler_matches_plan_py_on_fuzzed_shapes():
    """Random small keys: degrees 3..6, 1..6 advice and 1..8 fixed columns with custom rotations, 0..2 lookup arguments and
    trashcans, with and without a committed instance / public inputs / phases, k = 3..17."""
    rng = random.Random(20260)
    done = 0
    for i in range(60):
        try:
            vk = _random_shape(rng, i)
            V.validate(vk)
        except (V.VKError, ValueError, AssertionError, ZeroDivisionError):
            continue                                  # (a shape the builder cannot make: draw again)
        want = PL.compile_plan(vk).to_bytes()
        assert backend.plan_compile(vk.to_json()) == want, vk.to_json()[:400]
        done += 1
    assert done >= 30


def test_cpp_compiler_refuses_what_the_reference_cannot_emit():
    vk, _ = V.lookup_table_vk()
    base = json.loads(vk.to_json())

    def refused(mutate, needle):
        d = json.loads(json.dumps(base))
        mutate(d)
        with pytest.raises(backend.H2VError, match=needle):
            backend.plan_compile(json.dumps(d))

    refused(lambda d: d["gates"].append(["selector", 0]), "selector")                       # languages/aiken.rs:134-156
    refused(lambda d: d["lookups"][0][0].append(["challenge", 0]), "challenge")
    refused(lambda d: d.update(extra_field=1), "unknown field")
    refused(lambda d: d.update(transcript_repr=bls.R), "canonical")
    refused(lambda d: d["permutation_columns"].append(["fixed", 0]), "permutation")        # commitment count mismatch
    refused(lambda d: d["gates"].append(["advice", 99]), "out of range")
    refused(lambda d: d["fixed_commitments"].__setitem__(0, "00" * 48), "G1")               # compression flag not set
    refused(lambda d: d.update(s_g2="c0" + "00" * 95), "infinity")
    refused(lambda d: d.update(n_committed_instances=2), "out of range")
    refused(lambda d: d.update(challenge_phase=[1]), "phase")
    with pytest.raises(backend.H2VError, match="JSON"):
        backend.plan_compile("{not json")
    # a point of the curve outside G1 is refused like in plan.py (bls12_381.g1_decompress checks the subgroup)
    x = 4
    while True:
        y = bls.fp_sqrt((x ** 3 + 4) % bls.P)
        if y is not None and not bls.g1_in_subgroup((x, y)):
            break
        x += 1
    raw = bytearray(x.to_bytes(48, "big"))
    raw[0] |= 0x80 | (0x20 if y > bls.P - y else 0)
    refused(lambda d: d["fixed_commitments"].__setitem__(0, bytes(raw).hex()), "subgroup")
