"""Recursion (IVC) accumulator fold (emitters/aiken.rs:648-757, docs/algorithms.html "Recursion").
CPU: the big-integer model (ivc.py), the plan compiler and the C oracle agree on layout, fold and verdicts.
GPU: the device path (accumulator decompression, three extra sums, challenge hash, two fold MSMs) against the oracle."""
import json
import random

import pytest

from plutus_halo2_verifier_gen_amd import bls12_381 as bls
from plutus_halo2_verifier_gen_amd import ivc
from plutus_halo2_verifier_gen_amd import plan as PL
from plutus_halo2_verifier_gen_amd import synth
from plutus_halo2_verifier_gen_amd import vk as V

R, P = bls.R, bls.P


@pytest.fixture(scope="module")
def case():
    from oracle import binding as orc
    vk, td = V.ivc_vk()
    pl = PL.compile_plan(vk)
    ov = orc.OracleVK(orc.vk_desc(json.loads(vk.to_json()), vk.omega, vk.omega_inv, vk.barycentric_weight))
    batch = synth.forge_batch(vk, td, 12, seed=5, plan=pl, workers=1)
    return vk, td, pl, ov, batch


def test_layout_follows_the_emitter():
    """Index arithmetic of emitters/aiken.rs:705-741 on the IVC-shaped key: N = 28, F = 15."""
    vk, _ = V.ivc_vk()
    lay = ivc.layout(vk)
    n, f = vk.n_public_inputs, lay["F"]
    assert f == 1 + len(vk.fixed_commitments) + len(vk.permutation_commitments) + 4
    one_based = lambda k: k + 1
    assert [one_based(k) for k in lay["left_x"]] == [n - f - 8, n - f - 9]
    assert [one_based(k) for k in lay["left_y"]] == [n - f - 6, n - f - 7]
    assert one_based(lay["left_scalar"]) == n - f - 5
    assert [one_based(k) for k in lay["right_x"]] == [n - f - 3, n - f - 4]
    assert [one_based(k) for k in lay["right_y"]] == [n - f - 1, n - f - 2]
    assert one_based(lay["right_scalar"]) == n - f
    assert [one_based(k) for k in lay["fixed_scalars"]] == list(range(n - f + 1, n + 1))
    # the packing base is (2^56)^4 as a decimal literal in the emitter (aiken.rs:707)
    assert ivc.B224 == 26959946667150639794667015087019630673637144422540572481103610249216
    for v in (0, 1, P - 1, 12345 << 300):
        assert ivc.coord(*ivc.split_coord(v)) == v % P
    small = V.simple_mul_vk()[0]
    small.recursion_vks = []
    with pytest.raises(ValueError):
        ivc.layout(small)                    # "Not enough public inputs to support recursion" (aiken.rs:702)


def test_g1_from_coords_reference_vector(kats, orc):
    """bls_utils.ak:119-128 `coord_generator`: g1_from_coords(x_G, y') is the generator for a y' that is NOT the generator's
    y but has its sign - the reference-held vector for the function the recursion fold rebuilds accumulator points with."""
    x, y = int(kats["g1_from_coords_generator"]["x"], 16), int(kats["g1_from_coords_generator"]["y"], 16)
    assert x == bls.G1_GEN[0] and y != bls.G1_GEN[1]
    assert ivc.g1_from_coords(x, y) == bls.G1_GEN
    # the same through the byte encoding the device path builds (x big-endian + the parity flag of y) and the oracle's decompress
    enc = bytearray(x.to_bytes(48, "big"))
    enc[0] |= 0x80 | (0x20 if y > (bls.P - 1) // 2 else 0)
    ok, pt = orc.g1_decompress(bytes(enc))
    assert ok and pt == bls.G1_GEN


def test_g1_from_coords_uses_only_the_sign_of_y():
    pt = bls.g1_mul(bls.G1_GEN, 99)
    assert ivc.g1_from_coords(pt[0], pt[1]) == pt
    other = P - 5 if pt[1] > P - pt[1] else 5          # same parity class as y, not y
    assert (other > P - other) == (pt[1] > P - pt[1])
    assert ivc.g1_from_coords(pt[0], other) == pt
    assert ivc.g1_from_coords(pt[0], P - pt[1]) == bls.g1_neg(pt)
    with pytest.raises(ivc.Reject):
        x = 1
        while bls.fp_sqrt((x ** 3 + 4) % P) is not None:
            x += 1
        ivc.g1_from_coords(x, 1)


def test_plan_sections_and_interpreter(case):
    vk, td, pl, ov, batch = case
    assert pl.is_recursive and pl.n_terms == pl.n_main_terms + 2 + ivc.layout(vk)["F"]
    assert pl.term_names[pl.n_main_terms:pl.n_main_terms + 3] == ["acc_left", "acc_right", "neg_g1_generator"]
    inst = batch.instance_ints(0, vk.n_public_inputs)
    scal, _, st = PL.run_plan(pl, batch.proof(0), inst, None)
    lay = ivc.layout(vk)
    assert st is None
    assert scal[pl.n_main_terms] == inst[lay["left_scalar"]] and scal[pl.n_main_terms + 1] == inst[lay["right_scalar"]]
    assert scal[pl.n_main_terms + 2:] == [inst[k] for k in lay["fixed_scalars"]]
    bad = list(inst)
    bad[0] ^= 1
    assert PL.run_plan(pl, batch.proof(0), bad, None)[2] == "recursion"
    blob = pl.to_bytes()
    hdr = [int.from_bytes(blob[8 + 4 * k:12 + 4 * k], "little") for k in range(PL.PLAN_HDR_WORDS)]
    assert hdr[0] == PL.PLAN_VERSION and hdr[25] == 1 and hdr[26] == pl.n_main_terms and hdr[27:35] == pl.acc_coords
    plain = PL.compile_plan(V.simple_mul_vk()[0])
    assert not plain.is_recursive and plain.n_main_terms == plain.n_terms


def test_oracle_fold_matches_the_big_integer_model(case):
    from oracle import binding as orc
    vk, td, pl, ov, batch = case
    assert list(ov.verify_batch(batch.proofs, batch.proof_off, batch.instances, None, threads=4)) == [1] * batch.n
    sg2 = bls.g2_decompress(bytes.fromhex(vk.s_g2))
    for i in (0, 7):
        inst = batch.instance_ints(i, vk.n_public_inputs)
        proof = batch.proof(i)
        scal, _, _ = PL.run_plan(pl, proof, inst, None)
        pts = [bls.g1_decompress(proof[o:o + 48]) for o in pl.points]
        er = None
        for t, (k, idx) in enumerate(pl.terms[:pl.n_main_terms]):
            er = bls.g1_add(er, bls.g1_mul(pts[idx] if k == PL.TERM_PROOF_POINT else pl.vk_bases[idx], scal[t]))
        el2, er2, c = ivc.fold(vk, inst, pts[pl.pi_point], er)
        ok, tr = ov.verify(proof, inst, None, trace=True)
        assert ok and tr.point("el") == el2 and tr.point("er") == er2
        assert bls.pairing(el2, sg2) == bls.pairing(er2, bls.G2_GEN)


def test_oracle_rejects_every_accumulator_corruption(case):
    from oracle import binding as orc
    vk, td, pl, ov, batch = case
    rng = random.Random(3)
    want = {"acc_limb": ("point", "pairing"), "acc_scalar": ("pairing",), "acc_fixed_scalar": ("pairing",),
            "acc_sign": ("pairing",), "acc_vk_hash": ("recursion",)}
    n_pi = vk.n_public_inputs
    for kind, reasons in want.items():
        for i in range(3):
            p, ins = synth.corrupt(pl, batch.proof(i), batch.instances[32 * n_pi * i:32 * n_pi * (i + 1)], kind, rng)
            ok, tr = ov.verify(p, [int.from_bytes(ins[32 * j:32 * j + 32], "little") for j in range(n_pi)], None, trace=True)
            assert not ok and orc.STATUS[tr.status] in reasons, (kind, orc.STATUS[tr.status])
    assert synth.corrupt(PL.compile_plan(V.simple_mul_vk()[0]), b"", b"", "acc_limb", rng) is None


@pytest.mark.gpu
def test_ivc_fold_on_gpu(case):
    from plutus_halo2_verifier_gen_amd import backend
    vk, td, pl, ov, batch = case
    dp = backend.DevicePlan(pl.to_bytes(), 0)
    n_pi = vk.n_public_inputs
    big = synth.forge_batch(vk, td, 96, seed=8, plan=pl, workers=1)
    mixed = synth.with_rejects(pl, big, n_pi, fraction=0.5, seed=4, kinds=list(synth.CORRUPTIONS))
    got = dp.verify_batch(mixed.proofs, mixed.proof_off, mixed.instances, None)
    want = ov.verify_batch(mixed.proofs, mixed.proof_off, mixed.instances, None, threads=8)
    assert list(got) == list(want) == mixed.expected and 0 < sum(got) < mixed.n
    # every pairing engine takes the FOLDED left-hand point (el_jac), the six-lanes-per-proof one included (96 proofs: ten waves,
    # the last with four idle groups)
    ws = backend.Workspace(dp, mixed.n)
    for engine in (6, 12, 16, 32, 64):
        ws.set_option(backend.Workspace.OPT_PAIRING_ENGINE, engine)
        assert list(dp.verify_batch(mixed.proofs, mixed.proof_off, mixed.instances, None, ws=ws)) == mixed.expected, engine
        assert ws.timings().pairing_lanes_per_proof == engine
    ws.close()
    # el / er entering the pairing are the folded ones, bit for bit
    for i in (mixed.expected.index(1), 0):
        proof = big.proof(i)
        inst = big.instances[32 * n_pi * i:32 * n_pi * (i + 1)]
        ok, otr = ov.verify(proof, big.instance_ints(i, n_pi), None, trace=True)
        tr = dp.trace(proof, inst, None)
        assert ok and tr["accept"] == 1 and tr["el"] == otr.point("el") and tr["er"] == otr.point("er")
        assert tr["msm_scalars"][pl.n_main_terms:] == PL.run_plan(pl, proof, big.instance_ints(i, n_pi), None)[0][pl.n_main_terms:]
    # status bits of the accumulator-specific rejections
    rng = random.Random(6)
    for kind, bit in (("acc_vk_hash", 32), ("acc_limb", 8 | 16), ("acc_sign", 16)):
        p, ins = synth.corrupt(pl, big.proof(1), big.instances[32 * n_pi:64 * n_pi], kind, rng)
        tr = dp.trace(p, ins, None)
        assert tr["accept"] == 0 and tr["status"] & bit, (kind, tr["status"])
