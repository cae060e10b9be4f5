"""RLC batch-accept mode and its bucket (Pippenger) G1 MSM: GPU parity tests through the C-ABI.

Reference algebra: the dual-MSM form er = final_com + v (-G1) + x3 pi, el = pi of
/root/reference/aiken-verifier/aiken_halo2/lib/halo2_kzg.ak:37-43 and the pairing equation of
aiken-verifier/templates/verification_h2.hbs:125-128, summed over a batch with 128-bit coefficients.
The bar: accept[] of the RLC entry point == the oracle's per-proof verdicts (the batch check only decides WHICH kernels
produce them), and the bucket MSM == the oracle's naive fold (bls_utils.ak:77-86) bit for bit."""
import json
import random

import pytest

from plutus_halo2_verifier_gen_amd import bls12_381 as bls

pytestmark = pytest.mark.gpu
R = bls.R


@pytest.fixture(scope="module")
def be():
    from plutus_halo2_verifier_gen_amd import backend
    assert backend.device_count() >= 1, "no GPU visible"
    return backend


@pytest.fixture(scope="module")
def circuits():
    from plutus_halo2_verifier_gen_amd import backend, plan as PL, vk as V
    from oracle import binding as orc
    out = {}
    for name, build in V.BUILDERS.items():
        vk, td = build()
        pl = PL.compile_plan(vk)
        dp = backend.DevicePlan(pl.to_bytes(), 0)
        ov = orc.OracleVK(orc.vk_desc(json.loads(vk.to_json()), vk.omega, vk.omega_inv, vk.barycentric_weight))
        out[name] = (vk, td, pl, dp, ov)
    return out


def _bases(orc, k, seed):
    """k distinct points of G1 as (affine, compressed), multiples of the generator"""
    rng = random.Random(seed)
    g = bls.G1_GEN
    pts = [orc.g1_msm([rng.randrange(1, R)], [g]) for _ in range(k)]
    return pts, [bls.g1_compress(p) for p in pts]


@pytest.mark.parametrize("n", [1, 63, 4096, 65536, 600000])
def test_bucket_msm_matches_the_naive_fold(be, orc, n):
    """h2v_probe_g1_msm_pippenger == orc.g1_msm (the reference's fold of scale + add).  The n terms reuse a small set of
    distinct bases (so equal points meet in the buckets: the complete-addition path), and include zero scalars, the
    point at infinity, the scalars 1 and r - 1, opposite points with equal scalars and GLV edge values.
    600 000 terms: window buckets of 2047 .. 2560 entries, the range in which round 2's size classes handed out more lanes
    than the launch had (k_pip_scan's histogram stopped at 2047; tests/test_host_logic.py has the integer model)."""
    rng = random.Random(n)
    k = min(n, 200)
    aff, comp = _bases(orc, k, 100 + n)
    inf = bls.g1_compress(None)
    scalars, bases, which = [], [], []
    lam = bls.GLV_LAMBDA if hasattr(bls, "GLV_LAMBDA") else (0xac45a4010001a40200000000ffffffff)
    edge = [0, 1, R - 1, lam, lam - 1, lam + 1, (1 << 128) - 1, 1 << 128, (1 << 255) % R, R - lam]
    for i in range(n):
        j = rng.randrange(k)
        s = rng.randrange(R) if rng.random() < 0.9 else edge[rng.randrange(len(edge))]
        if n > 8 and i % 97 == 5:
            bases.append(inf); which.append(None)
        else:
            bases.append(comp[j]); which.append(j)
        scalars.append(s)
    if n >= 63:   # P and -P with the same scalar, twice the same term
        neg = bls.g1_compress((aff[0][0], bls.P - aff[0][1]))
        scalars[0:4] = [12345, 12345, 777, 777]
        bases[0:4] = [comp[0], neg, comp[1], comp[1]]
        which[0:4] = [0, "neg0", 1, 1]
    got = be.probe_g1_msm_pippenger(scalars, bases)
    # expected: group the scalars per distinct base (exact, mod r) and fold the k sums with the oracle
    sums = [0] * k
    for s, w in zip(scalars, which):
        if w is None:
            continue
        if w == "neg0":
            sums[0] = (sums[0] - s) % R
        else:
            sums[w] = (sums[w] + s) % R
    want = orc.g1_msm(sums, aff)
    assert got == want
    if n <= 63:   # and literally term by term for the small sizes
        pts = [None if w is None else ((aff[0][0], bls.P - aff[0][1]) if w == "neg0" else aff[w]) for w in which]
        assert orc.g1_msm(scalars, pts) == got


def test_bucket_msm_degenerate_inputs(be, orc):
    aff, comp = _bases(orc, 3, 9)
    inf = bls.g1_compress(None)
    assert be.probe_g1_msm_pippenger([0, 0, 0], comp) is None                       # all scalars zero
    assert be.probe_g1_msm_pippenger([5, 7], [inf, inf]) is None                    # all bases infinity
    assert be.probe_g1_msm_pippenger([R - 1, 1], [comp[0], comp[0]]) is None        # (r - 1) P + P
    assert be.probe_g1_msm_pippenger([1], [comp[2]]) == aff[2]
    assert be.probe_g1_msm_pippenger([2, R - 2, 9], [comp[1], comp[1], comp[2]]) == orc.g1_msm([9], [aff[2]])


NAMES = ["simple_mul", "lookup_table", "atms_with_lookups", "sha256", "secp256k1", "trashcan_mix"]
# corruptions that are caught before the pairing (status != 0): the proof is rejected by itself and leaves the batch
PRE_PAIRING = ["bad_point_flag", "point_not_on_curve", "point_not_in_subgroup", "noncanonical_scalar",
               "noncanonical_instance", "truncated"]
AT_PAIRING = ["flip_first_scalar", "flip_last_scalar", "wrong_public_input", "wrong_pi", "infinity_commitment"]


@pytest.mark.parametrize("name", NAMES)
def test_rlc_all_accepting_batch(be, circuits, name):
    from plutus_halo2_verifier_gen_amd import synth
    vk, td, pl, dp, ov = circuits[name]
    n = 96
    batch = synth.forge_batch(vk, td, n, seed=31, plan=pl, workers=1, ci_identity=(name == "sha256"))
    ws = be.Workspace(dp, n)
    got, fell_back = dp.verify_batch_rlc(batch.proofs, batch.proof_off, batch.instances, batch.committed, ws=ws, seed=bytes(range(32)))
    want = ov.verify_batch(batch.proofs, batch.proof_off, batch.instances, batch.committed, threads=8)
    assert list(got) == list(want) == [1] * n
    assert not fell_back            # verdict == AND of the oracle's per-proof verdicts, reached by the batch check alone
    ok, tm = ws.rlc_result()
    assert ok and tm.msm_terms == n * sum(1 for k, _ in pl.terms if k != 1) + sum(1 for k, _ in pl.terms if k == 1)
    # another seed (the OS's): same verdict
    got2, fb2 = dp.verify_batch_rlc(batch.proofs, batch.proof_off, batch.instances, batch.committed, ws=ws)
    assert list(got2) == [1] * n and not fb2


@pytest.mark.parametrize("name", NAMES)
def test_rlc_single_corruptions(be, circuits, name):
    """Any single corruption kind: rejected before the pairing -> that proof alone is rejected and the batch check still
    passes for the rest; caught only by the pairing -> the batch check fails and the per-proof kernels produce the vector."""
    from plutus_halo2_verifier_gen_amd import synth
    vk, td, pl, dp, ov = circuits[name]
    n = 40
    good = synth.forge_batch(vk, td, n, seed=33, plan=pl, workers=1, ci_identity=(name == "sha256"))
    ws = be.Workspace(dp, n)
    ws.set_option(be.OPT_RLC_ROUTE, -1)      # every call takes the batch check first, whatever the calls before it met
    n_pi = vk.n_public_inputs
    for kind in PRE_PAIRING + AT_PAIRING:
        rng = random.Random(sum(kind.encode()))
        victim = rng.randrange(n)
        res = synth.corrupt(pl, good.proof(victim), good.instances[32 * n_pi * victim:32 * n_pi * (victim + 1)], kind, rng)
        if res is None:
            continue
        proofs = [good.proof(i) for i in range(n)]
        insts = [good.instances[32 * n_pi * i:32 * n_pi * (i + 1)] for i in range(n)]
        proofs[victim], insts[victim] = res
        off = [0]
        for p in proofs:
            off.append(off[-1] + len(p))
        pb, ib = b"".join(proofs), b"".join(insts)
        got, fell_back = dp.verify_batch_rlc(pb, off, ib, good.committed, ws=ws, seed=b"\x07" * 32)
        want = ov.verify_batch(pb, off, ib, good.committed, threads=8)
        assert list(got) == list(want), kind
        assert want[victim] == 0 and sum(want) == n - 1, kind
        assert fell_back == (kind in AT_PAIRING), kind
        # and the per-proof entry point agrees
        assert list(dp.verify_batch(pb, off, ib, good.committed, ws=ws)) == list(want), kind


def test_rlc_mixed_batch_and_status(be, circuits):
    """40 % of the proofs corrupted with every kind at once: vector == oracle; status words == the per-proof mode's."""
    import ctypes as C
    from plutus_halo2_verifier_gen_amd import synth
    vk, td, pl, dp, ov = circuits["lookup_table"]
    n = 128
    batch = synth.forge_batch(vk, td, n, seed=35, plan=pl, workers=1)
    kinds = [k for k in synth.CORRUPTIONS if not k.startswith("acc_")]
    batch = synth.with_rejects(pl, batch, vk.n_public_inputs, fraction=0.4, seed=11, kinds=kinds)
    ws = be.Workspace(dp, n)
    ws.set_option(be.OPT_RLC_ROUTE, -1)      # (the second call below must meet the batch check, not the routing)
    got, fell_back = dp.verify_batch_rlc(batch.proofs, batch.proof_off, batch.instances, batch.committed, ws=ws)
    want = ov.verify_batch(batch.proofs, batch.proof_off, batch.instances, batch.committed, threads=8)
    assert list(got) == list(want) == batch.expected
    assert fell_back
    # only pre-pairing rejects: the batch check passes, nothing falls back, and the rejects are exactly those proofs
    batch2 = synth.forge_batch(vk, td, n, seed=36, plan=pl, workers=1)
    batch2 = synth.with_rejects(pl, batch2, vk.n_public_inputs, fraction=0.3, seed=12, kinds=PRE_PAIRING)
    got2, fb2 = dp.verify_batch_rlc(batch2.proofs, batch2.proof_off, batch2.instances, batch2.committed, ws=ws)
    assert list(got2) == batch2.expected and not fb2 and 0 < sum(got2) < n


@pytest.mark.parametrize("name,n", [("simple_mul", 4096), ("lookup_table", 1000), ("sha256", 300)])
@pytest.mark.parametrize("n_rej", [1, 2, 41])
def test_rlc_fallback_localises_rejects(be, circuits, name, n, n_rej):
    """A failed batch check is followed by GROUP checks (64 proofs per group: one small bucket MSM per side and one pairing
    per group) and the per-proof kernels only run inside the groups that fail.  1, 2 and 41 rejecting proofs at random
    positions - corruptions only the pairing catches, mixed with some that are caught before it -, full groups and a ragged
    last one (1000, 300 proofs), MSM block sizes that do and do not divide a group (16 / 34 / 60 terms): the vector is
    the construction's and the oracle's, whatever the seed; status words are zero exactly for the accepted proofs."""
    import torch
    from plutus_halo2_verifier_gen_amd import synth
    vk, td, pl, dp, ov = circuits[name]
    n_pi = vk.n_public_inputs
    good = synth.forge_batch(vk, td, n, seed=51, plan=pl, workers=8, ci_identity=(name == "sha256"))
    rng = random.Random(1000 * n_rej + n)
    victims = rng.sample(range(n), n_rej)
    pre = set(rng.sample(range(n), 3)) - set(victims)        # rejected before the pairing: take no part in any combination
    proofs = [good.proof(i) for i in range(n)]
    insts = [good.instances[32 * n_pi * i:32 * n_pi * (i + 1)] for i in range(n)]
    expected = [1] * n
    for i in victims:
        proofs[i], insts[i] = synth.corrupt(pl, proofs[i], insts[i], AT_PAIRING[rng.randrange(3)], rng)
        expected[i] = 0
    for i in pre:
        proofs[i], insts[i] = synth.corrupt(pl, proofs[i], insts[i], "noncanonical_scalar", rng)
        expected[i] = 0
    off = [0]
    for p_ in proofs:
        off.append(off[-1] + len(p_))
    pb, ib = b"".join(proofs), b"".join(insts)
    ws = be.Workspace(dp, n)
    for seed in (b"\x21" * 32, None):
        got, fell_back = dp.verify_batch_rlc(pb, off, ib, good.committed, ws=ws, seed=seed)
        assert list(got) == expected and fell_back
    # the oracle on the groups that hold a reject (and one clean group)
    groups = sorted({i // 64 for i in victims} | {i // 64 for i in pre} | {0})[:6]
    for g in groups:
        lo, hi = 64 * g, min(n, 64 * g + 64)
        sub_off = [o - off[lo] for o in off[lo:hi + 1]]
        want = ov.verify_batch(pb[off[lo]:off[hi]], sub_off, ib[32 * n_pi * lo:32 * n_pi * hi],
                               good.committed[48 * lo:48 * hi] if good.committed else None, threads=16)
        assert list(want) == expected[lo:hi]
    # device-resident form: status words
    dev = torch.device("cuda", 0)
    up = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) if b else None
    d_p, d_i, d_c = up(pb), up(ib), up(good.committed)
    d_off = torch.tensor(off, dtype=torch.int64).to(dev)
    acc = torch.zeros(n, dtype=torch.uint8, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    dp.verify_batch_rlc_device(n, d_p.data_ptr(), d_off.data_ptr(), d_i.data_ptr(), d_c.data_ptr() if d_c is not None else None,
                               acc.data_ptr(), st.data_ptr(), ws=ws, seed=b"\x22" * 32)
    torch.cuda.synchronize()
    assert acc.cpu().tolist() == expected and [int(x == 0) for x in st.cpu().tolist()] == expected
    ws.close()


def test_rlc_two_batch_sizes_in_flight_on_one_workspace(be, circuits):
    """the group stage of the fall-back caches its argument array per batch size.  Two RLC calls of
    DIFFERENT sizes back to back on ONE ordinary workspace and ONE non-default stream, nothing synchronised in between, the
    first holding a reject only the pairing catches (so its group kernels read the arguments when they finally run): the
    upload of the second size's arguments is ordered behind them on the stream, and both vectors are the construction's.
    Then the same workspace with ANOTHER plan whose MSM has more per-proof terms (the cache is keyed on the plan's load
    counter and on the term-list stride, not on an address)."""
    import torch
    from plutus_halo2_verifier_gen_amd import synth
    dev = torch.device("cuda", 0)
    up = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) if b else None
    ptr = lambda t: t.data_ptr() if t is not None else None
    s = torch.cuda.Stream(device=dev)

    def make(name, n, seed, victims):
        vk, td, pl, dp, ov = circuits[name]
        n_pi = vk.n_public_inputs
        good = synth.forge_batch(vk, td, n, seed=seed, plan=pl, workers=8)
        rng = random.Random(seed)
        proofs = [good.proof(i) for i in range(n)]
        insts = [good.instances[32 * n_pi * i:32 * n_pi * (i + 1)] for i in range(n)]
        exp = [1] * n
        for i in victims:
            proofs[i], insts[i] = synth.corrupt(pl, proofs[i], insts[i], "wrong_pi", rng)
            exp[i] = 0
        off = [0]
        for p_ in proofs:
            off.append(off[-1] + len(p_))
        d = (up(b"".join(proofs)), torch.tensor(off, dtype=torch.int64).to(dev), up(b"".join(insts)), up(good.committed))
        return dp, n, exp, d

    big = make("lookup_table", 1100, 61, [700])
    ws = be.Workspace(big[0], 1100, lanes=None, chunk=None)
    assert ws.lanes()[0] == 1
    seq = [make("simple_mul", 1000, 62, [3]), make("simple_mul", 520, 63, []), make("simple_mul", 777, 64, [600, 601]), big,
           make("simple_mul", 1000, 65, [999])]
    outs = []
    for dp, n, exp, d in seq:
        acc = torch.full((n,), 7, dtype=torch.uint8, device=dev)
        dp.verify_batch_rlc_device(n, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), acc.data_ptr(), None, ws=ws, stream=s.cuda_stream, seed=bytes(range(32)))
        outs.append((acc, exp))
    s.synchronize()
    for k, (acc, exp) in enumerate(outs):
        assert acc.cpu().tolist() == exp, k
    ws.close()


@pytest.mark.parametrize("laned", [False, True])
def test_rlc_calls_are_routed_by_the_observed_failing_group_rate(be, circuits, laned):
    """include/h2v.h, ROUTING: batches in which most groups of 64 hold a pairing-only reject make the workspace send RLC calls
    straight to the per-proof kernels (rlc_result: batch_accepted = 0 and no batch-check timings) from the second such call on;
    a dozen clean batches bring it back to the batch check; every vector is the construction's throughout; with
    H2V_OPT_RLC_ROUTE = -1 nothing is ever routed."""
    import torch
    from plutus_halo2_verifier_gen_amd import synth
    vk, td, pl, dp, ov = circuits["simple_mul"]
    n, n_pi = 1024, vk.n_public_inputs
    dev = torch.device("cuda", 0)
    up = lambda x: torch.frombuffer(bytearray(x), dtype=torch.uint8).to(dev)
    good = synth.forge_batch(vk, td, n, seed=81, plan=pl, workers=8)
    rng = random.Random(82)
    proofs = [good.proof(i) for i in range(n)]
    insts = [good.instances[32 * n_pi * i:32 * n_pi * (i + 1)] for i in range(n)]
    exp_dirty = [1] * n
    for i in rng.sample(range(n), 24):            # 24 rejects in 16 groups: most groups fail
        proofs[i], insts[i] = synth.corrupt(pl, proofs[i], insts[i], "wrong_pi", rng)
        exp_dirty[i] = 0
    off = [0]
    for p_ in proofs:
        off.append(off[-1] + len(p_))
    d_dirty = (up(b"".join(proofs)), torch.tensor(off, dtype=torch.int64).to(dev), up(b"".join(insts)))
    d_clean = (up(good.proofs), torch.tensor(good.proof_off, dtype=torch.int64).to(dev), up(good.instances))
    s = torch.cuda.Stream(device=dev)

    def run(ws, d, want):
        acc = torch.full((n,), 7, dtype=torch.uint8, device=dev)
        dp.verify_batch_rlc_device(n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), None, acc.data_ptr(), None, ws=ws, stream=s.cuda_stream, seed=bytes(range(32)))
        s.synchronize()
        torch.cuda.synchronize()
        assert acc.cpu().tolist() == want
        ok, tm = ws.rlc_result()
        return ok, tm.total_ms

    for route_off in (False, True):
        ws = be.Workspace(dp, n, lanes=2, chunk=n) if laned else be.Workspace(dp, n)
        if route_off:
            ws.set_option(be.OPT_RLC_ROUTE, -1)
        seen = [run(ws, d_dirty, exp_dirty) for _ in range(5)]
        assert all(not ok for ok, _t in seen)
        routed = [t == 0 for _ok, t in seen]
        if route_off:
            assert not any(routed)
        else:
            assert not routed[0] and routed[-1], routed                   # the estimate moves a quarter of the way per call
        back = [run(ws, d_clean, [1] * n) for _ in range(14)]
        assert back[-1][0] and back[-1][1] > 0, back                       # the batch check runs (and passes) again
        if not route_off:
            assert not back[0][0] and back[0][1] == 0                      # ... but the first clean call was still routed
        ws.close()


def test_rlc_duplicate_proofs_and_small_batches(be, circuits):
    """The same proof many times in one batch (equal points with different coefficients meet in the buckets), batches of
    1 and 2 proofs, and a recursive plan (no batch form: runs per proof behind the same entry point)."""
    from plutus_halo2_verifier_gen_amd import synth
    vk, td, pl, dp, ov = circuits["simple_mul"]
    one = synth.forge_batch(vk, td, 1, seed=41, plan=pl, workers=1)
    n = 64
    off = [1120 * i for i in range(n + 1)]
    ws = be.Workspace(dp, n)
    got, fb = dp.verify_batch_rlc(one.proofs * n, off, one.instances * n, None, ws=ws, seed=b"\x01" * 32)
    assert list(got) == [1] * n and not fb
    for m in (1, 2):
        got, fb = dp.verify_batch_rlc(one.proofs * m, off[:m + 1], one.instances * m, None, ws=ws)
        assert list(got) == [1] * m and not fb
    bad = bytearray(one.proofs)
    bad[50] ^= 1                                   # inside a1: another curve point or none at all
    got, fb = dp.verify_batch_rlc(bytes(bad), [0, 1120], one.instances, None, ws=ws)
    assert list(got) == [ov.verify(bytes(bad), one.instance_ints(0, 3), None)] == [0]
    vk2, td2, pl2, dp2, ov2 = circuits["ivc"]
    b2 = synth.forge_batch(vk2, td2, 6, seed=42, plan=pl2, workers=1)
    b2 = synth.with_rejects(pl2, b2, vk2.n_public_inputs, fraction=0.5, seed=2, kinds=["acc_scalar", "flip_first_scalar"])
    ws2 = be.Workspace(dp2, 6)
    got, fb = dp2.verify_batch_rlc(b2.proofs, b2.proof_off, b2.instances, b2.committed, ws=ws2)
    assert list(got) == b2.expected and not fb


def test_rlc_host_batches_growing_into_the_download_block_slack(be, circuits):
    """A host-buffer RLC call downloads accept[n] and, at the next multiple of 8 behind it, the batch verdict word.  The pinned
    block of a workspace grows with slack (n + n / 4 + 64): after a 100-proof call a 186-proof call still "fitted" by its accept
    bytes alone and put the word past the end of the block (hipMemcpyAsync: invalid argument -> H2V_E_DEVICE; found by
    tools/soak.py).  Ordinary workspace (h2v_verify_batch_rlc) and a one-lane submit / wait stream, every size around the
    old block's end."""
    from plutus_halo2_verifier_gen_amd import synth
    vk, td, pl, dp, ov = circuits["simple_mul"]
    pool = synth.forge_batch(vk, td, 200, seed=61, plan=pl, workers=4)
    n_pi = vk.n_public_inputs

    def first(m):
        return pool.proofs[:pool.proof_off[m]], pool.proof_off[:m + 1], pool.instances[:32 * n_pi * m]

    ws = be.Workspace(dp, 256)
    bs = be.BatchStream(dp, 256, 1, rlc=True, seed=b"\x05" * 32)
    keep = []
    for m in [100] + list(range(180, 200)):          # 100 proofs: a block of 189 bytes
        pr, off, ins = first(m)
        got, fb = dp.verify_batch_rlc(pr, off, ins, None, ws=ws, seed=b"\x03" * 32)
        assert list(got) == [1] * m and not fb, m
        hb, k_ = dp.host_batch(pr, off, ins, None)
        keep.append(k_)
        out = bs.push(hb, m)
        if out is not None:
            assert list(out[0]) == [1] * len(out[0]) and not out[1]
    for out in bs.drain():
        assert list(out[0]) == [1] * len(out[0]) and not out[1]
    bs.close()
    ws.close()


def test_rlc_full_size_batch(be, circuits):
    """BASELINE configs[1] size through the batch mode: 4096 accepting proofs -> one bucket MSM of 40 966 terms + one
    pairing; then with 1 % byte flips (the reference example's corruption): the flipped proofs, and only they, reject."""
    from plutus_halo2_verifier_gen_amd import synth
    vk, td, pl, dp, ov = circuits["simple_mul"]
    n = 4096
    batch = synth.forge_batch(vk, td, n, seed=51, plan=pl, workers=8)
    ws = be.Workspace(dp, n)
    got, fb = dp.verify_batch_rlc(batch.proofs, batch.proof_off, batch.instances, batch.committed, ws=ws)
    assert list(got) == [1] * n and not fb
    ok, tm = ws.rlc_result()
    assert ok and tm.msm_terms == n * 10 + 6
    rej = synth.with_rejects(pl, batch, 3, fraction=0.01, seed=77, kinds=["flip_first_scalar"])
    got, fb = dp.verify_batch_rlc(rej.proofs, rej.proof_off, rej.instances, rej.committed, ws=ws)
    assert list(got) == rej.expected and fb
    sample = list(range(0, n, 97))
    for i in sample:
        assert got[i] == int(ov.verify(rej.proof(i), rej.instance_ints(i, 3), None))


def test_rlc_api_misuse(be, circuits):
    """API misuse is an error code, never a launch: the device form without a workspace, a workspace created for a smaller
    plan, a seed of the wrong length."""
    from plutus_halo2_verifier_gen_amd import synth
    vk, td, pl, dp, ov = circuits["simple_mul"]
    with pytest.raises(be.H2VError, match="workspace"):
        dp.verify_batch_rlc_device(4, 16, 16, 16, None, 16, None, ws=None)        # (pointers are never touched)
    vk2, td2, pl2, dp2, ov2 = circuits["lookup_table"]
    b2 = synth.forge_batch(vk2, td2, 2, seed=3, plan=pl2, workers=1)
    with pytest.raises(be.H2VError, match="workspace"):
        dp2.verify_batch_rlc(b2.proofs, b2.proof_off, b2.instances, b2.committed, ws=be.Workspace(dp, 8))
    with pytest.raises(ValueError):
        dp2.verify_batch_rlc(b2.proofs, b2.proof_off, b2.instances, b2.committed, seed=b"short")
    assert dp.verify_batch_rlc(b"", [0], b"", None) == (b"", False)              # the empty batch


@pytest.mark.parametrize("name", ["simple_mul", "lookup_table", "sha256"])
def test_random_byte_flips_three_way(be, circuits, name):
    """Differential fuzz: half of the proofs of a batch get ONE random bit flipped at a random position of the proof bytes,
    of a public input or of the committed instance.  The oracle, the per-proof GPU path and the RLC GPU path must give the
    same vector (most flips hit a commitment: another curve point, a point off the curve or outside G1, a changed flag; or
    a scalar: wrong or non-canonical), including the status the device reports for proofs rejected before the pairing."""
    from plutus_halo2_verifier_gen_amd import synth
    vk, td, pl, dp, ov = circuits[name]
    n, n_pi = 192, vk.n_public_inputs
    batch = synth.forge_batch(vk, td, n, seed=71, plan=pl, workers=4, ci_identity=False)
    rng = random.Random(len(name))
    proofs = [bytearray(batch.proof(i)) for i in range(n)]
    inst = bytearray(batch.instances)
    ci = bytearray(batch.committed) if batch.committed else None
    touched = []
    for i in range(n):
        if rng.random() < 0.5:
            continue
        touched.append(i)
        where = rng.random()
        if where < 0.8 or (n_pi == 0 and ci is None):
            pos = rng.randrange(len(proofs[i]))
            proofs[i][pos] ^= 1 << rng.randrange(8)
        elif where < 0.9 and n_pi:
            pos = 32 * n_pi * i + rng.randrange(32 * n_pi)
            inst[pos] ^= 1 << rng.randrange(8)
        elif ci is not None:
            pos = 48 * i + rng.randrange(48)
            ci[pos] ^= 1 << rng.randrange(8)
        else:
            pos = rng.randrange(len(proofs[i]))
            proofs[i][pos] ^= 1 << rng.randrange(8)
    off = [0]
    for p in proofs:
        off.append(off[-1] + len(p))
    pb, ib, cb = b"".join(bytes(p) for p in proofs), bytes(inst), (bytes(ci) if ci is not None else None)
    want = ov.verify_batch(pb, off, ib, cb, threads=8)
    ws = be.Workspace(dp, n)
    got_pp = dp.verify_batch(pb, off, ib, cb, ws=ws)
    got_rlc, _fb = dp.verify_batch_rlc(pb, off, ib, cb, ws=ws)
    assert list(got_pp) == list(want)
    assert list(got_rlc) == list(want)
    assert all(want[i] == 1 for i in range(n) if i not in touched)
    assert sum(want[i] for i in touched) <= 2      # a one-bit change is (almost) always fatal; never silently ignored en masse


def test_rlc_infinity_points(be, circuits):
    """The point at infinity is a valid G1 encoding: as pi (the left-hand side of one proof vanishes), as a commitment,
    as every commitment of a proof.  Same verdicts as the oracle in both modes; such proofs take part in the batch check
    (their terms are skipped by the bucket MSM) and make it fail when their own equation fails."""
    from plutus_halo2_verifier_gen_amd import synth
    vk, td, pl, dp, ov = circuits["simple_mul"]
    n = 24
    good = synth.forge_batch(vk, td, n, seed=81, plan=pl, workers=1)
    inf = bls.g1_compress(None)
    proofs = [bytearray(good.proof(i)) for i in range(n)]
    o_pi = pl.points[pl.pi_point]
    proofs[3][o_pi:o_pi + 48] = inf                                  # pi = O
    proofs[7][pl.points[1]:pl.points[1] + 48] = inf                  # one commitment = O
    for o in pl.points:                                              # every G1 element = O
        proofs[11][o:o + 48] = inf
    off = [0]
    for p in proofs:
        off.append(off[-1] + len(p))
    pb = b"".join(bytes(p) for p in proofs)
    want = ov.verify_batch(pb, off, good.instances, None, threads=4)
    ws = be.Workspace(dp, n)
    assert list(dp.verify_batch(pb, off, good.instances, None, ws=ws)) == list(want)
    got, fb = dp.verify_batch_rlc(pb, off, good.instances, None, ws=ws)
    assert list(got) == list(want) and fb
    assert want[3] == 0 and want[7] == 0 and sum(want) >= n - 3


def test_rlc_one_stream_form(be, circuits):
    """The one-stream form of the RLC mode (decompression ahead of the combiner on the caller's stream: for callers with many
    batches in flight) through H2V_OPT_STREAMS = 1 and through h2v_rlc_opts.flags & H2V_RLC_ONE_STREAM, in-process: a batch with
    every corruption kind gives the construction's vector, as the two-stream form does."""
    import torch
    from plutus_halo2_verifier_gen_amd import synth
    dev = torch.device("cuda", 0)
    up = lambda x: torch.frombuffer(bytearray(x), dtype=torch.uint8).to(dev) if x else None
    ptr = lambda t: t.data_ptr() if t is not None else None
    for name in ("simple_mul", "lookup_table", "sha256"):
        vk, td, pl, dp, ov = circuits[name]
        n = 300
        b = synth.forge_batch(vk, td, n, seed=17, plan=pl, workers=4, ci_identity=(name == "sha256"))
        b = synth.with_rejects(pl, b, vk.n_public_inputs, fraction=0.1, seed=18, kinds=list(synth.CORRUPTIONS))
        d = (up(b.proofs), torch.tensor(b.proof_off, dtype=torch.int64).to(dev), up(b.instances), up(b.committed))
        for streams, flag in ((1, False), (0, False), (-1, True)):
            ws = be.Workspace(dp, n)
            ws.set_option(be.OPT_STREAMS, streams)
            acc = torch.full((n,), 7, dtype=torch.uint8, device=dev)
            dp.verify_batch_rlc_device(n, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), acc.data_ptr(), None, ws=ws, stream=None, seed=bytes(range(32)), one_stream=flag)
            torch.cuda.synchronize()
            assert acc.cpu().tolist() == b.expected, (name, streams, flag)
            ws.close()
