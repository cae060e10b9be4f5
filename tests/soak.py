#!/usr/bin/env python3
"""Soak run of the C-ABI on one GPU: several host threads, each with workspaces of its own, verify randomly composed
batches of several circuits at once - random sizes (1 .. --max-n, ragged), random mixes of forged and corrupted proofs
(every corruption kind of synth.CORRUPTIONS), through every calling form of include/h2v.h: host buffers with and without a
caller-owned workspace, h2v_verify_batch_submit / _wait streams, device pointers on laned workspaces with deferred joins and
several calls in flight, per-proof and RLC mode, and workspaces with randomly forced launch shapes (pairing engine, MSM terms
per lane).  Every accept vector is compared with the construction (a forged proof accepts, a corrupted one rejects); the
"fuzz" form mutates proofs at random (bit flips, random bytes, 48-byte windows spliced in from other proofs, encodings of
the point at infinity, changed lengths) and takes the expected verdicts of the mutated proofs from the ORACLE (which is why
this file lives under tests/).  The plans are shared by the threads (h2v.h: "may be shared by threads, each with its
own workspace"); the pool of sixteen library streams is shared by all their laned workspaces.
Exit code 0: every call agreed.  usage: soak.py [--minutes M] [--threads T] [--max-n N] [--seed S] [--circuits a,b] [--forms host,host_ws,host_laned,host_rlc,submit,device,device_rlc,fuzz,churn,multi]"""
import argparse
import os
import random
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def permute(synth, batch, order, n_pi):
    proofs = [batch.proof(i) for i in order]
    off = [0]
    for p in proofs:
        off.append(off[-1] + len(p))
    inst = b"".join(batch.instances[32 * n_pi * i:32 * n_pi * (i + 1)] for i in order)
    ci = None if batch.committed is None else b"".join(batch.ci(i) for i in order)
    return synth.Batch(n=len(order), proofs=b"".join(proofs), proof_off=off, instances=inst, committed=ci, expected=[batch.expected[i] for i in order])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=3.0)
    ap.add_argument("--threads", type=int, default=3)
    ap.add_argument("--max-n", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--circuits", default="simple_mul,lookup_table,atms_with_lookups,trashcan_mix,phased,secp256k1,ivc,sha256")
    ap.add_argument("--forms", default="host,host_ws,host_laned,host_rlc,submit,device,device_rlc,fuzz,churn,multi")
    args = ap.parse_args()
    import json
    import torch
    from plutus_halo2_verifier_gen_amd import backend, plan as PL, synth, vk as V
    from oracle import binding as orc
    dev = torch.device("cuda", 0)
    pools, oracles = {}, {}
    oracle_lock = threading.Lock()
    t0 = time.time()
    for k, name in enumerate(args.circuits.split(",")):
        vk, td = V.BUILDERS[name]()
        pl = PL.compile_plan(vk)
        dp = backend.DevicePlan(pl.to_bytes(), 0)
        m = 1200 if len(pl.terms) <= 24 else 300
        b = synth.forge_batch(vk, td, m, seed=900 + k, plan=pl, workers=8)
        b = synth.with_rejects(pl, b, vk.n_public_inputs, fraction=0.12, seed=950 + k, kinds=list(synth.CORRUPTIONS))
        pools[name] = (vk, pl, dp, b)
        oracles[name] = orc.OracleVK(orc.vk_desc(json.loads(vk.to_json()), vk.omega, vk.omega_inv, vk.barycentric_weight))
    print("pools forged in %.1f s: %s" % (time.time() - t0, {k: v[3].n for k, v in pools.items()}), flush=True)
    deadline = time.time() + 60.0 * args.minutes
    stats = {"calls": 0, "proofs": 0, "bad": 0}
    by_form = {}
    lock = threading.Lock()
    failures = []

    def worker(tid):
        rng = random.Random(args.seed * 1000 + tid)
        stream = torch.cuda.Stream(device=dev)
        wss = {}            # (circuit, kind) -> workspace owned by this thread

        def ws_for(name, kind, n):
            key = (name, kind)
            dp = pools[name][2]
            if key in wss and wss[key].max_batch >= n:
                return wss[key]
            if key in wss:
                wss.pop(key).close()
            cap = max(n, 64)
            if kind == "plain":
                w = backend.Workspace(dp, cap, lanes=1, chunk=cap)
            elif kind == "laned":
                w = backend.Workspace(dp, cap, lanes=rng.choice([2, 3, 4, 8, 16]), chunk=max(16, cap // rng.choice([1, 2, 3, 5])))
                w.defer_joins(True)
            else:           # the library's own choice of lanes and chunk
                w = backend.Workspace(dp, cap)
            if rng.random() < 0.5:      # a forced launch shape: the verdicts may not depend on it
                w.set_option(backend.Workspace.OPT_PAIRING_ENGINE, rng.choice([0, 6, 12, 16, 32, 64]))
                w.set_option(backend.Workspace.OPT_MSM_TERMS_PER_LANE, rng.choice([0, 1, 2, 3, 4]))
            wss[key] = w
            return w

        def mutate(name, b):
            """b with about 5 % of its proofs mutated at random; expected verdicts of those from the oracle"""
            vk, pl, dp, pool = pools[name]
            n_pi = vk.n_public_inputs
            proofs = [bytearray(b.proof(i)) for i in range(b.n)]
            inst = bytearray(b.instances)
            ci = bytearray(b.committed) if b.committed else None
            touched = sorted(rng.sample(range(b.n), max(1, b.n // 20)))
            for i in touched:
                kind = rng.choice(["bit", "bit", "byte", "splice", "infinity", "shorter", "longer", "instance", "committed"])
                pr = proofs[i]
                if kind == "instance" and n_pi:
                    inst[32 * n_pi * i + rng.randrange(32 * n_pi)] ^= 1 << rng.randrange(8)
                elif kind == "committed" and ci is not None:
                    ci[48 * i + rng.randrange(48)] ^= 1 << rng.randrange(8)
                elif kind == "byte":
                    pr[rng.randrange(len(pr))] = rng.randrange(256)
                elif kind == "splice":
                    other = pool.proof(rng.randrange(pool.n))
                    o = 48 * rng.randrange(max(1, min(len(pr), len(other)) // 48))
                    pr[o:o + 48] = other[o:o + 48]
                elif kind == "infinity":
                    o = rng.choice(pl.points)
                    pr[o:o + 48] = bytes([rng.choice([0xC0, 0xC0, 0xE0, 0x40])]) + bytes(47)
                elif kind == "shorter":
                    del pr[len(pr) - rng.randrange(1, 49):]
                elif kind == "longer":
                    pr.extend(rng.randrange(256) for _ in range(rng.randrange(1, 49)))
                else:
                    pr[rng.randrange(len(pr))] ^= 1 << rng.randrange(8)
            off = [0]
            for pr in proofs:
                off.append(off[-1] + len(pr))
            sub_off = [0]
            for i in touched:
                sub_off.append(sub_off[-1] + len(proofs[i]))
            sub_inst = b"".join(bytes(inst[32 * n_pi * i:32 * n_pi * (i + 1)]) for i in touched)
            sub_ci = None if ci is None else b"".join(bytes(ci[48 * i:48 * i + 48]) for i in touched)
            with oracle_lock:
                want = oracles[name].verify_batch(b"".join(bytes(proofs[i]) for i in touched), sub_off, sub_inst, sub_ci, threads=8)
            exp = list(b.expected)
            for i, v in zip(touched, want):
                exp[i] = int(v)
            return synth.Batch(n=b.n, proofs=b"".join(bytes(pr) for pr in proofs), proof_off=off, instances=bytes(inst), committed=bytes(ci) if ci is not None else None, expected=exp)

        def draw(name, cap=None):
            vk, pl, dp, pool = pools[name]
            r = rng.random()
            n = 1 + int((args.max_n - 1) * r ** 3)          # mostly small, sometimes the full size
            if rng.random() < 0.1:
                n = rng.choice([1, 63, 64, 65, 255, 256, 257, 2047, 2048])
            n = min(n, cap or args.max_n, args.max_n)
            if rng.random() < 0.25:                         # an all-valid batch (the RLC mode's fast path)
                good = [i for i in range(pool.n) if pool.expected[i]]
                order = [rng.choice(good) for _ in range(n)]
                if rng.random() < 0.5 and n > 1:            # ... with exactly one reject somewhere
                    bad = [i for i in range(pool.n) if not pool.expected[i]]
                    order[rng.randrange(n)] = rng.choice(bad)
            else:
                order = [rng.randrange(pool.n) for _ in range(n)]
            return permute(synth, pool, order, vk.n_public_inputs)

        def report(form, name, b, got, extra=""):
            ok = list(got) == b.expected
            with lock:
                stats["calls"] += 1
                stats["proofs"] += b.n
                by_form[form] = by_form.get(form, 0) + 1
                if not ok:
                    stats["bad"] += 1
                    diff = [i for i in range(b.n) if got[i] != b.expected[i]]
                    failures.append("thread %d %s %s n=%d %s: %d wrong verdicts, first at %s" % (tid, form, name, b.n, extra, len(diff), diff[:8]))
                    print("MISMATCH", failures[-1], flush=True)

        up = lambda x: torch.frombuffer(bytearray(x), dtype=torch.uint8).to(dev) if x else None
        ptr = lambda t: t.data_ptr() if t is not None else None
        while time.time() < deadline and not failures:
            name = rng.choice(list(pools))
            vk, pl, dp, pool = pools[name]
            form = rng.choice(args.forms.split(","))
            try:
                if form == "multi":                     # ONE laned workspace for all the circuits (h2v_workspace_create_multi), calls of several plans in flight
                    if "multi" not in wss:
                        wss["multi"] = backend.Workspace.multi([pools[c][2] for c in pools], args.max_n, lanes=rng.choice([0, 3, 8, 16]), chunk=rng.choice([0, 0, 100, 700]))
                        wss["multi"].defer_joins(True)
                    w = wss["multi"]
                    held = []
                    for _ in range(rng.randrange(1, 7)):
                        nm = rng.choice(list(pools))
                        dpm = pools[nm][2]
                        b = draw(nm, cap=2048)
                        d = (up(b.proofs), torch.tensor(b.proof_off, dtype=torch.int64).to(dev), up(b.instances), up(b.committed))
                        acc = torch.full((b.n,), 7, dtype=torch.uint8, device=dev)
                        if rng.random() < 0.5:
                            dpm.verify_batch_rlc_device(b.n, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), acc.data_ptr(), None, ws=w, stream=stream.cuda_stream,
                                                        seed=bytes(rng.randrange(256) for _ in range(32)))
                        else:
                            dpm.verify_batch_device(b.n, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), acc.data_ptr(), None, ws=w, stream=stream.cuda_stream)
                        held.append((nm, b, d, acc))
                    w.join(stream.cuda_stream)
                    stream.synchronize()
                    for nm, b, d, acc in held:
                        report(form, nm, b, acc.cpu().tolist())
                elif form == "churn":                     # a plan and a workspace that live for one call (the registry under concurrency)
                    b = draw(name, cap=300)
                    dp2 = backend.DevicePlan(pl.to_bytes(), 0, fixed_base_window_bits=rng.choice([0, 0, 4, 8, 12]))
                    w2 = backend.Workspace(dp2, b.n) if rng.random() < 0.5 else backend.Workspace(dp2, b.n, lanes=rng.choice([1, 2, 5]), chunk=max(1, b.n // 3))
                    if rng.random() < 0.5:
                        report(form, name, b, dp2.verify_batch(b.proofs, b.proof_off, b.instances, b.committed, ws=w2))
                    else:
                        report(form, name, b, dp2.verify_batch_rlc(b.proofs, b.proof_off, b.instances, b.committed, ws=w2)[0], "rlc")
                    w2.close()
                    dp2.close()
                elif form == "fuzz":
                    b = mutate(name, draw(name, cap=512))
                    w = ws_for(name, rng.choice(["plain", "auto"]), b.n)
                    report(form, name, b, dp.verify_batch(b.proofs, b.proof_off, b.instances, b.committed, ws=w), "per-proof")
                    got, fb = dp.verify_batch_rlc(b.proofs, b.proof_off, b.instances, b.committed, ws=w, seed=bytes(rng.randrange(256) for _ in range(32)))
                    report(form, name, b, got, "rlc fell_back=%s" % fb)
                elif form == "host":
                    b = draw(name)
                    report(form, name, b, dp.verify_batch(b.proofs, b.proof_off, b.instances, b.committed))
                elif form in ("host_ws", "host_laned"):
                    b = draw(name)
                    w = ws_for(name, "plain" if form == "host_ws" else "auto", b.n)
                    report(form, name, b, dp.verify_batch(b.proofs, b.proof_off, b.instances, b.committed, ws=w))
                elif form == "host_rlc":
                    b = draw(name)
                    w = ws_for(name, rng.choice(["plain", "auto"]), b.n)
                    got, fb = dp.verify_batch_rlc(b.proofs, b.proof_off, b.instances, b.committed, ws=w, seed=bytes(rng.randrange(256) for _ in range(32)))
                    report(form, name, b, got, "fell_back=%s" % fb)
                elif form == "submit":
                    depth = rng.choice([1, 2, 4, 7])
                    rlc = rng.random() < 0.5
                    bs = backend.BatchStream(dp, min(args.max_n, 1024), depth, rlc=rlc, seed=bytes(rng.randrange(256) for _ in range(32)) if rlc else None)
                    sent, keep = [], []
                    for _ in range(rng.randrange(1, 3 * depth + 1)):
                        b = draw(name, cap=1024)
                        hb, k_ = dp.host_batch(b.proofs, b.proof_off, b.instances, b.committed)
                        keep.append(k_)
                        sent.append(b)
                        out = bs.push(hb, b.n)
                        if out is not None:
                            report(form, name, sent.pop(0), out[0], "rlc=%s" % rlc)
                    for out in bs.drain():
                        report(form, name, sent.pop(0), out[0], "rlc=%s" % rlc)
                    bs.close()
                else:
                    rlc = form == "device_rlc"
                    calls = rng.randrange(1, 7)
                    batches = [draw(name) for _ in range(calls)]
                    w = ws_for(name, "laned", max(b.n for b in batches))
                    held = []
                    for b in batches:
                        d = (up(b.proofs), torch.tensor(b.proof_off, dtype=torch.int64).to(dev), up(b.instances), up(b.committed))
                        acc = torch.full((b.n,), 7, dtype=torch.uint8, device=dev)
                        st = torch.full((b.n,), -1, dtype=torch.int32, device=dev)
                        if rlc:
                            dp.verify_batch_rlc_device(b.n, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), acc.data_ptr(), st.data_ptr(), ws=w, stream=stream.cuda_stream,
                                                       seed=bytes(rng.randrange(256) for _ in range(32)))
                        else:
                            dp.verify_batch_device(b.n, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), acc.data_ptr(), st.data_ptr(), ws=w, stream=stream.cuda_stream)
                        held.append((b, d, acc, st))
                    w.join(stream.cuda_stream)
                    stream.synchronize()
                    for b, d, acc, st in held:
                        report(form, name, b, acc.cpu().tolist(), "lanes=%s calls=%d" % (w.lanes(), calls))
                        if [int(x == 0) for x in st.cpu().tolist()] != b.expected:
                            with lock:
                                failures.append("thread %d %s %s n=%d: status words disagree with accept" % (tid, form, name, b.n))
            except Exception as e:                           # an error code from the library is a failure of the soak as well
                with lock:
                    failures.append("thread %d %s %s: %r" % (tid, form, name, e))
                    print("ERROR", failures[-1], flush=True)
        for w in wss.values():
            w.close()

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(args.threads)]
    for t in ths:
        t.start()
    last = time.time()
    while any(t.is_alive() for t in ths):
        time.sleep(1.0)
        if time.time() - last > 30:
            last = time.time()
            with lock:
                print("... %d calls, %d proofs, %d mismatching calls" % (stats["calls"], stats["proofs"], stats["bad"]), flush=True)
    for t in ths:
        t.join()
    print("soak: %d calls (%s), %d proofs, %d threads, %.1f min: %s" % (stats["calls"], ", ".join("%s %d" % kv for kv in sorted(by_form.items())), stats["proofs"], args.threads,
                                                                       args.minutes, "FAILED" if failures else "every verdict as constructed"))
    for f in failures:
        print("  ", f)
    backend.shutdown()
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
