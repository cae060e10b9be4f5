#!/usr/bin/env python3
"""Experiment: does a fresh laned workspace run slower for its first second(s)?  Repeated timed rounds (3 x depth calls each) on
atms_with_lookups x 2048 right after the workspace is created, with and without another plan's workspace alive."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from plutus_halo2_verifier_gen_amd import backend, plan as PL, synth, vk as V

dev = torch.device("cuda", 0)
n = 2048
parts = {}
for k, name in enumerate(("lookup_table", "atms_with_lookups")):
    vk, td = V.BUILDERS[name]()
    pl = PL.compile_plan(vk)
    dp = backend.DevicePlan(pl.to_bytes(), 0)
    b = synth.forge_batch(vk, td, n, seed=500 + k, plan=pl, workers=8)
    up = lambda x: torch.frombuffer(bytearray(x), dtype=torch.uint8).to(dev) if x else None
    parts[name] = (dp, (up(b.proofs), torch.tensor(b.proof_off, dtype=torch.int64).to(dev), up(b.instances), up(b.committed)))
ptr = lambda t: t.data_ptr() if t is not None else None
s = torch.cuda.Stream(device=dev)
accs = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(16)]
T0 = time.perf_counter()


def rounds(name, ws, k, label):
    dp, d = parts[name]
    depth = ws.depth(n)
    out = []
    for r in range(k):
        t0 = time.perf_counter()
        for _ in range(3):
            for a in accs[:depth]:
                dp.verify_batch_device(n, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), a.data_ptr(), None, ws=ws, stream=s.cuda_stream)
        ws.join(s.cuda_stream)
        s.synchronize()
        out.append((time.perf_counter() - t0) / (3 * depth) * 1e3)
    print("%-46s t=%6.2fs  ms per call: %s" % (label, time.perf_counter() - T0, " ".join("%.2f" % x for x in out)), flush=True)


def mk(name):
    w = backend.Workspace(parts[name][0], n, lanes=0, chunk=0)
    w.defer_joins(True)
    return w


def tune(name, ws, label):
    dp, d = parts[name]
    r = ws.tune(dp, n, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), s.cuda_stream)
    print("%-46s t=%6.2fs  tune: default %.2f best %.2f engine %d tpl %d (%d measured)" % (label, time.perf_counter() - T0, r.default_ms, r.best_ms, r.pairing_engine, r.msm_terms_per_lane, r.n_measured), flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "tune":
    l = mk("lookup_table")
    tune("lookup_table", l, "lookup: tune on a fresh workspace")
    rounds("lookup_table", l, 4, "lookup after its tune")
    a = mk("atms_with_lookups")
    tune("atms_with_lookups", a, "atms: tune on a fresh workspace (lookup's alive)")
    rounds("atms_with_lookups", a, 8, "atms after its tune")
    tune("atms_with_lookups", a, "atms: tune again")
    rounds("atms_with_lookups", a, 4, "atms after the second tune")
    l.close()
    tune("atms_with_lookups", a, "atms: tune, lookup's workspace closed")
    a.close()
    a = mk("atms_with_lookups")
    rounds("atms_with_lookups", a, 4, "atms, fresh workspace, untuned")
    tune("atms_with_lookups", a, "atms: tune on that one")
    backend.shutdown()
    sys.exit(0)
a = mk("atms_with_lookups")
rounds("atms_with_lookups", a, 12, "atms, fresh workspace, nothing else alive")
time.sleep(3.0)
rounds("atms_with_lookups", a, 6, "atms, same workspace after 3 s of idle GPU")
l = mk("lookup_table")
rounds("lookup_table", l, 6, "lookup, fresh workspace (atms's alive)")
rounds("atms_with_lookups", a, 6, "atms again (lookup's alive and warm)")
a2 = mk("atms_with_lookups")
rounds("atms_with_lookups", a2, 8, "atms, SECOND fresh workspace (two others alive)")
a.close(); l.close()
rounds("atms_with_lookups", a2, 6, "atms second workspace, the others closed")
backend.shutdown()
