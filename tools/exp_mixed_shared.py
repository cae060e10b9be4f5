#!/usr/bin/env python3
"""Experiment: BASELINE configs[2] (lookup_table x 2048 + atms_with_lookups x 2048 per step) on two laned workspaces (one per
plan) against ONE laned workspace serving both plans (h2v_workspace_create_multi).  Prints ms per step for both."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from plutus_halo2_verifier_gen_amd import backend, plan as PL, synth, vk as V

dev = torch.device("cuda", 0)
n, steps = 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 120
parts = []
for k, name in enumerate(("lookup_table", "atms_with_lookups")):
    vk, td = V.BUILDERS[name]()
    pl = PL.compile_plan(vk)
    dp = backend.DevicePlan(pl.to_bytes(), 0)
    b = synth.forge_batch(vk, td, n, seed=500 + k, plan=pl, workers=8)
    up = lambda x: torch.frombuffer(bytearray(x), dtype=torch.uint8).to(dev) if x else None
    d = (up(b.proofs), torch.tensor(b.proof_off, dtype=torch.int64).to(dev), up(b.instances), up(b.committed))
    parts.append((name, pl, dp, d))
ptr = lambda t: t.data_ptr() if t is not None else None
s = torch.cuda.Stream(device=dev)
accs = [[torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(16)] for _ in parts]


def run(wss, label):
    def step(k):
        for pi, ((name, pl, dp, d), ws) in enumerate(zip(parts, wss)):
            dp.verify_batch_device(n, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), accs[pi][k % 16].data_ptr(), None, ws=ws, stream=s.cuda_stream)
    for k in range(16):
        step(k)
    for ws in set(wss):
        ws.join(s.cuda_stream)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        for ws in set(wss):
            ws.join(s.cuda_stream)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ok = all(bool(a.all().item()) for row in accs for a in row)
        print("%-34s %.4f ms per step (2 x %d proofs), %.0f proofs/s, verdicts ok: %s" % (label, 1e3 * el / steps, n, 2 * n * steps / el, ok), flush=True)


two = [backend.Workspace(dp, n, lanes=0, chunk=0) for (name, pl, dp, d) in parts]
for w in two:
    w.defer_joins(True)
run(two, "two workspaces (8 lanes each)")
for w in two:
    w.close()
for lanes in (0, 16):
    one = backend.Workspace.multi([p[2] for p in parts], n, lanes=lanes, chunk=0)
    one.defer_joins(True)
    run([one, one], "one multi-plan workspace, lanes=%d" % lanes)
    one.close()
backend.shutdown()
