#!/usr/bin/env python3
"""PCIe-inclusive rate of the HOST-BUFFER entry points (what a Rust caller binding include/h2v.h would use), beside the
device-resident step bench.py times.  Never the `value` of the bench line.

  python tools/bench_host_path.py [--mode per-proof|rlc] [--batch 4096] [--steps 40]

The batch lives in host memory the caller owns (ctypes buffers built once, as a Rust Vec<u8> would be); every step hands it
to h2v_verify_batch_submit on one of two workspaces and collects the previous step with h2v_verify_batch_wait, so the
pack + upload of step k+1 runs beside the kernels of step k.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from plutus_halo2_verifier_gen_amd import backend, plan as PL, synth, vk as V  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="per-proof", choices=["per-proof", "rlc"])
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--workspaces", type=int, default=2, help="round 2's form: W ordinary workspaces")
ap.add_argument("--lanes", type=int, default=0, help="ONE laned workspace with this many batches in flight instead (round 3)")
args = ap.parse_args()
vk, td = V.simple_mul_vk()
pl = PL.compile_plan(vk)
b = synth.forge_batch(vk, td, args.batch, seed=1000, plan=pl, workers=16)
dp = backend.DevicePlan(pl.to_bytes(), 0)
W = args.lanes or args.workspaces
if args.lanes:
    one = backend.Workspace(dp, args.batch, lanes=args.lanes, chunk=args.batch)
    wss = [one] * W           # (the same workspace: submit / wait go through its ring of staging slots)
else:
    wss = [backend.Workspace(dp, args.batch) for _ in range(W)]
    for w_ in wss:
        w_.hint_in_flight(W)      # (from 4 up: launch shapes that issue fewer instructions)
hb, keep = dp.host_batch(b.proofs, b.proof_off, b.instances, b.committed)
rlc = args.mode == "rlc"
seed = bytes(range(32))


def run(steps):
    acc = None
    for k in range(steps):
        ws = wss[k % W]
        if k >= W:
            acc, _ = ws.wait(args.batch)          # batch k - W ran on this workspace
        dp.submit(hb, ws, rlc=rlc, seed=seed, one_stream=rlc and W >= 3)
    for k in range(max(0, steps - W), steps):
        acc, _ = wss[k % W].wait(args.batch)
    return acc


run(max(6, W))
t0 = time.perf_counter()
acc = run(args.steps)
dt = (time.perf_counter() - t0) / args.steps
# the blocking single call, for comparison
t0 = time.perf_counter()
for _ in range(10):
    one_acc = dp.verify_batch_rlc(b.proofs, b.proof_off, b.instances, b.committed, ws=wss[0], seed=seed)[0] if rlc else \
        dp.verify_batch(b.proofs, b.proof_off, b.instances, b.committed, ws=wss[0])
dt1 = (time.perf_counter() - t0) / 10
print(json.dumps({"what": "host-buffer path (PCIe-inclusive): h2v_verify_batch_submit / _wait, %s" % (
                      "ONE laned workspace, %d batches in flight" % W if args.lanes else "%d workspaces" % W), "mode": args.mode,
                  "proofs_per_step": args.batch, "ms_per_step": round(dt * 1e3, 4), "proofs_per_s": round(args.batch / dt, 1),
                  "all_accepted": sum(acc) == args.batch,
                  "blocking_call_ms_incl_python_marshalling": round(dt1 * 1e3, 4), "blocking_call_all_accepted": sum(one_acc) == args.batch}))
