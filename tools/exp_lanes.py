#!/usr/bin/env python3
"""Experiment: one laned workspace (library-owned lanes) against the caller-driven pipelining of bench.py.
  python tools/exp_lanes.py [--workload simple_mul] [--mode per-proof|rlc]
Prints one line per configuration: proofs/s and ms per 4096 proofs."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="simple_mul")
    ap.add_argument("--mode", default="per-proof")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--single", default="4096,8192,20480")
    ap.add_argument("--lanes", default="4,5,8")
    ap.add_argument("--chunks", default="0")
    ap.add_argument("--stream-steps", type=int, default=20)
    args = ap.parse_args()
    import torch
    from plutus_halo2_verifier_gen_amd import backend, plan as PL, synth, vk as V
    dev = torch.device("cuda", 0)
    vk, td = V.BUILDERS[args.workload]()
    pl = PL.compile_plan(vk)
    B = args.batch
    nmax = max([int(x) for x in args.single.split(",") if x] + [B])
    base = synth.forge_batch(vk, td, min(nmax, 4096), seed=1000, workers=16, plan=pl)
    reps = -(-nmax // base.n)
    proofs = base.proofs * reps
    off = [0]
    for r in range(reps):
        for i in range(base.n):
            off.append(off[-1] + base.proof_off[i + 1] - base.proof_off[i])
    inst = base.instances * reps
    ci = base.committed * reps if base.committed else None
    dp = backend.DevicePlan(pl.to_bytes(), device=0)
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) if b else torch.zeros(1, dtype=torch.uint8, device=dev)
    d_proofs, d_inst = to_dev(proofs), to_dev(inst)
    d_off = torch.tensor(off, dtype=torch.int64).to(dev)
    d_ci = to_dev(ci) if ci else None
    seed = bytes(range(32))
    rlc = args.mode == "rlc"

    caller = torch.cuda.Stream(device=dev)     # (not the legacy default stream: it synchronises with every blocking stream)
    torch.cuda.set_stream(caller)

    def call(ws, n, acc, st_, stream=None):
        stream = stream or caller.cuda_stream
        ptrs = (n, d_proofs.data_ptr(), d_off.data_ptr(), d_inst.data_ptr(), d_ci.data_ptr() if d_ci is not None else None, acc.data_ptr(), st_.data_ptr())
        if rlc:
            dp.verify_batch_rlc_device(*ptrs, ws=ws, stream=stream, seed=seed)
        else:
            dp.verify_batch_device(*ptrs, ws=ws, stream=stream)

    def single(n, lanes, chunk, reps_=6):
        if lanes < 0:    # whatever h2v_workspace_create returns for a workspace of n proofs (laned from four chunks up)
            ws = backend.Workspace(dp, n)
        else:
            ws = backend.Workspace(dp, n, lanes=lanes, chunk=chunk) if lanes else backend.Workspace.__new__(backend.Workspace)
        if lanes == 0:   # classic, one pipeline
            import ctypes as C
            ws._h = C.c_void_p()
            backend.check(backend.lib().h2v_workspace_create_lanes(dp.handle, n, 1, n, C.byref(ws._h)))
        acc = torch.zeros(n, dtype=torch.uint8, device=dev)
        st_ = torch.zeros(n, dtype=torch.int32, device=dev)
        for _ in range(max(2, lanes, 3)):
            call(ws, n, acc, st_)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps_):
            call(ws, n, acc, st_)
            torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / reps_
        ok = int(acc.sum().item()) == n
        lc = ws.lanes()
        ws.close()
        return el, ok, lc

    for n in [int(x) for x in args.single.split(",") if x]:
        for lanes in [0, -1] + [int(x) for x in args.lanes.split(",") if x]:
            for chunk in [int(x) for x in args.chunks.split(",")]:
                if lanes <= 0 and chunk != int(args.chunks.split(",")[0]):
                    continue
                el, ok, lc = single(n, lanes, chunk)
                print("single call n=%6d lanes=%2d chunk=%5d -> %8.3f ms  %9.0f proofs/s  (%.3f ms per 4096) ok=%s stagger=%s" % (
                    n, lc[0], lc[1], el * 1e3, n / el, el * 1e3 * 4096 / n, ok, os.environ.get("H2V_LANE_STAGGER", "1")), flush=True)

    # a stream of batches of B proofs through ONE laned workspace with deferred joins
    K = args.stream_steps
    if not K:
        return
    for lanes in [int(x) for x in args.lanes.split(",") if x] + ([11] if args.stream_steps else []):
        ws = backend.Workspace(dp, B, lanes=lanes, chunk=B)
        ws.defer_joins(True)
        accs = [torch.zeros(B, dtype=torch.uint8, device=dev) for _ in range(lanes)]
        sts = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(lanes)]
        for k in range(lanes):
            call(ws, B, accs[k % lanes], sts[k % lanes])
        ws.join(caller.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            call(ws, B, accs[k % lanes], sts[k % lanes])
        ws.join(caller.cuda_stream)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / K
        ok = all(int(a.sum().item()) == B for a in accs)
        print("stream of %d x %d, lanes=%2d -> %8.3f ms per batch  %9.0f proofs/s ok=%s" % (K, B, lanes, el * 1e3, B / el, ok), flush=True)
        ws.close()


if __name__ == "__main__":
    main()
