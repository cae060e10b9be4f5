#!/usr/bin/env python3
"""Benchmark of the Halo2/KZG verification hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch of synthetic proofs per GPU, every input already resident in HBM:
transcript replay + Fr combiner, G1 decompression, then
  --mode per-proof (default; BASELINE configs[1] "G1 MSM + one pairing per proof"): per-proof G1 MSM + fused pairing check;
  --mode rlc: the batch-accept fast path (h2v_verify_batch_rlc_device): one bucketed Pippenger G1 MSM over every per-proof
              point of the batch + ONE pairing (+ the per-proof kernels, skipped on the device unless the batch check fails).
Workload: BASELINE.json configs[1] - simple_mul, 4096 proofs per GPU; --workload / --batch select the other configs.
--scaling weak (default): every rank verifies --batch proofs; strong: the ranks split ONE batch of --batch proofs by
contiguous index ranges (shard.shard_range).  Either way there is no data-path collective; the accept bytes of all timed
steps are gathered on rank 0 over RCCL (one collective after the last step, inside the timed region).
--pipeline lanes (default): every step is ONE call on ONE laned workspace (h2v_workspace_create_lanes + deferred joins):
the library keeps the steps in flight on its own lanes and streams; the caller owns one workspace and one stream and sets
no environment variable.  --pipeline streams: round 2's caller-driven form (--inflight P workspaces on P torch streams),
kept as a probe.  Every step runs completely inside the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# Integer-multiply issue ceiling: MEASURED on this chip with tools/ubench/imad.hip
# (hipcc --offload-arch=gfx950 -O3 tools/ubench/imad.hip -o imad && ./imad > profiles/rNN_imad_ubench.txt); the kept
# stdout is read here: cycles per v_mad_u64_u32 wave-instruction per SIMD at 4 waves per SIMD (2.4 GHz, 1024 SIMDs).
# No file, no peak: int_roofline.peak / frac are null rather than a remembered constant.


def imad_costs():
    """(cycles per v_mad_u64_u32 wave-instruction per SIMD, cycles per plain 32-bit VALU instruction, file) from the newest
    profiles/r*_imad_ubench.txt: the 4-waves-per-SIMD section; the ubench's per-SIMD reading (v2: s_memtime stamps grouped by
    physical SIMD) when the file has it, else the event-time reading of the old format."""
    import glob

    def version_key(path):
        m = re.match(r"r(\d+)_", os.path.basename(path))
        return int(m.group(1)) if m else -1

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_imad_ubench.txt")), key=version_key)
    if not files:
        return None, None, None
    got, sect = {}, None
    for line in open(files[-1]):
        m = re.match(r"--- (\d+) wave", line)
        if m:
            sect = int(m.group(1))
        m = re.match(r"(mad_u64_u32|add_u32) .*?~([0-9.]+) cycles", line)
        if m and sect == 4:
            m2 = re.search(r"per SIMD that held \d+ waves: ([0-9.]+) cycles", line)
            got[m.group(1)] = float(m2.group(1)) if m2 else float(m.group(2))
    return got.get("mad_u64_u32"), got.get("add_u32"), os.path.basename(files[-1])


def imad_peak():
    cyc, _c_min, src = imad_costs()
    if not cyc:
        return None, None
    return 1024 * 64 * 2.4e9 / cyc / 1e12, src


IMAD_PEAK_TOPS, IMAD_PEAK_SOURCE = imad_peak()
# v_mad_u64_u32 per field operation (csrc/h2v_field.hpp): product-scanning multiply 392, square 301
MAD_MUL, MAD_SQR = 392, 301
MAD_DBL = 2 * MAD_MUL + 5 * MAD_SQR          # dbl-2009-l
MAD_MADD = 8 * MAD_MUL + 3 * MAD_SQR         # mixed addition
MAD_ADD = 12 * MAD_MUL + 4 * MAD_SQR         # full Jacobian addition

WORKLOADS = {
    # name: ([(vk builder, proofs per GPU), ...], BASELINE config label).  A workload of several parts is a MIXED batch: one
    # step = one call per part (its own plan, its own laned workspace), all on the one caller stream; proofs/s over all parts
    "simple_mul": ([("simple_mul", 4096)], "simple_mul x4096 per GPU (BASELINE configs[1])"),
    "lookup_atms_mixed": ([("lookup_table", 2048), ("atms_with_lookups", 2048)],
                          "lookup_table x2048 + atms_with_lookups x2048, one mixed batch per step (BASELINE configs[2])"),
    "lookup_mixed": ([("lookup_table", 2048)], "lookup_table x2048 (half of BASELINE configs[2])"),
    "atms_with_lookups": ([("atms_with_lookups", 2048)], "atms_with_lookups x2048 (half of BASELINE configs[2])"),
    "sha256": ([("sha256", 1024)], "sha256-shaped x1024 (BASELINE configs[3])"),
    "secp256k1": ([("secp256k1", 512)], "secp256k1-shaped x512 (BASELINE configs[4])"),
    "ivc": ([("ivc", 1024)], "IVC-shaped recursive circuit x1024 per GPU (accumulator fold, DESIGN.md section 10)"),
}


def pmc_table(workload, batch, mode):
    """Counters per launch from the newest committed PMC summary that was collected on THIS workload / batch / mode (header
    line `# workload=... batch=... mode=... [pmc_steps=N]`, written by tools/scripts/profile_round.sh over separate rocprofv3
    --pmc passes of the one-step-at-a-time form of this command): ({kernel: {counter: (launches, mean per launch)}}, header
    dict, file name) or (None, None, None)."""
    import glob

    def version_key(path):
        m = re.match(r"r(\d+)_v(\d+)", os.path.basename(path))
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.txt")), key=version_key, reverse=True):
        lines = open(path).read().splitlines()
        if not lines or not lines[0].startswith("#"):
            continue
        hdr = dict(kv.split("=", 1) for kv in lines[0][1:].split() if "=" in kv)
        if hdr.get("workload") != workload or hdr.get("batch") != str(batch) or hdr.get("mode") != mode:
            continue
        tab = {}
        for line in lines[1:]:
            f = line.split()
            if len(f) >= 4 and f[0].startswith("k_"):
                try:
                    tab.setdefault(f[0], {})[f[1]] = (int(f[2]), float(f[3]))
                except ValueError:
                    pass
        if tab:
            return tab, hdr, os.path.basename(path)
    return None, None, None


def pmc_traffic(tab, kernel):
    """HBM-side bytes per launch of `kernel` from a pmc_table: MI355X_MICROARCH.md, HBM / rocprofv3 - FETCH_SIZE and WRITE_SIZE
    are KiB on gfx950, and FETCH_SIZE tallies 128-byte requests at 64 bytes ("double it before comparing with a byte count");
    WRITE_SIZE is exact.  (corrected, uncorrected) or (None, None)."""
    c = (tab or {}).get(kernel, {})
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None, None
    f, w = c["FETCH_SIZE"][1], c["WRITE_SIZE"][1]
    return int((2 * f + w) * 1024), int((f + w) * 1024)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240, help="timed steps (default 240 = about a second: a batch spends ~30 ms in the pipeline and the timed region ends with "
                                                         "draining it - 60 steps read 3-7 %% low, 3000 steps 1.11 M / 3.05 M proofs/s per proof / RLC against 1.09 / 3.01 M at 240)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="simple_mul", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="proofs per GPU (weak) / in total (strong) - of EVERY part of a mixed workload; default: the workload's BASELINE size")
    ap.add_argument("--mode", default="per-proof", choices=["per-proof", "rlc"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--pipeline", default="lanes", choices=["lanes", "streams"],
                    help="lanes: one laned workspace per plan, the library pipelines the steps (default); streams: --inflight workspaces on torch streams")
    ap.add_argument("--lanes", type=int, default=0, help="--pipeline lanes: number of lanes (0: the library's choice)")
    ap.add_argument("--chunk", type=int, default=0, help="--pipeline lanes: proofs per lane launch (0: the library's choice for the plan); calls of at most half of it are coalesced")
    ap.add_argument("--inflight", type=int, default=0, help="--pipeline streams: steps in flight (workspaces / streams); default: probed")
    ap.add_argument("--reject-fraction", type=float, default=0.0, help="timed steps run on a batch in which this fraction of the proofs has the reference example's byte flip")
    ap.add_argument("--reject-count", type=int, default=0, help="timed steps run on a batch with exactly this many corrupted proofs (pairing-only rejects: wrong_pi)")
    ap.add_argument("--msm-tpl", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="per-proof MSM: terms per lane (h2v_workspace_set_option; 2 .. 4 share the doublings of a lane's terms)")
    ap.add_argument("--pairing", type=int, default=0, choices=[0, 1, 6, 12, 16, 32, 64], help="pairing engine: lanes per proof (h2v_workspace_set_option; 0: the launcher's choice)")
    ap.add_argument("--shared-workspace", action="store_true", help="mixed workloads, --pipeline lanes: ONE laned workspace for all the parts' plans (h2v_workspace_create_multi) instead of one per plan")
    ap.add_argument("--no-tune", action="store_true", help="skip h2v_workspace_tune (the untimed measurement of the candidate launch shapes on this batch before the "
                                                         "warm-up steps); the launcher's own thresholds then decide")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product path); gloo = rehearsal of the N > 1 code path on a box with one GPU "
                         "(every rank on the device H2V_BENCH_DEVICE names, accept bytes gathered through host memory)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: leave the accept gather out (rehearsals: what the collective costs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rlc-secondary", action="store_true", help="per-proof runs: skip the extra measurement of the RLC mode")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="proofs the CPU baseline verifies (2048 x ~9 ms of one core each = ~20 core-seconds, ~1.2 s on the 16 threads)")
    ap.add_argument("--hint", type=int, default=0, help="h2v_workspace_hint_in_flight value (default: the steps in flight); the counter passes of "
                                                     "tools/scripts/profile_round.sh run one step at a time with the shapes of the timed run")
    ap.add_argument("--no-alone", action="store_true", help="skip the one-step-at-a-time pass (with --timed-only: the profiled command launches the timed steps only)")
    ap.add_argument("--timed-only", action="store_true",
                    help="launch nothing but warm-up + the timed steps (no in-flight probe, no one-step pass, no RLC secondary, no reject "
                         "dataset, no CPU baseline): the form profiled under rocprofv3, whose per-kernel averages then cover the same "
                         "launches as the line's kernel_ms")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    parts_spec, label = WORKLOADS[args.workload]
    # --pipeline streams: the steps-in-flight count is PROBED (a few untimed steps with each candidate, the best one is used
    # for the timed region and reported); --pipeline lanes: the library's lanes are the steps in flight.
    small = (args.batch or parts_spec[0][1]) <= 1024
    inflight_candidates = [args.inflight] if args.inflight else ([11, 7, 5, 3, 1] if args.mode == "rlc" else [4, 2, 1] if small else [5, 3, 2, 1])
    inflight = inflight_candidates[0]
    # (round 2 exported GPU_MAX_HW_QUEUES=16 here; the library's streams now get hardware queues of their own - csrc: make_stream)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    if os.environ.get("H2V_BENCH_DEVICE") is not None:   # rehearsal on a one-GPU box: all ranks share that device
        local_rank = int(os.environ["H2V_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)          # one process per GPU; bind before the communicator is created
    dev = torch.device("cuda", local_rank)
    ranks_seen, backend_seen = 1, None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)  # RCCL on ROCm
        else:
            dist.init_process_group(backend="gloo")
        ranks_seen, backend_seen = dist.get_world_size(), str(dist.get_backend())   # what the communicator itself says
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    from plutus_halo2_verifier_gen_amd import backend, plan as PL, shard, synth, vk as V

    ncpu = os.cpu_count() or 1
    workers = max(1, min(16, ncpu // max(1, world)))

    def to_dev(b):
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) if b else torch.zeros(1, dtype=torch.uint8, device=dev)

    class Part:
        """one plan of the workload with this rank's share of its batch, resident in HBM"""

    parts, t_forge = [], 0.0
    for pi, (vk_name, default_batch) in enumerate(parts_spec):
        P = Part()
        P.name, P.B_arg = vk_name, args.batch or default_batch
        if args.scaling == "strong":
            lo, hi = shard.shard_range(P.B_arg, rank, world)   # this rank's contiguous range of the ONE batch
            P.B, P.B_total = hi - lo, P.B_arg
        else:
            P.B, P.B_total = P.B_arg, P.B_arg * world
        if P.B == 0:
            raise SystemExit("strong scaling: more ranks than proofs")
        P.vk, P.td = V.BUILDERS[vk_name]()
        P.pl = PL.compile_plan(P.vk)
        t0 = time.time()
        P.batch = synth.forge_batch(P.vk, P.td, P.B, seed=1000 + rank + 100 * pi, workers=workers, plan=P.pl)
        t_forge += time.time() - t0
        P.timed_batch, P.expected = P.batch, None
        if args.reject_fraction > 0 or args.reject_count > 0:
            # timed reject scenario: the steps run on a batch that holds corrupted proofs (RLC mode: the batch check fails and the
            # fall-back decides).  --reject-count: exactly that many proofs with a wrong pi (only the pairing catches them);
            # --reject-fraction: the reference example's byte flip (examples/simple_mul.rs:87-95)
            import random as _random
            n_pi = P.vk.n_public_inputs
            if args.reject_count > 0:
                rng = _random.Random(4242 + rank + pi)
                bad = set(rng.sample(range(P.B), min(args.reject_count, P.B)))
                ps, ins, exp = [], [], [1] * P.B
                for i in range(P.B):
                    pr, it = P.batch.proof(i), P.batch.instances[32 * n_pi * i:32 * n_pi * (i + 1)]
                    if i in bad:
                        pr, it = synth.corrupt(P.pl, pr, it, "wrong_pi", rng)
                        exp[i] = 0
                    ps.append(pr); ins.append(it)
                off_ = [0]
                for pr in ps:
                    off_.append(off_[-1] + len(pr))
                P.timed_batch = synth.Batch(n=P.B, proofs=b"".join(ps), proof_off=off_, instances=b"".join(ins), committed=P.batch.committed, expected=exp)
            else:
                P.timed_batch = synth.with_rejects(P.pl, P.batch, n_pi, fraction=args.reject_fraction, seed=78 + rank + pi, kinds=["flip_first_scalar"])
            P.expected = list(P.timed_batch.expected)
        P.dp = backend.DevicePlan(P.pl.to_bytes(), device=local_rank)
        P.d_proofs = to_dev(P.timed_batch.proofs)
        P.d_off = torch.tensor(P.timed_batch.proof_off, dtype=torch.int64).to(dev)
        P.d_inst = to_dev(P.timed_batch.instances)
        P.d_ci = to_dev(P.timed_batch.committed) if P.timed_batch.committed else None
        parts.append(P)
    # the accept / status bytes of a step: the parts' vectors one after the other
    off_b, B = [], 0
    for P in parts:
        P.off = B
        B += P.B
    B_total = sum(P.B_total for P in parts)
    timed_expected = None if all(P.expected is None for P in parts) else sum(([1] * P.B if P.expected is None else P.expected for P in parts), [])
    recursive = any(P.vk.recursion_vks is not None for P in parts)

    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")   # where the collectives' tensors live
    Bmax = B
    if world > 1:
        t = torch.tensor([B], device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        Bmax = int(t.item())
    rlc_seed = bytes((7 * k + 1) & 0xff for k in range(32))   # fixed for the timed steps (reproducible); a service draws it per batch

    stream_pool = []
    coalesce_state = []                      # per part: calls of the timed run that share one launch per kernel (1: none)
    caller = torch.cuda.Stream(device=dev)   # --pipeline lanes: the ONE stream the caller submits on (not the legacy NULL stream)
    RING = 16                                # accept / status buffers the steps cycle through (>= lanes: a lane runs its chunks in order)

    def part_ptrs(P, acc, st_):
        return (P.B, P.d_proofs.data_ptr(), P.d_off.data_ptr(), P.d_inst.data_ptr(), P.d_ci.data_ptr() if P.d_ci is not None else None,
                acc.data_ptr() + P.off, st_.data_ptr() + 4 * P.off)

    def apply_options(ws, P):
        """forced shapes (--msm-tpl / --pairing), else what h2v_workspace_tune chose for this part's plan in the timed run"""
        tpl, eng = args.msm_tpl or getattr(P, "tuned", (0, 0))[1], args.pairing or getattr(P, "tuned", (0, 0))[0]
        ws.set_option(backend.Workspace.OPT_MSM_TERMS_PER_LANE, tpl)
        ws.set_option(backend.Workspace.OPT_PAIRING_ENGINE, eng)

    class Run:
        """what a timed run leaves behind: elapsed seconds, the last step's accept bytes, per-step kernel timings per part"""
        def __init__(self, el, accept, timings, rlc_result, close, in_flight, all_steps_ok):
            self.el, self.accept, self.timings, self.rlc_result, self.close, self.in_flight, self.all_steps_ok = el, accept, timings, rlc_result, close, in_flight, all_steps_ok

    def finish_timing(t0):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    def gather_all(d_accepts, steps, n_slots):
        """ONE collective for the accept bytes of all timed steps (rank r sends steps x Bmax bytes), after the last step: a
        gather per step inside the loop would serialise the steps in flight on the host."""
        k_keep = min(steps, n_slots)
        send = torch.zeros(k_keep, Bmax, dtype=torch.uint8, device=dev)
        for j in range(k_keep):
            send[j, :B].copy_(d_accepts[(steps - 1 - j) % n_slots])
        send = send.to(cdev)
        recv = [torch.zeros(k_keep, Bmax, dtype=torch.uint8, device=cdev) for _ in range(world)] if rank == 0 else None
        dist.gather(send, recv, dst=0)
        return recv

    def check_gathered(recv):
        def size_of(r):
            return sum(shard.shard_range(P.B_arg, r, world)[1] - shard.shard_range(P.B_arg, r, world)[0] if args.scaling == "strong" else P.B_arg for P in parts)
        gather_state["ok"] = all(bool(recv[r][:, :size_of(r)].all().item()) for r in range(world))

    def expected_ok(acc_np):
        return bool((acc_np == 1).all()) if timed_expected is None else [int(x) for x in acc_np] == timed_expected

    def lanes_run(mode, steps, warmup, gather, lanes=None):
        """--pipeline lanes: every step is one call per part on that part's ONE laned workspace; joins are deferred, so the
        library keeps as many steps in flight as it has lanes.  One workspace per plan, one caller stream."""
        shared = args.shared_workspace and len(parts) > 1
        if shared:
            one = backend.Workspace.multi([P.dp for P in parts], max(P.B for P in parts), lanes=args.lanes if lanes is None else lanes, chunk=0)
            wss = [one] * len(parts)
        else:
            wss = [backend.Workspace(P.dp, max(P.B, args.chunk), lanes=args.lanes if lanes is None else lanes, chunk=args.chunk) for P in parts]
        for ws, P in zip(wss, parts):
            ws.defer_joins(True)
            apply_options(ws, P)
            if args.hint:
                ws.hint_in_flight(args.hint)
        n_lanes = max(ws.lanes()[0] for ws in wss)
        # (include/h2v.h, COALESCING: per-proof calls of at most half a chunk on a deferring laned workspace run gathered)
        coalesce_state[:] = [ws.lanes()[1] // P.B if mode == "per-proof" and 2 * P.B <= ws.lanes()[1] else 1 for ws, P in zip(wss, parts)]
        in_flight = min(ws.depth(P.B, mode == "rlc") for ws, P in zip(wss, parts))
        # (--timed-only: the profiled form launches warm-up + timed steps and nothing else - the tuner's candidate engines would show
        #  up in the kernel statistics; profile_round.sh forces the shapes the untimed first run chose instead)
        if mode == "per-proof" and not args.no_tune and not args.timed_only and not args.msm_tpl and not args.pairing and not any(hasattr(P, "tuned") for P in parts):
            # untimed: the library measures its candidate launch shapes on this very batch, in this very regime (all lanes busy)
            tuned_state.clear()
            for P, ws in zip(parts, wss):
                r_ = ws.tune(P.dp, P.B, P.d_proofs.data_ptr(), P.d_off.data_ptr(), P.d_inst.data_ptr(), P.d_ci.data_ptr() if P.d_ci is not None else None, caller.cuda_stream)
                P.tuned = (r_.pairing_engine, r_.msm_terms_per_lane)
                tuned_state.append({"circuit": P.name, "pairing_engine": r_.pairing_engine, "msm_terms_per_lane": r_.msm_terms_per_lane, "configurations_measured": r_.n_measured,
                                    "ms_per_call_launchers_rule": round(r_.default_ms, 4), "ms_per_call_chosen": round(r_.best_ms, 4)})
        d_accepts = [torch.zeros(B, dtype=torch.uint8, device=dev) for _ in range(RING)]
        d_statuses = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(RING)]
        cs = caller.cuda_stream
        torch.cuda.synchronize()

        def step(k):
            for P, ws in zip(parts, wss):
                if mode == "rlc":
                    P.dp.verify_batch_rlc_device(*part_ptrs(P, d_accepts[k % RING], d_statuses[k % RING]), ws=ws, stream=cs, seed=rlc_seed)
                else:
                    P.dp.verify_batch_device(*part_ptrs(P, d_accepts[k % RING], d_statuses[k % RING]), ws=ws, stream=cs)

        for k in range(max(warmup, n_lanes)):   # untimed; at least one step on every lane (a lane's first use creates it)
            step(k)
        for ws in set(wss):
            ws.join(cs)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        for ws in set(wss):
            ws.join(cs)
        recv = None
        if gather and world > 1:
            with torch.cuda.stream(caller):
                recv = gather_all(d_accepts, steps, RING)
        el = finish_timing(t0)
        if recv is not None and rank == 0:
            check_gathered(recv)
        ok = all(expected_ok(d_accepts[(steps - 1 - j) % RING].cpu().numpy()) for j in range(min(steps, RING)))

        def close():
            for ws in set(wss):
                ws.close()
        nP = len(parts)
        back = (lambda j, p: j * nP + (nP - 1 - p)) if shared else (lambda j, p: j)     # (a shared workspace sees the parts' calls interleaved)
        return Run(el, d_accepts[(steps - 1) % RING].cpu().numpy(), lambda j, p: wss[p].timings(back(j, p)), lambda j, p: wss[p].rlc_result(calls_back=back(j, p)), close, in_flight, ok)

    def streams_run(mode, inflight, steps, warmup, gather, sync_every_step=False):
        """--pipeline streams (round 2): `inflight` workspaces per part on `inflight` torch streams, driven from here"""
        wss = [[backend.Workspace(P.dp, P.B) for P in parts] for _ in range(inflight)]
        for row in wss:
            for w_, P in zip(row, parts):
                w_.hint_in_flight(args.hint or inflight)     # (from 4 up the library prefers launch shapes that issue fewer instructions)
                apply_options(w_, P)
        # (the same few torch streams in every measurement of this process: every stream that was ever created keeps a
        #  hardware queue busy in the runtime's round-robin, and later measurements would collide with the earlier ones')
        while len(stream_pool) < inflight:
            stream_pool.append(torch.cuda.Stream(device=dev))
        streams = stream_pool[:inflight] if inflight > 1 else [None]
        d_accepts = [torch.zeros(B, dtype=torch.uint8, device=dev) for _ in range(inflight)]
        d_statuses = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(inflight)]
        torch.cuda.synchronize()

        def step(k):
            slot = k % inflight
            st = streams[slot]
            ctx = torch.cuda.stream(st) if st is not None else None
            if ctx is not None:
                ctx.__enter__()
            try:
                stream = torch.cuda.current_stream().cuda_stream
                for pi, P in enumerate(parts):
                    if mode == "rlc":
                        P.dp.verify_batch_rlc_device(*part_ptrs(P, d_accepts[slot], d_statuses[slot]), ws=wss[slot][pi], stream=stream, seed=rlc_seed, one_stream=inflight >= 3)
                    else:
                        P.dp.verify_batch_device(*part_ptrs(P, d_accepts[slot], d_statuses[slot]), ws=wss[slot][pi], stream=stream)
                    if sync_every_step:
                        torch.cuda.synchronize()     # (one call at a time: each part's kernels alone on the chip)
            finally:
                if ctx is not None:
                    ctx.__exit__(None, None, None)

        for k in range(max(warmup, inflight)):   # untimed; at least one step on every workspace (their first use allocates)
            step(k)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        recv = None
        if gather and world > 1:
            torch.cuda.synchronize()
            recv = gather_all(d_accepts, steps, inflight)
        el = finish_timing(t0)
        if recv is not None and rank == 0:
            check_gathered(recv)

        def close():
            for row in wss:
                for w_ in row:
                    w_.close()
        ok = all(expected_ok(d_accepts[(steps - 1 - j) % inflight].cpu().numpy()) for j in range(min(steps, inflight)))
        return Run(el, d_accepts[(steps - 1) % inflight].cpu().numpy(),
                   lambda j, p: wss[(steps - 1 - j) % inflight][p].timings(j // inflight),
                   lambda j, p: wss[(steps - 1 - j) % inflight][p].rlc_result(calls_back=j // inflight), close, inflight, ok)

    def pick_inflight(mode, candidates):
        """a few untimed steps with each candidate number of steps in flight; returns (best, {candidate: ms per step})"""
        if len(candidates) == 1:
            return candidates[0], None
        seen = {}
        for cand in candidates:
            k = args.steps            # as many steps as the timed region: filling and draining the pipeline count the same
            r_ = streams_run(mode, cand, k, min(args.warmup, 3), False)
            r_.close()
            seen[cand] = round(r_.el / k * 1e3, 4)
        best = min(seen, key=seen.get)
        if world > 1:   # every rank the same count (the slowest rank's view decides nothing: take rank 0's)
            t = torch.tensor([best], device=cdev)
            dist.broadcast(t, src=0)
            best = int(t.item())
        return best, seen

    PP_KEYS = ["transcript_combiner", "g1_decompress", "g1_msm", "g1_msm_fixed", "pairing"]
    RLC_KEYS = ["transcript_combiner", "g1_decompress", "rlc_prepare", "bucket_sort", "bucket_accumulate", "bucket_reduce", "pairing"]

    def kernel_times(run, mode, k_steps, p):
        """averages of the event-timed kernel durations of part p over the last k_steps steps of a run"""
        info = {}
        if mode == "rlc":
            # steps that the workspace ROUTED to the per-proof kernels (include/h2v.h: ROUTING) have no batch-check record
            routed = [j for j in range(k_steps) if run.rlc_result(j, p)[1].total_ms == 0 and not run.rlc_result(j, p)[0]]
            routed_state[p] = len(routed) / float(k_steps)
            if len(routed) * 2 > k_steps:
                return kernel_times(run, "per-proof", k_steps, p)
        if mode == "rlc":
            acc = {k: 0.0 for k in RLC_KEYS}
            span, all_ok = 0.0, True
            steps_rlc = [j for j in range(k_steps) if j not in routed]
            k_steps = len(steps_rlc)
            for j in steps_rlc:
                ok, tm = run.rlc_result(j, p)
                all_ok = all_ok and ok
                for nm, v in zip(RLC_KEYS, [tm.transcript_combiner_ms, tm.g1_decompress_ms, tm.prepare_ms, tm.bucket_sort_ms,
                                            tm.bucket_accumulate_ms, tm.bucket_reduce_ms, tm.pairing_ms]):
                    acc[nm] += v
                span += tm.total_ms
                info = {"msm_terms": tm.msm_terms, "window_bits": tm.window_bits, "windows_per_glv_half": tm.windows, "max_entries_per_lane": tm.max_chain}
            return {k: v / k_steps for k, v in acc.items()}, span / k_steps, dict(launches=1, msm_lpt=0, pair_lanes=64, rlc_shape=info, all_batch_ok=all_ok, var_lpt=0)
        acc = {k: 0.0 for k in PP_KEYS}
        span, launches, msm_lpt, pair_lanes, var_lpt = 0.0, 1, 2, 32, 0
        for j in range(k_steps):
            tm = run.timings(j, p)
            launches = max(1, tm.launches)
            msm_lpt = tm.msm_lanes_per_term or 2
            var_lpt = tm.msm_var_lanes_per_term
            pair_lanes = tm.pairing_lanes_per_proof or 32
            for nm, v in zip(PP_KEYS, [tm.transcript_combiner_ms, tm.g1_decompress_ms, tm.g1_msm_ms, tm.g1_msm_fixed_ms, tm.pairing_ms]):
                acc[nm] += v
            span += tm.total_ms
        if msm_lpt != 3:
            acc.pop("g1_msm_fixed")     # (one MSM kernel: no fixed-base launch beside it)
        return {k: v / k_steps for k, v in acc.items()}, span / k_steps, dict(launches=launches, msm_lpt=msm_lpt, pair_lanes=pair_lanes, rlc_shape=None,
                                                                                all_batch_ok=None, var_lpt=var_lpt)

    gather_state = {"ok": None}
    tuned_state = []
    routed_state = {}
    if args.timed_only:
        args.no_cpu_baseline = args.no_rlc_secondary = True
        inflight_candidates = inflight_candidates[:1]
    inflight_probe = None
    if args.pipeline == "streams":
        inflight, inflight_probe = pick_inflight(args.mode, inflight_candidates)
        run = streams_run(args.mode, inflight, args.steps, args.warmup, not args.no_gather)
    else:
        run = lanes_run(args.mode, args.steps, args.warmup, not args.no_gather)
        inflight = run.in_flight
    co_main = list(coalesce_state)           # (of the main run: the secondary RLC measurement overwrites the list)
    elapsed, accept = run.el, run.accept
    k_steps = min(args.steps, 48 if not (args.shared_workspace and len(parts) > 1) else 48 // len(parts))   # (the library's event ring holds 64 calls per workspace)
    per_part = [kernel_times(run, args.mode, k_steps, p) for p in range(len(parts))]   # [(kernel_ms overlapped, latency, shape)]
    steps_ok = run.all_steps_ok
    run.close()
    # The kernels' OWN durations: with several steps in flight the event-timed durations above include what the kernels lose
    # to each other (a pairing launch "takes" 9 ms of a 4.3 ms step).  A second pass runs ONE call at a time - same launch
    # shapes (the in-flight hint of the timed run), the host synchronises after every call - and `roofline` / `int_roofline`
    # are computed from ITS durations; the overlapped ones are reported beside them.
    alone = None
    per_part_alone = None
    coalesced = args.pipeline == "lanes" and args.mode == "per-proof" and any(f > 1 for f in co_main)
    # (coalesced calls: a one-call-at-a-time pass would launch kernels over B proofs that the timed steps never launch - their
    #  launches serve coalesce_state[p] calls each; the figures stay those of the timed steps, each call's share of its launch)
    if (inflight > 1 or args.timed_only) and not args.no_alone and not coalesced:
        saved_hint = args.hint
        args.hint = args.hint or inflight
        r1 = streams_run(args.mode, 1, 6, 2, False, sync_every_step=True)
        args.hint = saved_hint
        per_part_alone = [kernel_times(r1, args.mode, 6, p) for p in range(len(parts))]
        alone = {"ms_per_step": r1.el / 6 * 1e3}
        if any(a[2]["msm_lpt"] != b[2]["msm_lpt"] or a[2]["pair_lanes"] != b[2]["pair_lanes"] for a, b in zip(per_part_alone, per_part)):
            alone["note"] = "launch shapes differ from the timed run's"
        r1.close()

    ok_all = bool(steps_ok)   # every checked step returned exactly the expected vector (all ones, or the reject dataset's)
    if world > 1:
        flag = torch.tensor([1 if ok_all else 0], device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok_all = bool(flag.item())

    # In the default (per-proof) run the batch-accept mode is measured as well - same batch, same resident inputs, its own
    # timed region - and reported under "rlc_mode" of the one JSON line; `value` stays the per-proof figure the BASELINE
    # config ("G1 MSM + one pairing per proof") names.
    rlc_secondary = None
    if args.mode == "per-proof" and not args.no_rlc_secondary and not recursive and timed_expected is None:
        if args.pipeline == "streams":
            inflight2, probe2 = pick_inflight("rlc", [11, 7, 5, 3, 1])
            r2 = streams_run("rlc", inflight2, args.steps, args.warmup, False)
        else:
            r2, probe2 = lanes_run("rlc", args.steps, args.warmup, False), None
            inflight2 = r2.in_flight
        ok2, tm2 = r2.rlc_result(0, 0)
        rlc_secondary = {"value": round(B_total * args.steps / r2.el, 2), "unit": "proofs/s", "ms_per_step": round(r2.el / args.steps * 1e3, 4),
                         "steps_in_flight": inflight2, "inflight_probe_ms_per_step": probe2, "all_accepted": bool(r2.all_steps_ok), "batch_check_passed": ok2,
                         "bucket_msm_terms": tm2.msm_terms, "k_pip_accumulate_ms": round(tm2.bucket_accumulate_ms, 4),
                         "msm_GBps_algorithmic": round((128 * tm2.msm_terms + 144) / (tm2.bucket_accumulate_ms * 1e-3) / 1e9, 3) if tm2.bucket_accumulate_ms > 0 else None,
                         "note": "ALL-HONEST-PROVERS figure (a batch with a rejecting proof pays the fall-back: --mode rlc --reject-count 1); "
                                 "python bench.py --mode rlc prints the full line"}
        r2.close()

    # second dataset (untimed): 1 % of the proofs get the reference example's byte flip (examples/simple_mul.rs:87-95,
    # first scalar of the proof) - exactly those proofs must be rejected (rlc: through the per-proof fall-back)
    reject_check = None
    if rank == 0 and not args.timed_only:
        tot_c, tot_r, exact, fb_any = 0, 0, True, None
        for P in parts:
            rej = synth.with_rejects(P.pl, P.batch, P.vk.n_public_inputs, fraction=0.01, seed=77, kinds=["flip_first_scalar"])
            ws_r = backend.Workspace(P.dp, P.B)
            if args.mode == "rlc":
                got, fell_back = P.dp.verify_batch_rlc(rej.proofs, rej.proof_off, rej.instances, rej.committed, ws=ws_r)
                fb_any = bool(fb_any) or fell_back
            else:
                got = P.dp.verify_batch(rej.proofs, rej.proof_off, rej.instances, rej.committed, ws=ws_r)
            ws_r.close()
            tot_c += rej.expected.count(0); tot_r += int(P.B - sum(got)); exact = exact and list(got) == rej.expected
        reject_check = {"fraction": 0.01, "corrupted": tot_c, "rejected": tot_r, "exactly_the_corrupted_ones": exact, "fell_back_to_per_proof_kernels": fb_any}

    if rank == 0:
        result = report(args, parts, per_part, per_part_alone, alone, elapsed, inflight, inflight_probe, label, B, B_total, world, ranks_seen, backend_seen,
                        timed_expected, ok_all, gather_state["ok"], reject_check, t_forge, PL)
        if args.mode == "rlc":
            result["config"]["rlc_steps_routed_to_the_per_proof_kernels"] = round(max(routed_state.values()), 3) if routed_state else 0.0
        if coalesced:
            result["config"]["calls_coalesced_per_launch"] = co_main[0] if len(co_main) == 1 else list(co_main)
            for key in ("roofline", "msm_roofline"):   # (counter files describe launches of B x calls proofs: per-call traffic is not in them)
                if isinstance(result.get(key), dict) and result[key].get("traffic") is not None:
                    result[key]["traffic"] = None
                    result[key]["traffic_source"] = None
            if result.get("issue_budget"):       # (the counter passes launch the kernels over B proofs, one call at a time: not the launches of the timed steps)
                result["issue_budget"] = None
                result["issue_budget_note"] = "not computed: the timed steps run coalesced launches, the counter passes single calls"
            result["kernel_ms_is"] = "each call's SHARE of the launch that served its group of coalesced calls (include/h2v.h: COALESCING), in the timed steps"
        if len(parts) > 1 and args.pipeline == "lanes":
            result["config"]["workspaces"] = "one laned workspace for all plans (h2v_workspace_create_multi)" if args.shared_workspace else "one laned workspace per plan"
        result["config"]["tuned_launch_shapes"] = tuned_state or None    # h2v_workspace_tune per plan (0 = the launcher's rule was not beaten by 3 %)
        if rlc_secondary is not None:
            result["rlc_mode"] = rlc_secondary
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(parts, args.cpu_sample, accept)
        elif not args.no_cpu_baseline:
            result["cpu_baseline"] = None
        print(json.dumps(result))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()   # rank 0 is still checking the reject dataset / printing: leave together
        dist.destroy_process_group()
    if not ok_all:
        raise SystemExit("bench: the GPU's verdict vector differs from the synthetic batch's expected one")


def part_model(P, PL, mode, shape):
    """One part's kernels for the shapes its launcher reported: (kernel names, ALGORITHMIC bytes per launch - SURVEY.md section
    8d / DESIGN.md section 4 -, analytical lane-level multiply-adds per launch), each keyed like kernel_ms."""
    pl, B = P.pl, P.B
    T = pl.n_terms
    slots = len(pl.points) + pl.n_ci
    n_fix_terms = sum(1 for kind, _ in pl.terms if kind == PL.TERM_VK_BASE)
    msm_lpt, pair_lanes, var_lpt, rlc_shape = shape["msm_lpt"], shape["pair_lanes"], shape["var_lpt"], shape["rlc_shape"]
    # the combiner keeps its Fr register file in LDS when >= 8 proofs per block fit (h2v_capi.hip: vm_lds_slots)
    lds_slots = 64
    while lds_slots >= 8 and pl.n_regs * 32 * lds_slots + 8192 + 1024 > 160 * 1024:
        lds_slots >>= 1
    dec_name = "k_g1_decompress_queue"
    vm_name = "k_transcript_combiner_lds" if lds_slots >= 8 else "k_transcript_combiner"
    tab_point = 4 * MAD_DBL + 3 * MAD_ADD + 48 * MAD_MUL + 14 * MAD_SQR    # window tables of one point: [1..8]P, normalised, and x beta
    pairing_lane = 35 * (6 * 196 + 196) + 63 * (4 * 196 + 196) + 315 * (2 * 196 + 196) + 136 * (3 * 196 + 196)   # coop program: MUL / SQR / CSQR / LINE
    if mode == "rlc":
        n_terms_r = rlc_shape["msm_terms"]
        W = rlc_shape["windows_per_glv_half"]
        kname = {"transcript_combiner": vm_name, "g1_decompress": dec_name, "rlc_prepare": "k_rlc_prepare", "bucket_sort": "k_pip_digits",
                 "bucket_accumulate": "k_pip_accumulate", "bucket_reduce": "k_pip_reduce", "pairing": "k_pairing_rlc"}
        bytes_per_launch = {
            "bucket_accumulate": 128 * n_terms_r + 144,                   # the G1 MSM the metric names: 32 B scalar + 96 B base per term
            "g1_decompress": B * slots * (48 + 96 + 1),
            "transcript_combiner": B * (pl.proof_len + 32 * pl.n_pi + 48 * pl.n_ci + 32 * T + 4),
            "pairing": 2 * 144 + 1 + 4 + 2 * 68 * 192,
            "rlc_prepare": B * (32 * T + 5 + 32 * (T - n_fix_terms) + 4),
            "bucket_sort": 128 * n_terms_r, "bucket_reduce": W * (1 << (rlc_shape["window_bits"] - 1)) * 176,
        }
        mads = {
            "bucket_accumulate": 2 * n_terms_r * W * MAD_MADD,           # one mixed addition per (term, GLV half, window); zero digits are rare
            "g1_decompress": B * slots * (377 * MAD_SQR + 86 * MAD_MUL + 126 * MAD_DBL + 10 * MAD_ADD),
            "transcript_combiner": B * sum(128 for ins in pl.instrs if ins[0] == PL.OP_MUL),
            "pairing": 32 * pairing_lane,
            "rlc_prepare": B * T * 3 * 128, "bucket_sort": n_terms_r * MAD_MUL,
            "bucket_reduce": W * (1 << (rlc_shape["window_bits"] - 1)) * 19 * MAD_ADD,
        }
        return kname, bytes_per_launch, mads, "bucket_accumulate"
    # msm_lpt == 3: the MSM ran as TWO kernels - ladders over the n_var per-proof terms (shape var_lpt) and, beside
    # them, the fixed-base kernel over the n_fix VK-base terms (all-window tables: one mixed addition per window, no doubling)
    split = msm_lpt == 3
    shp = var_lpt if split else msm_lpt
    T_lad = T - n_fix_terms if split else T
    tpl = shp - 16 if shp in (18, 19, 20) else 1     # several terms per lane (shared doublings)
    quad = shp if shp == 8 else 0                    # a quad per GLV half (small launches of few terms)
    kname = {"g1_msm": {18: "k_g1_msm_multi", 19: "k_g1_msm_multi", 20: "k_g1_msm_multi", 1: "k_g1_msm_merged", 8: "k_g1_msm_quad"}.get(shp, "k_g1_msm"),
             "g1_msm_fixed": "k_g1_msm_fixed", "g1_decompress": dec_name, "transcript_combiner": vm_name,
             "pairing": {6: "k_pairing_six", 12: "k_pairing_coop_twelve", 16: "k_pairing_coop_narrow", 64: "k_pairing_coop_wide", 1: "k_pairing_check"}.get(pair_lanes, "k_pairing_coop")}
    # lanes per coefficient 1 / 2 / 4 (narrow / normal / wide engine): a lane multiplies 1/nq of a coefficient's terms and reduces once
    nq = {12: 1, 16: 1, 64: 4}.get(pair_lanes, 2)
    pairing_lane = sum(calls * ((terms // nq) * 196 + 196) for calls, terms in ((35, 12), (63, 8), (315, 4), (136, 6)))
    if pair_lanes == 6:
        # six lanes per proof (csrc/h2v_pairing_six.hpp): a lane owns an Fp2 coefficient; MUL / SQR / LINE are 6 / 4 / 3
        # Karatsuba terms of 3 products + 2 reductions, the cyclotomic squaring 5 products + 2 reductions + two folds of
        # 14 multiply-adds, and per Miller round one product + reduction for the line's (-lambda) xP; a wave carries 10
        # proofs on 64 lanes
        pairing_lane = 35 * (3 * 6 + 2) * 196 + 63 * (3 * 4 + 2) * 196 + 136 * (3 * 3 + 2) * 196 + 315 * ((5 + 2) * 196 + 28) + 68 * 2 * 196
    bytes_per_launch = {
        "g1_msm": B * (128 * T_lad + 144),
        "g1_msm_fixed": B * (128 * n_fix_terms + 144),
        "g1_decompress": B * slots * (48 + 96 + 1),
        "transcript_combiner": B * (pl.proof_len + 32 * pl.n_pi + 48 * pl.n_ci + 32 * T + 4),
        "pairing": B * (96 + 144 + 1 + 4) + 2 * 68 * 192,
    }
    # per ladder lane: 32 windows of 4 doublings + one mixed addition per GLV half the lane carries (tables are built
    # ahead); the launcher reports whether a term ran on two lanes (one half each) or on one (both halves)
    lpt = 1 if tpl > 1 else 2 if quad == 8 else (shp if shp in (1, 2) else 2)
    msm_halves = 2 // lpt
    msm_lane = 128 * MAD_DBL + (32 * msm_halves - 1) * MAD_MADD
    # window width of the VK bases' all-window tables (h2v_capi.hip: h2v_plan_load): 12 bits (22 additions per base) unless the
    # tables would pass 2 GB, then 8 (33)
    n_bases = len(pl.vk_bases) if hasattr(pl, "vk_bases") else n_fix_terms
    fix_adds = 22 if n_bases * 22 * 2048 * 112 <= (2 << 30) else 33
    mads = {
        "g1_msm": (B * -(-T_lad // tpl) * 128 * MAD_DBL + B * T_lad * 66 * MAD_MADD + B * (-(-T_lad // tpl) - 1) * MAD_ADD + B * 3 * MAD_MUL) if tpl > 1
                  else (B * T_lad * 8 * 33 * 18 * MAD_MUL + B * (lpt * T_lad - 1) * MAD_ADD + B * 3 * MAD_MUL) if quad  # every lane of a quad runs each level's multiplication
                  else B * T_lad * lpt * msm_lane + B * (lpt * T_lad - 1) * MAD_ADD + B * 3 * MAD_MUL,
        "g1_msm_fixed": B * n_fix_terms * fix_adds * MAD_MADD + B * n_fix_terms * MAD_ADD,
        # (twelve lanes per proof: the narrow engine's per-lane program + per Miller round one product and reduction for the
        #  line products; five proofs on the 64 lanes of a wave)
        "pairing": -(-B // 10) * 64 * pairing_lane if pair_lanes == 6 else -(-B // 5) * 64 * (pairing_lane + 68 * 2 * 196) if pair_lanes == 12
                   else B * (1 if pair_lanes == 1 else pair_lanes) * pairing_lane,
        "g1_decompress": B * slots * (377 * MAD_SQR + 86 * MAD_MUL + 126 * MAD_DBL + 10 * MAD_ADD + tab_point),
        "transcript_combiner": B * sum(128 for ins in pl.instrs if ins[0] == PL.OP_MUL),
    }
    if not split:
        bytes_per_launch.pop("g1_msm_fixed"); mads.pop("g1_msm_fixed")
    return kname, bytes_per_launch, mads, "g1_msm"


def report(args, parts, per_part, per_part_alone, alone, elapsed, inflight, inflight_probe, label, B, B_total, world, ranks_seen, backend_seen,
           timed_expected, ok_all, gathered_ok, reject_check, t_forge, PL):
    """The one JSON line: aggregates the parts of the workload (one part unless the workload is a mixed batch)."""
    ms_per_step = elapsed / args.steps * 1e3
    own = per_part_alone if per_part_alone is not None else per_part
    # (an RLC run whose steps the workspace routed to the per-proof kernels is modelled as what ran)
    mode = "rlc" if args.mode == "rlc" and own[0][2]["rlc_shape"] else "per-proof"
    models = [part_model(P, PL, mode, own[i][2]) for i, P in enumerate(parts)]
    msm_key = models[0][3]
    keys = []
    for i in range(len(parts)):
        for k in own[i][0]:
            if k not in keys and k in models[i][0]:
                keys.append(k)
    kernel_ms = {k: sum(own[i][0].get(k, 0.0) for i in range(len(parts))) for k in keys}            # own durations, summed over the parts' launches
    kernel_ms_overlapped = {k: sum(per_part[i][0].get(k, 0.0) for i in range(len(parts))) for k in keys}
    launches_of = {k: sum(own[i][2]["launches"] for i in range(len(parts)) if k in own[i][0]) for k in keys}
    bytes_per_step = {k: sum(models[i][1].get(k, 0) for i in range(len(parts)) if k in own[i][0]) for k in keys}
    mads = {k: sum(models[i][2].get(k, 0) for i in range(len(parts)) if k in own[i][0]) for k in keys}
    names_of = {k: [models[i][0][k] for i in range(len(parts)) if k in own[i][0]] for k in keys}
    kname = {k: "+".join(sorted(set(v))) for k, v in names_of.items()}
    batch_latency_ms = max(x[1] for x in per_part)
    launches = sum(x[2]["launches"] for x in own)
    msm_lpt, pair_lanes, var_lpt = own[0][2]["msm_lpt"], own[0][2]["pair_lanes"], own[0][2]["var_lpt"]
    rlc_shape, all_batch_ok = own[0][2]["rlc_shape"], own[0][2]["all_batch_ok"]
    pmc_batch = parts[0].B
    tab, hdr, pmc_src = pmc_table(args.workload, pmc_batch, mode)

    def traffic_of(k):
        vals = [pmc_traffic(tab, nm) for nm in names_of[k]]
        if not vals or any(v[0] is None for v in vals):
            return None, None
        return int(sum(v[0] for v in vals) / len(vals)), int(sum(v[1] for v in vals) / len(vals))

    def roof(k):
        # the contract's HBM roofline of one kernel: ALGORITHMIC bytes per launch / its average launch duration (its own)
        n_l = launches_of[k]
        gbps = bytes_per_step[k] / (kernel_ms[k] * 1e-3) / 1e9 if kernel_ms[k] > 0 else 0.0
        traffic, traffic_raw = traffic_of(k)
        waves = [tab[nm]["SQ_WAVES"][1] for nm in names_of[k] if tab and nm in tab and "SQ_WAVES" in tab[nm]]
        return {"kernel": kname[k], "bound": "hbm", "achieved": round(gbps, 4), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(gbps / HBM_PEAK_GBPS, 6), "traffic": traffic, "traffic_uncorrected_fetch_plus_write": traffic_raw, "traffic_source": pmc_src if traffic is not None else None,
                "avg_launch_ms": round(kernel_ms[k] / n_l, 4),
                "duration_is": "the kernel's own (one call at a time, same launch shape)" if alone else "from the timed steps (one step in flight)",
                "avg_launch_ms_in_the_timed_steps": round(kernel_ms_overlapped[k] / n_l, 4), "launches_per_step": n_l,
                # a launch of W waves occupies min(1, W / #SIMDs) of the chip: why a kernel's own duration can exceed the step it is overlapped in
                "simd_share": round(min(1.0, sum(waves) / len(waves) / 1024.0), 3) if waves else None,
                "algorithmic_bytes_per_launch": bytes_per_step[k] // n_l,
                "note": "integer-issue bound (see int_roofline / issue_budget): ~10^5 multiply-adds per 128-byte MSM term"}

    def int_roof(k):
        tops = mads[k] / (kernel_ms[k] * 1e-3) / 1e12 if kernel_ms[k] > 0 else 0.0
        return {"kernel": kname[k], "bound": "int-mad issue (measured v_mad_u64_u32 ceiling)", "achieved": round(tops, 3),
                "peak": round(IMAD_PEAK_TOPS, 2) if IMAD_PEAK_TOPS else None, "peak_source": IMAD_PEAK_SOURCE,
                "unit": "T lane-mad/s", "frac": round(tops / IMAD_PEAK_TOPS, 4) if IMAD_PEAK_TOPS else None,
                "mads_per_launch": mads[k] // launches_of[k]}

    if mode == "rlc":
        # the longest launches of an RLC batch are lone-wave chains (the ONE pairing, the bucket reduction): "dominant" is
        # the longest of the launches that fill the chip
        dominant = max(("g1_decompress", "transcript_combiner", "bucket_accumulate", "rlc_prepare", "bucket_sort"), key=kernel_ms.get)
    else:
        dominant = max(kernel_ms, key=kernel_ms.get)

    # ---- the issue budget: the falsifiable consistency check of the line (VERDICT r3, next #5a).  With the PMC summary of the
    # same workload (SQ_INSTS_VALU per launch and kernel, collected one step at a time with the timed run's launch shapes):
    #   issue_ms_lower = sum_k INSTS_k x c_min / (#SIMDs x f)          - no SIMD can issue faster than one wave64 instruction per c_min
    #   issue_ms_mad   = sum_k (mads_k / 64 x c_mad + (INSTS_k - mads_k / 64) x c_min) / (#SIMDs x f)
    # c_min = 2 cycles (a 64-lane instruction on a 32-lane SIMD), c_mad = the measured v_mad_u64_u32 interval at >= 2 waves per
    # SIMD (profiles/r*_imad_ubench.txt), f = 2.4 GHz.  The line FAILS (assertion) when the step is faster than issue_ms_lower:
    # then the counters, the launch counts or the clock are not what the line claims.
    issue = None
    c_mad, _c_add, _src = imad_costs()
    if tab and c_mad:
        c_min, n_simd, f_hz = 2.0, 1024.0, 2.4e9
        pmc_steps = int(hdr.get("pmc_steps", 0)) or None
        per_kernel, tot_insts, tot_mad_instr, covered = {}, 0.0, 0.0, True
        for k in keys:
            insts = 0.0
            for i in range(len(parts)):
                if k not in own[i][0]:
                    continue
                nm = models[i][0][k]
                if nm not in tab or "SQ_INSTS_VALU" not in tab[nm]:
                    covered = False
                    continue
                insts += tab[nm]["SQ_INSTS_VALU"][1] * own[i][2]["launches"]
            mad_instr = mads[k] / 64.0
            per_kernel[kname[k]] = {"insts_valu": int(insts), "mad_wave_instrs_analytical": int(mad_instr),
                                    "mad_share_of_issue": round(mad_instr / insts, 3) if insts else None}
            tot_insts += insts
            tot_mad_instr += min(mad_instr, insts)
        if covered and tot_insts > 0:
            lower = tot_insts * c_min / (n_simd * f_hz) * 1e3
            mad_ms = (tot_mad_instr * c_mad + (tot_insts - tot_mad_instr) * c_min) / (n_simd * f_hz) * 1e3
            issue = {"insts_valu_per_step": int(tot_insts), "mad_wave_instrs_per_step": int(tot_mad_instr), "c_min_cycles": c_min, "c_mad_cycles": c_mad,
                     "c_mad_source": _src, "issue_ms_lower": round(lower, 4), "issue_ms_mad": round(mad_ms, 4),
                     "issue_bound_frac": round(mad_ms / ms_per_step, 4), "per_kernel": per_kernel, "counters_from": pmc_src,
                     "counters_collected_on": "one step at a time, the timed run's launch shapes" + (" (%d steps)" % pmc_steps if pmc_steps else "")}
            # several ranks may share a device in the one-GPU rehearsal: recorded, not asserted, there
            assert lower <= ms_per_step * 1.02 or world > 1, ("the step is faster than the issue floor of its own instruction count", lower, ms_per_step)

    slots0 = len(parts[0].pl.points) + parts[0].pl.n_ci
    result = {
        "metric": "halo2_proofs_verified_per_sec",
        "value": round(B_total * args.steps / elapsed, 2),
        "unit": "proofs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": label, "mode": args.mode, "proofs_per_gpu": B, "proofs_per_step_all_gpus": B_total,
                   "parts": [{"circuit": P.name, "proofs_per_gpu": P.B, "proof_bytes": P.pl.proof_len, "msm_terms_per_proof": P.pl.n_terms,
                              "g1_points_per_proof": len(P.pl.points) + P.pl.n_ci, "public_inputs": P.pl.n_pi, "plan_instructions": len(P.pl.instrs)} for P in parts],
                   "proof_bytes": parts[0].pl.proof_len, "msm_terms_per_proof": parts[0].pl.n_terms, "g1_points_per_proof": slots0,
                   "public_inputs": parts[0].pl.n_pi, "plan_instructions": len(parts[0].pl.instrs),
                   "pipeline": "lanes (one laned workspace per plan, one caller stream; the library keeps the steps in flight)" if args.pipeline == "lanes"
                               else "streams (caller-driven: one workspace and one torch stream per step in flight)",
                   "steps_in_flight": inflight, "inflight_probe_ms_per_step": inflight_probe,
                   "timed_dataset": "all accepting" if timed_expected is None else "%d of %d proofs corrupted" % (timed_expected.count(0), B),
                   "ranks_seen": ranks_seen, "dist_backend": backend_seen,
                   "parallelism": ("independent proofs sharded per GPU (%s); accept bytes of all steps gathered once over %s" % (
                       args.scaling, "RCCL" if args.dist_backend == "nccl" else "gloo (rehearsal)")) if world > 1 else "1 GPU"},
        "roofline": roof(dominant),
        "msm_roofline": roof(msm_key),
        "msm_fixed_roofline": roof("g1_msm_fixed") if "g1_msm_fixed" in kernel_ms else None,
        "int_roofline": int_roof(dominant),
        "msm_int_roofline": int_roof(msm_key),
        "issue_budget": issue,
        "kernel_ms": {kname[k]: round(v, 4) for k, v in kernel_ms.items()},
        "kernel_ms_is": "each kernel's own duration, one call at a time (summed over the parts of a mixed batch)" if alone else "durations in the timed steps",
        "kernel_ms_in_the_timed_steps": {kname[k]: round(v, 4) for k, v in kernel_ms_overlapped.items()},
        "ms_per_step_one_step_at_a_time": round(alone["ms_per_step"], 4) if alone else None,
        "own_duration_pass_note": alone.get("note") if alone else None,
        "step_int_roofline": {"what": "analytical lane-level multiply-adds of ALL kernels of a step / ms_per_step, against the measured v_mad_u64_u32 ceiling",
                              "achieved": round(sum(mads.values()) / (elapsed / args.steps) / 1e12, 3), "peak": round(IMAD_PEAK_TOPS, 2) if IMAD_PEAK_TOPS else None,
                              "unit": "T lane-mad/s", "frac": round(sum(mads.values()) / (elapsed / args.steps) / 1e12 / IMAD_PEAK_TOPS, 4) if IMAD_PEAK_TOPS else None},
        "batch_latency_ms": round(batch_latency_ms, 4),
        "pipelines_per_step": launches, "msm_lanes_per_term": msm_lpt, "msm_ladder_shape_of_a_split": var_lpt if msm_lpt == 3 else None, "pairing_lanes_per_proof": pair_lanes if mode == "per-proof" else None,
        "shapes_per_part": [{"circuit": P.name, "msm_lanes_per_term": own[i][2]["msm_lpt"], "msm_ladder_shape_of_a_split": own[i][2]["var_lpt"],
                             "pairing_lanes_per_proof": own[i][2]["pair_lanes"]} for i, P in enumerate(parts)] if len(parts) > 1 else None,
        "all_accepted": ok_all if timed_expected is None else None,
        "verdicts_as_expected_every_checked_step": ok_all,
        "gathered_accept_vectors_all_ones": gathered_ok,
        "reject_dataset": reject_check,
        "forge_seconds": round(t_forge, 2),
    }
    if mode == "rlc":
        result["rlc"] = dict(rlc_shape, batch_check_passed_every_step=all_batch_ok,
                             soundness="accept[] equals the per-proof mode's except with probability <= 2^-128 over the seed")
    return result


def host_cores():
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # a container may see every host CPU but be allowed only a share of them (cgroup v2 cpu.max / v1 cfs quota):
    # threads beyond the share only time-slice, so the thread count follows the share
    try:
        quota = None
        if os.path.exists("/sys/fs/cgroup/cpu.max"):
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            quota = None if q == "max" else float(q) / float(per)
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = None if q <= 0 else q / per
        if quota:
            cores = max(1, min(cores, int(quota + 0.5)))
    except Exception:
        pass
    return cores


def cpu_baseline(parts, sample, gpu_accept):
    """The CPU oracle (oracle/c: C restatement of the reference verifier, 64-bit Montgomery limbs) timed on this
    box's host cores on a bounded sample of the same batch (a mixed batch: the same share of every part), and compared
    with the GPU result on that sample."""
    import json as _json
    from oracle import binding as orc  # cpu_baseline leg: the checker timed as the reported baseline

    cores = host_cores()
    n_tot, t_multi, t_single_sum, n_single, match, what = 0, 0.0, 0.0, 0, True, []
    for P in parts:
        vk, batch = P.vk, P.timed_batch
        ov = orc.OracleVK(orc.vk_desc(_json.loads(vk.to_json()), vk.omega, vk.omega_inv, vk.barycentric_weight))
        n = min(max(sample // len(parts), 8 * cores), batch.n)  # at least 8 proofs per thread
        n_pi = vk.n_public_inputs
        proofs = batch.proofs[:batch.proof_off[n]]
        inst = batch.instances[:32 * n_pi * n]
        ci = batch.committed[:48 * n] if batch.committed else None
        n1 = min(32, n)
        t0 = time.perf_counter()
        ov.verify_batch(proofs[:batch.proof_off[n1]], batch.proof_off[:n1 + 1], inst[:32 * n_pi * n1], ci[:48 * n1] if ci else None, threads=1)
        t_single_sum += time.perf_counter() - t0
        n_single += n1
        t0 = time.perf_counter()
        acc = ov.verify_batch(proofs, batch.proof_off[:n + 1], inst, ci, threads=cores)
        t_multi += time.perf_counter() - t0
        n_tot += n
        match = match and list(acc) == [int(x) for x in gpu_accept[P.off:P.off + n]]
        what.append("first %d proofs of the %s batch" % (n, P.name))
    return {"value": round(n_tot / t_multi, 2), "unit": "proofs/s", "cores": cores, "kind": "port",
            "sample": "%s, %d threads (pthreads over independent proofs)" % (" + ".join(what), cores),
            "single_core_proofs_per_s": round(n_single / t_single_sum, 2),
            "matches_gpu_on_sample": bool(match)}


if __name__ == "__main__":
    main()
