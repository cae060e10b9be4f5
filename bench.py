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
contiguous index ranges (shard.shard_range).  Either way there is no data-path collective; the per-step accept bytes are
gathered on rank 0 over RCCL as the north star asks.  --inflight P keeps P steps in flight on P workspaces / streams
(every step still runs completely inside the timed region).  Prints ONE JSON line on rank 0.
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


def imad_peak():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_imad_ubench.txt")))
    if not files:
        return None, None
    cyc, sect = None, None
    for line in open(files[-1]):
        m = re.match(r"--- (\d+) wave", line)
        if m:
            sect = int(m.group(1))
        m = re.match(r"mad_u64_u32 .*~([0-9.]+) cycles", line)
        if m and sect == 4:
            cyc = float(m.group(1))
    if not cyc:
        return None, None
    return 1024 * 64 * 2.4e9 / cyc / 1e12, os.path.basename(files[-1])


IMAD_PEAK_TOPS, IMAD_PEAK_SOURCE = imad_peak()
# v_mad_u64_u32 per field operation (csrc/h2v_field.hpp): product-scanning multiply 392, square 301
MAD_MUL, MAD_SQR = 392, 301
MAD_DBL = 2 * MAD_MUL + 5 * MAD_SQR          # dbl-2009-l
MAD_MADD = 8 * MAD_MUL + 3 * MAD_SQR         # mixed addition
MAD_ADD = 12 * MAD_MUL + 4 * MAD_SQR         # full Jacobian addition

WORKLOADS = {
    # name: (vk builder, proofs per GPU, BASELINE config label)
    "simple_mul": ("simple_mul", 4096, "simple_mul x4096 per GPU (BASELINE configs[1])"),
    "lookup_mixed": ("lookup_table", 2048, "lookup_table x2048 (half of BASELINE configs[2])"),
    "atms_with_lookups": ("atms_with_lookups", 2048, "atms_with_lookups x2048 (half of BASELINE configs[2])"),
    "sha256": ("sha256", 1024, "sha256-shaped x1024 (BASELINE configs[3])"),
    "secp256k1": ("secp256k1", 512, "secp256k1-shaped x512 (BASELINE configs[4])"),
    "ivc": ("ivc", 1024, "IVC-shaped recursive circuit x1024 per GPU (accumulator fold, DESIGN.md section 10)"),
}


def pmc_traffic(kernel, workload, batch, mode):
    """HBM bytes per launch (FETCH_SIZE + WRITE_SIZE, KiB on gfx950) of `kernel` from the newest committed PMC summary
    that was collected on THIS workload / batch / mode (header line `# workload=... batch=... mode=...`, written by
    tools/scripts/profile_round.sh over separate rocprofv3 --pmc passes of this command); (None, None) otherwise."""
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
        vals = {}
        for line in lines[1:]:
            f = line.split()
            if len(f) >= 4 and f[0] == kernel and f[1] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[f[1]] = float(f[3])
        if len(vals) == 2:
            return int((vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), os.path.basename(path)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="simple_mul", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="proofs per GPU (weak) / in total (strong); default: the workload's BASELINE size")
    ap.add_argument("--mode", default="per-proof", choices=["per-proof", "rlc"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--inflight", type=int, default=0, help="steps in flight (workspaces / streams); default: 2 (per-proof, batches above 1024), 4 (per-proof, smaller), 7 (rlc)")
    ap.add_argument("--msm-tpl", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="per-proof MSM: terms per lane (sets H2V_MSM_TPL; 2 / 4 share the doublings of a lane's terms)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product path); gloo = rehearsal of the N > 1 code path on a box with one GPU "
                         "(every rank on the device H2V_BENCH_DEVICE names, accept bytes gathered through host memory)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rlc-secondary", action="store_true", help="per-proof runs: skip the extra measurement of the RLC mode")
    ap.add_argument("--cpu-sample", type=int, default=1024)
    ap.add_argument("--hint", type=int, default=0, help="h2v_workspace_hint_in_flight value (default: the steps in flight); the counter passes of "
                                                     "tools/scripts/profile_round.sh run one step at a time with the shapes of the timed run")
    ap.add_argument("--timed-only", action="store_true",
                    help="launch nothing but warm-up + the timed steps (no in-flight probe, no one-step pass, no RLC secondary, no reject "
                         "dataset, no CPU baseline): the form profiled under rocprofv3, whose per-kernel averages then cover the same "
                         "launches as the line's kernel_ms")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # steps in flight: measured on MI355X - per-proof simple_mul x4096: 1 -> 6.11 ms per step, 2 -> 5.29,
    # 3 -> 5.25; the 512 / 1024-proof batches of the large circuits, whose kernels are lone-wave chains: 1 -> 4.98 / 5.51 ms,
    # 4 -> 2.74 / 3.31; RLC: 1 -> 5.0 ms, 4 -> 2.04, 5 -> 1.84 (stable over boxes and runs), 6 -> 1.80 OR 3.4-3.9 (bimodal:
    # with twelve streams the runtime sometimes maps two busy ones onto one hardware queue), 7 -> 2.0 (the tail of a batch -
    # bucket reduction, doublings, ONE pairing - is a few waves)
    # Not every box overlaps streams equally well (one measured 2.2 ms -> 6.5 ms per RLC step with seven in flight where the
    # others gain 3x), so without --inflight the count is PROBED: a few untimed steps with each candidate, the best one is used
    # for the timed region and reported (config.steps_in_flight, inflight_probe_ms_per_step).
    small = (args.batch or WORKLOADS[args.workload][1]) <= 1024
    inflight_candidates = [args.inflight] if args.inflight else ([11, 7, 5, 3, 1] if args.mode == "rlc" else [4, 2, 1] if small else [5, 3, 2, 1])
    inflight = inflight_candidates[0]
    if args.msm_tpl:
        os.environ["H2V_MSM_TPL"] = str(args.msm_tpl)
    # several steps in flight use 3 streams each: more hardware queues than the runtime's default of 4, or they serialise
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    if os.environ.get("H2V_BENCH_DEVICE") is not None:   # rehearsal on a one-GPU box: all ranks share that device
        local_rank = int(os.environ["H2V_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)          # one process per GPU; bind before the communicator is created
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)  # RCCL on ROCm
        else:
            dist.init_process_group(backend="gloo")
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    from plutus_halo2_verifier_gen_amd import backend, plan as PL, shard, synth, vk as V

    vk_name, default_batch, label = WORKLOADS[args.workload]
    B_arg = args.batch or default_batch
    if args.scaling == "strong":
        lo, hi = shard.shard_range(B_arg, rank, world)   # this rank's contiguous range of the ONE batch
        B = hi - lo
        B_total = B_arg
    else:
        B = B_arg
        B_total = B_arg * world
    if B == 0:
        raise SystemExit("strong scaling: more ranks than proofs")
    vk, td = V.BUILDERS[vk_name]()
    pl = PL.compile_plan(vk)
    ncpu = os.cpu_count() or 1
    workers = max(1, min(16, ncpu // max(1, world)))
    t0 = time.time()
    batch = synth.forge_batch(vk, td, B, seed=1000 + rank, workers=workers, plan=pl)
    t_forge = time.time() - t0

    dp = backend.DevicePlan(pl.to_bytes(), device=local_rank)

    def to_dev(b):
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) if b else torch.zeros(1, dtype=torch.uint8, device=dev)

    d_proofs = to_dev(batch.proofs)
    d_off = torch.tensor(batch.proof_off, dtype=torch.int64).to(dev)
    d_inst = to_dev(batch.instances)
    d_ci = to_dev(batch.committed) if batch.committed else None
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")   # where the collectives' tensors live
    Bmax = B
    if world > 1:
        t = torch.tensor([B], device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        Bmax = int(t.item())
    rlc_seed = bytes((7 * k + 1) & 0xff for k in range(32))   # fixed for the timed steps (reproducible); a service draws it per batch

    stream_pool = []

    def timed_run(mode, inflight, steps, warmup, gather):
        """warmup + `steps` timed passes in `mode` with `inflight` steps in flight; returns (elapsed s, workspaces, accept)"""
        wss = [backend.Workspace(dp, B) for _ in range(inflight)]
        for w_ in wss:
            w_.hint_in_flight(args.hint or inflight)     # (from 4 up the library prefers launch shapes that issue fewer instructions)
        # (the same few torch streams in every measurement of this process: every stream that was ever created keeps a
        #  hardware queue busy in the runtime's round-robin, and later measurements would collide with the earlier ones')
        while len(stream_pool) < inflight:
            stream_pool.append(torch.cuda.Stream(device=dev))
        streams = stream_pool[:inflight] if inflight > 1 else [None]
        d_accepts = [torch.zeros(B, dtype=torch.uint8, device=dev) for _ in range(inflight)]
        d_statuses = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(inflight)]
        use_gather = gather and world > 1
        g_send = [torch.zeros(Bmax, dtype=torch.uint8, device=cdev) for _ in range(inflight)] if use_gather else None
        gathered = [[torch.zeros(Bmax, dtype=torch.uint8, device=cdev) for _ in range(world)] for _ in range(inflight)] if (use_gather and rank == 0) else None
        torch.cuda.synchronize()

        def step(k):
            slot = k % inflight
            st = streams[slot]
            ctx = torch.cuda.stream(st) if st is not None else None
            if ctx is not None:
                ctx.__enter__()
            try:
                stream = torch.cuda.current_stream().cuda_stream
                ptrs = (B, d_proofs.data_ptr(), d_off.data_ptr(), d_inst.data_ptr(), d_ci.data_ptr() if d_ci is not None else None,
                        d_accepts[slot].data_ptr(), d_statuses[slot].data_ptr())
                if mode == "rlc":
                    dp.verify_batch_rlc_device(*ptrs, ws=wss[slot], stream=stream, seed=rlc_seed, one_stream=inflight >= 3)
                else:
                    dp.verify_batch_device(*ptrs, ws=wss[slot], stream=stream)
                if use_gather:
                    # final accept/reject gather over RCCL (xGMI): the rank's accept bytes (padded to the widest shard)
                    g_send[slot][:B].copy_(d_accepts[slot])
                    dist.gather(g_send[slot], gathered[slot] if gathered else None, dst=0)
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
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        if gathered is not None:   # rank 0: what arrived over the collective is every rank's (all-accepting) vector
            sizes = [shard.shard_range(B_arg, r, world)[1] - shard.shard_range(B_arg, r, world)[0] if args.scaling == "strong" else B_arg
                     for r in range(world)]
            last = gathered[(steps - 1) % inflight]
            gather_state["ok"] = all(bool(last[r][:sizes[r]].all().item()) for r in range(world))
        return el, wss, d_accepts[(steps - 1) % inflight].cpu().numpy()

    def pick_inflight(mode, candidates):
        """a few untimed steps with each candidate number of steps in flight; returns (best, {candidate: ms per step})"""
        if len(candidates) == 1:
            return candidates[0], None
        seen = {}
        for cand in candidates:
            k = args.steps            # as many steps as the timed region: filling and draining the pipeline count the same
            el_, w_, _a = timed_run(mode, cand, k, min(args.warmup, 3), False)
            for x in w_:
                x.close()
            seen[cand] = round(el_ / k * 1e3, 4)
        best = min(seen, key=seen.get)
        if world > 1:   # every rank the same count (the slowest rank's view decides nothing: take rank 0's)
            t = torch.tensor([best], device=cdev)
            dist.broadcast(t, src=0)
            best = int(t.item())
        return best, seen

    gather_state = {"ok": None}
    if args.timed_only:
        args.no_cpu_baseline = args.no_rlc_secondary = True
        inflight_candidates = inflight_candidates[:1]
    # with several steps in flight the event-timed kernel durations include what the kernels lose to each other; a short
    # pass with ONE step in flight gives the kernels' own durations beside them.  It runs FIRST: after runs with many
    # streams the runtime may map the combiner's and the decompression's stream onto one hardware queue, and the event-timed
    # duration of the second then includes its wait for the first (combiner 1.5 instead of 0.66 ms).
    kernel_ms_alone = None
    if args.mode == "per-proof" and not args.timed_only and len(inflight_candidates) > 1:
        _el1, wss1, _acc1 = timed_run("per-proof", 1, 5, 3, False)
        a1 = {"transcript_combiner": 0.0, "g1_decompress": 0.0, "g1_msm": 0.0, "pairing": 0.0}
        for j in range(5):
            tm = wss1[0].timings(j)
            a1["transcript_combiner"] += tm.transcript_combiner_ms / 5; a1["g1_decompress"] += tm.g1_decompress_ms / 5
            a1["g1_msm"] += tm.g1_msm_ms / 5; a1["pairing"] += tm.pairing_ms / 5
        kernel_ms_alone = dict(a1, ms_per_step=_el1 / 5 * 1e3, msm_lpt=tm.msm_lanes_per_term or 2, pair_lanes=tm.pairing_lanes_per_proof or 32)
        wss1[0].close()
        del wss1
    inflight, inflight_probe = pick_inflight(args.mode, inflight_candidates)
    elapsed, wss, accept = timed_run(args.mode, inflight, args.steps, args.warmup, True)

    # per-kernel device time over the timed steps (HIP events recorded on the kernels' own streams, event rings of the
    # workspaces): averages over the last min(steps, 64) steps
    k_steps = min(args.steps, 64)
    rlc_shape = None
    if args.mode == "rlc":
        names = ["transcript_combiner", "g1_decompress", "rlc_prepare", "bucket_sort", "bucket_accumulate", "bucket_reduce", "pairing"]
        acc = {k: 0.0 for k in names}
        span = 0.0
        all_batch_ok = True
        for j in range(k_steps):
            slot = (args.steps - 1 - j) % inflight
            ok, tm = wss[slot].rlc_result(calls_back=j // inflight)
            all_batch_ok = all_batch_ok and ok
            for nm, v in zip(names, [tm.transcript_combiner_ms, tm.g1_decompress_ms, tm.prepare_ms, tm.bucket_sort_ms,
                                     tm.bucket_accumulate_ms, tm.bucket_reduce_ms, tm.pairing_ms]):
                acc[nm] += v
            span += tm.total_ms
            rlc_shape = {"msm_terms": tm.msm_terms, "window_bits": tm.window_bits, "windows_per_glv_half": tm.windows,
                         "max_entries_per_lane": tm.max_chain}
        kernel_ms = {k: v / k_steps for k, v in acc.items()}
        batch_latency_ms = span / k_steps
        launches, msm_lpt, pair_lanes = 1, 0, 64
    else:
        acc = {"transcript_combiner": 0.0, "g1_decompress": 0.0, "g1_msm": 0.0, "pairing": 0.0}
        launches, msm_lpt, span, pair_lanes = 1, 2, 0.0, 32
        for j in range(k_steps):
            slot = (args.steps - 1 - j) % inflight
            tm = wss[slot].timings(j // inflight)
            launches = max(1, tm.launches)
            msm_lpt = tm.msm_lanes_per_term or 2
            pair_lanes = tm.pairing_lanes_per_proof or 32
            acc["transcript_combiner"] += tm.transcript_combiner_ms
            acc["g1_decompress"] += tm.g1_decompress_ms
            acc["g1_msm"] += tm.g1_msm_ms
            acc["pairing"] += tm.pairing_ms
            span += tm.total_ms
        kernel_ms = {k: v / k_steps for k, v in acc.items()}
        batch_latency_ms = span / k_steps
        all_batch_ok = None

    if inflight <= 1:
        kernel_ms_alone = None
    if kernel_ms_alone is not None:
        for w_ in wss[1:]:
            w_.close()            # (their event rings have been read; their streams give their hardware queues back)
    n_accept = int(accept.sum())
    ok_all = n_accept == B  # the synthetic batch is 100 % accepting
    if world > 1:
        flag = torch.tensor([1 if ok_all else 0], device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok_all = bool(flag.item())

    # In the default (per-proof) run the batch-accept mode is measured as well - same batch, same resident inputs, its own
    # timed region - and reported under "rlc_mode" of the one JSON line; `value` stays the per-proof figure the BASELINE
    # config ("G1 MSM + one pairing per proof") names.
    rlc_secondary = None
    if args.mode == "per-proof" and not args.no_rlc_secondary and vk.recursion_vks is None:
        for w_ in wss[1:]:
            w_.close()
        inflight2, probe2 = pick_inflight("rlc", [11, 7, 5, 3, 1])
        el2, wss2, acc2 = timed_run("rlc", inflight2, args.steps, args.warmup, False)
        ok2, tm2 = wss2[0].rlc_result()
        rlc_secondary = {"value": round(B_total * args.steps / el2, 2), "unit": "proofs/s", "ms_per_step": round(el2 / args.steps * 1e3, 4),
                         "steps_in_flight": inflight2, "inflight_probe_ms_per_step": probe2, "all_accepted": bool(int(acc2.sum()) == B), "batch_check_passed": ok2,
                         "bucket_msm_terms": tm2.msm_terms, "k_pip_accumulate_ms": round(tm2.bucket_accumulate_ms, 4),
                         "msm_GBps_algorithmic": round((128 * tm2.msm_terms + 144) / (tm2.bucket_accumulate_ms * 1e-3) / 1e9, 3) if tm2.bucket_accumulate_ms > 0 else None,
                         "note": "python bench.py --mode rlc prints the full line (roofline of the bucket kernel, kernel times)"}
        for w_ in wss2:
            w_.close()
        del wss2

    # second dataset (untimed): 1 % of the proofs get the reference example's byte flip (examples/simple_mul.rs:87-95,
    # first scalar of the proof) - exactly those proofs must be rejected (rlc: through the per-proof fall-back)
    reject_check = None
    if rank == 0 and not args.timed_only:
        rej = synth.with_rejects(pl, batch, vk.n_public_inputs, fraction=0.01, seed=77, kinds=["flip_first_scalar"])
        if args.mode == "rlc":
            got, fell_back = dp.verify_batch_rlc(rej.proofs, rej.proof_off, rej.instances, rej.committed, ws=wss[0])
        else:
            got, fell_back = dp.verify_batch(rej.proofs, rej.proof_off, rej.instances, rej.committed, ws=wss[0]), None
        reject_check = {"fraction": 0.01, "corrupted": rej.expected.count(0), "rejected": int(B - sum(got)),
                        "exactly_the_corrupted_ones": list(got) == rej.expected, "fell_back_to_per_proof_kernels": fell_back}

    if rank == 0:
        T = pl.n_terms
        slots = len(pl.points) + pl.n_ci
        n_fix_terms = sum(1 for kind, _ in pl.terms if kind == PL.TERM_VK_BASE)
        # the combiner keeps its Fr register file in LDS when >= 8 proofs per block fit (h2v_capi.hip: vm_lds_slots)
        lds_slots = 64
        while lds_slots >= 8 and pl.n_regs * 32 * lds_slots + 8192 + 1024 > 160 * 1024:
            lds_slots >>= 1
        dec_name = "k_g1_decompress" if (os.environ.get("H2V_DEC_QUEUE") == "0" or os.environ.get("H2V_SPLIT_DEC") == "0") else "k_g1_decompress_queue"
        vm_name = "k_transcript_combiner_lds" if lds_slots >= 8 else "k_transcript_combiner"
        tab_point = 4 * MAD_DBL + 3 * MAD_ADD + 48 * MAD_MUL + 14 * MAD_SQR    # window tables of one point: [1..8]P, normalised, and x beta
        pairing_lane = 35 * (6 * 196 + 196) + 63 * (4 * 196 + 196) + 315 * (2 * 196 + 196) + 136 * (3 * 196 + 196)   # coop program: MUL / SQR / CSQR / LINE
        # ALGORITHMIC bytes per launch (SURVEY.md section 8d / DESIGN.md section 4) and analytical lane-level multiply-adds
        if args.mode == "rlc":
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
            msm_key = "bucket_accumulate"
        else:
            tpl = msm_lpt - 16 if msm_lpt in (18, 19, 20) else 1     # several terms per lane (H2V_MSM_TPL): shared doublings
            quad = msm_lpt if msm_lpt == 8 else 0                # a quad per GLV half (small launches of few terms)

            def shape_names(lpt_code, lanes):   # the kernels behind the launcher's reported shapes
                return {"g1_msm": {18: "k_g1_msm_multi2", 19: "k_g1_msm_multi3", 20: "k_g1_msm_multi4", 3: "k_g1_msm_fixed", 1: "k_g1_msm_merged", 8: "k_g1_msm_quad"}.get(lpt_code, "k_g1_msm"),
                        "g1_decompress": dec_name, "transcript_combiner": vm_name,
                        "pairing": {16: "k_pairing_coop_narrow", 64: "k_pairing_coop_wide", 1: "k_pairing_check"}.get(lanes, "k_pairing_coop")}
            kname = shape_names(msm_lpt, pair_lanes)
            # lanes per coefficient 1 / 2 / 4 (narrow / normal / wide engine): a lane multiplies 1/nq of a coefficient's terms and reduces once
            nq = {16: 1, 64: 4}.get(pair_lanes, 2)
            pairing_lane = sum(calls * ((terms // nq) * 196 + 196) for calls, terms in ((35, 12), (63, 8), (315, 4), (136, 6)))
            bytes_per_launch = {
                "g1_msm": B * (128 * T + 144),
                "g1_decompress": B * slots * (48 + 96 + 1),
                "transcript_combiner": B * (pl.proof_len + 32 * pl.n_pi + 48 * pl.n_ci + 32 * T + 4),
                "pairing": B * (96 + 144 + 1 + 4) + 2 * 68 * 192,
            }
            # per MSM lane: 32 windows of 4 doublings + one mixed addition per GLV half the lane carries (tables are built
            # ahead); the launcher reports whether a term ran on two lanes (one half each) or on one (both halves)
            msm_fixed = msm_lpt == 3   # fixed-base mode: VK-base terms cost 65 mixed additions and no doubling
            lpt = 1 if (msm_fixed or tpl > 1) else 2 if quad == 8 else msm_lpt
            msm_halves = 2 // lpt
            msm_lane = 128 * MAD_DBL + (32 * msm_halves - 1) * MAD_MADD
            mads = {
                "g1_msm": (B * (T - n_fix_terms) * msm_lane + B * n_fix_terms * 65 * MAD_MADD + B * (T - 1) * MAD_ADD + B * 3 * MAD_MUL) if msm_fixed
                          else (B * -(-T // tpl) * 128 * MAD_DBL + B * T * 66 * MAD_MADD + B * (-(-T // tpl) - 1) * MAD_ADD + B * 3 * MAD_MUL) if tpl > 1
                          else (B * T * 8 * 33 * 18 * MAD_MUL + B * (lpt * T - 1) * MAD_ADD + B * 3 * MAD_MUL) if quad  # every lane of a quad runs each level's multiplication
                          else B * T * lpt * msm_lane + B * (lpt * T - 1) * MAD_ADD + B * 3 * MAD_MUL,
                "pairing": B * (1 if pair_lanes == 1 else pair_lanes) * pairing_lane,
                "g1_decompress": B * slots * (377 * MAD_SQR + 86 * MAD_MUL + 126 * MAD_DBL + 10 * MAD_ADD + tab_point),
                "transcript_combiner": B * sum(128 for ins in pl.instrs if ins[0] == PL.OP_MUL),
            }
            msm_key = "g1_msm"

        def roof(k):
            # the contract's HBM roofline of one kernel: ALGORITHMIC bytes per launch / its average launch duration
            gbps = bytes_per_launch[k] / (kernel_ms[k] * 1e-3) / 1e9 if kernel_ms[k] > 0 else 0.0
            traffic, src = pmc_traffic(kname[k], args.workload, B, args.mode)
            return {"kernel": kname[k], "bound": "hbm", "achieved": round(gbps, 4), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(gbps / HBM_PEAK_GBPS, 6), "traffic": traffic, "traffic_source": src,
                    "avg_launch_ms": round(kernel_ms[k] / launches, 4), "launches_per_step": launches,
                    "algorithmic_bytes_per_launch": bytes_per_launch[k] // launches,
                    "note": "integer-issue bound (see int_roofline): ~10^5 multiply-adds per 128-byte MSM term"}

        def int_roof(k):
            tops = mads[k] / (kernel_ms[k] * 1e-3) / 1e12 if kernel_ms[k] > 0 else 0.0
            return {"kernel": kname[k], "bound": "int-mad issue (measured v_mad_u64_u32 ceiling)", "achieved": round(tops, 3),
                    "peak": round(IMAD_PEAK_TOPS, 2) if IMAD_PEAK_TOPS else None, "peak_source": IMAD_PEAK_SOURCE,
                    "unit": "T lane-mad/s", "frac": round(tops / IMAD_PEAK_TOPS, 4) if IMAD_PEAK_TOPS else None,
                    "mads_per_launch": mads[k] // launches}

        dominant = max(kernel_ms, key=kernel_ms.get)
        result = {
            "metric": "halo2_proofs_verified_per_sec",
            "value": round(B_total * args.steps / elapsed, 2),
            "unit": "proofs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": label, "mode": args.mode, "proofs_per_gpu": B, "proofs_per_step_all_gpus": B_total, "proof_bytes": pl.proof_len,
                       "msm_terms_per_proof": T, "g1_points_per_proof": slots, "public_inputs": pl.n_pi, "plan_instructions": len(pl.instrs),
                       "steps_in_flight": inflight, "inflight_probe_ms_per_step": inflight_probe,
                       "parallelism": ("independent proofs sharded per GPU (%s); accept gather over RCCL" % args.scaling) if world > 1 else "1 GPU"},
            "roofline": roof(dominant),
            "msm_roofline": roof(msm_key),
            "int_roofline": int_roof(dominant),
            "msm_int_roofline": int_roof(msm_key),
            "kernel_ms": {kname[k]: round(v, 4) for k, v in kernel_ms.items()},
            "kernel_ms_one_step_in_flight": ({shape_names(kernel_ms_alone["msm_lpt"], kernel_ms_alone["pair_lanes"])[k]: round(v, 4)
                                              for k, v in kernel_ms_alone.items() if k in kname} if kernel_ms_alone and args.mode == "per-proof" else None),
            "ms_per_step_one_step_in_flight": round(kernel_ms_alone["ms_per_step"], 4) if kernel_ms_alone else None,
            "step_int_roofline": {"what": "analytical lane-level multiply-adds of ALL kernels of a step / ms_per_step, against the measured v_mad_u64_u32 ceiling",
                                  "achieved": round(sum(mads.values()) / (elapsed / args.steps) / 1e12, 3), "peak": round(IMAD_PEAK_TOPS, 2) if IMAD_PEAK_TOPS else None,
                                  "unit": "T lane-mad/s", "frac": round(sum(mads.values()) / (elapsed / args.steps) / 1e12 / IMAD_PEAK_TOPS, 4) if IMAD_PEAK_TOPS else None},
            "batch_latency_ms": round(batch_latency_ms, 4),
            "pipelines_per_step": launches, "msm_lanes_per_term": msm_lpt, "pairing_lanes_per_proof": pair_lanes if args.mode == "per-proof" else None,
            "all_accepted": ok_all,
            "gathered_accept_vectors_all_ones": gather_state["ok"],
            "reject_dataset": reject_check,
            "forge_seconds": round(t_forge, 2),
        }
        if rlc_secondary is not None:
            result["rlc_mode"] = rlc_secondary
        if args.mode == "rlc":
            result["rlc"] = dict(rlc_shape, batch_check_passed_every_step=all_batch_ok,
                                 soundness="accept[] equals the per-proof mode's except with probability <= 2^-128 over the seed")
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(vk, batch, pl, args.cpu_sample, accept)
        elif not args.no_cpu_baseline:
            result["cpu_baseline"] = None
        print(json.dumps(result))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()   # rank 0 is still checking the reject dataset / printing: leave together
        dist.destroy_process_group()
    if not ok_all:
        raise SystemExit("bench: GPU rejected proofs of an all-accepting synthetic batch")


def cpu_baseline(vk, batch, pl, sample, gpu_accept):
    """The CPU oracle (oracle/c: C restatement of the reference verifier, 64-bit Montgomery limbs) timed on this
    box's host cores on a bounded sample of the same batch, and compared with the GPU result on that sample."""
    import json as _json
    from oracle import binding as orc  # cpu_baseline leg: the checker timed as the reported baseline

    ov = orc.OracleVK(orc.vk_desc(_json.loads(vk.to_json()), vk.omega, vk.omega_inv, vk.barycentric_weight))
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
    n = min(max(sample, 8 * cores), batch.n)  # at least 8 proofs per thread
    n_pi = vk.n_public_inputs
    proofs = batch.proofs[:batch.proof_off[n]]
    inst = batch.instances[:32 * n_pi * n]
    ci = batch.committed[:48 * n] if batch.committed else None
    n1 = min(32, n)
    t0 = time.perf_counter()
    ov.verify_batch(proofs[:batch.proof_off[n1]], batch.proof_off[:n1 + 1], inst[:32 * n_pi * n1], ci[:48 * n1] if ci else None, threads=1)
    t_single = (time.perf_counter() - t0) / n1
    t0 = time.perf_counter()
    acc = ov.verify_batch(proofs, batch.proof_off[:n + 1], inst, ci, threads=cores)
    t_multi = time.perf_counter() - t0
    return {"value": round(n / t_multi, 2), "unit": "proofs/s", "cores": cores, "kind": "port",
            "sample": "first %d proofs of the same batch, %d threads (pthreads over independent proofs)" % (n, cores),
            "single_core_proofs_per_s": round(1.0 / t_single, 2),
            "matches_gpu_on_sample": bool(list(acc) == [int(x) for x in gpu_accept[:n]])}


if __name__ == "__main__":
    main()
