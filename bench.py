#!/usr/bin/env python3
"""Benchmark of the Halo2/KZG verification hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch of synthetic proofs per GPU: transcript replay + Fr combiner,
G1 decompression, per-proof G1 MSM and the fused pairing check, with every input already resident in HBM.
Workload: BASELINE.json configs[1] - simple_mul, 4096 proofs per GPU (independent proofs shard across GPUs with no
data-path collective: weak scaling; the per-step accept vectors are gathered on rank 0 over RCCL as the north star
asks).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_imad_ubench.txt")), key=os.path.getmtime)
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
    "sha256": ("sha256", 1024, "sha256-shaped x1024 per GPU (BASELINE configs[3])"),
    "secp256k1": ("secp256k1", 512, "secp256k1-shaped x512 per GPU (BASELINE configs[4])"),
    "ivc": ("ivc", 1024, "IVC-shaped recursive circuit x1024 per GPU (accumulator fold, DESIGN.md section 10)"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="simple_mul", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="proofs per GPU (default: the workload's BASELINE size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1024)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    torch.cuda.set_device(local_rank)          # one process per GPU; bind before the communicator is created
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)  # RCCL on ROCm
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    from plutus_halo2_verifier_gen_amd import backend, plan as PL, synth, vk as V

    vk_name, default_batch, label = WORKLOADS[args.workload]
    B = args.batch or default_batch
    vk, td = V.BUILDERS[vk_name]()
    pl = PL.compile_plan(vk)
    ncpu = os.cpu_count() or 1
    workers = max(1, min(16, ncpu // max(1, world)))
    t0 = time.time()
    batch = synth.forge_batch(vk, td, B, seed=1000 + rank, workers=workers, plan=pl)
    t_forge = time.time() - t0

    dp = backend.DevicePlan(pl.to_bytes(), device=local_rank)
    ws = backend.Workspace(dp, B)

    def to_dev(b, dtype=torch.uint8):
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) if b else torch.zeros(1, dtype=torch.uint8, device=dev)

    d_proofs = to_dev(batch.proofs)
    d_off = torch.tensor(batch.proof_off, dtype=torch.int64).to(dev)
    d_inst = to_dev(batch.instances)
    d_ci = to_dev(batch.committed) if batch.committed else None
    d_accept = torch.zeros(B, dtype=torch.uint8, device=dev)
    d_status = torch.zeros(B, dtype=torch.int32, device=dev)
    gathered = [torch.zeros(B, dtype=torch.uint8, device=dev) for _ in range(world)] if (world > 1 and rank == 0) else None
    torch.cuda.synchronize()

    def step():
        stream = torch.cuda.current_stream().cuda_stream
        dp.verify_batch_device(B, d_proofs.data_ptr(), d_off.data_ptr(), d_inst.data_ptr(),
                               d_ci.data_ptr() if d_ci is not None else None, d_accept.data_ptr(), d_status.data_ptr(),
                               ws=ws, stream=stream)
        if world > 1:
            # final accept/reject gather over RCCL (xGMI): B bytes per rank
            dist.gather(d_accept, gathered, dst=0)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel device time over the timed steps (HIP events recorded on the kernels' own streams)
    # (a step launches every kernel once per pipeline chunk; *_ms are sums over the chunks of one step)
    k_steps = min(args.steps, 64)
    acc = {"transcript_combiner": 0.0, "g1_decompress": 0.0, "g1_msm": 0.0, "pairing": 0.0}
    launches = 1
    msm_lpt = 2
    for back in range(k_steps):
        tm = ws.timings(back)
        launches = max(1, tm.launches)
        msm_lpt = tm.msm_lanes_per_term or 2
        acc["transcript_combiner"] += tm.transcript_combiner_ms
        acc["g1_decompress"] += tm.g1_decompress_ms
        acc["g1_msm"] += tm.g1_msm_ms
        acc["pairing"] += tm.pairing_ms
    kernel_ms = {k: v / max(1, k_steps) for k, v in acc.items()}

    accept = d_accept.cpu().numpy()
    n_accept = int(accept.sum())
    ok_all = n_accept == B  # the synthetic batch is 100 % accepting
    if world > 1:
        flag = torch.tensor([1 if ok_all else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok_all = bool(flag.item())

    # second dataset (untimed): 1 % of the proofs get the reference example's byte flip (examples/simple_mul.rs:87-95,
    # first scalar of the proof) - exactly those proofs must be rejected
    reject_check = None
    if rank == 0:
        rej = synth.with_rejects(pl, batch, vk.n_public_inputs, fraction=0.01, seed=77, kinds=["flip_first_scalar"])
        got = dp.verify_batch(rej.proofs, rej.proof_off, rej.instances, rej.committed, ws=ws)
        reject_check = {"fraction": 0.01, "corrupted": rej.expected.count(0), "rejected": int(B - sum(got)),
                        "exactly_the_corrupted_ones": list(got) == rej.expected}

    if rank == 0:
        T = pl.n_terms
        slots = len(pl.points) + pl.n_ci
        # ALGORITHMIC bytes per launch (SURVEY.md §8d / DESIGN.md §4)
        bytes_per_launch = {
            "g1_msm": B * (128 * T + 144),
            "g1_decompress": B * slots * (48 + 96 + 1),
            "transcript_combiner": B * (pl.proof_len + 32 * pl.n_pi + 48 * pl.n_ci + 32 * T + 4),
            "pairing": B * (96 + 144 + 1 + 4) + 2 * 68 * 192,
        }
        # the combiner keeps its Fr register file in LDS when >= 8 proofs per block fit (h2v_capi.hip: vm_lds_slots)
        lds_slots = 64
        while lds_slots >= 8 and pl.n_regs * 32 * lds_slots + 8192 + 1024 > 160 * 1024:
            lds_slots >>= 1
        kname = {"g1_msm": "k_g1_msm_fixed" if msm_lpt == 3 else "k_g1_msm_merged" if msm_lpt == 1 else "k_g1_msm", "g1_decompress": "k_g1_decompress" if (os.environ.get("H2V_DEC_QUEUE") == "0" or os.environ.get("H2V_SPLIT_DEC") == "0") else "k_g1_decompress_queue",
                 "transcript_combiner": "k_transcript_combiner_lds" if lds_slots >= 8 else "k_transcript_combiner",
                 "pairing": "k_pairing_coop" if os.environ.get("H2V_PAIRING") != "legacy" else "k_pairing_check"}

        def pmc_traffic(kernel):
            """HBM bytes per launch (FETCH_SIZE + WRITE_SIZE, KiB on gfx950) from the newest committed PMC summary
            (tools/pmc_summary.py over separate rocprofv3 --pmc passes of this same command); None when absent."""
            import glob
            files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*pmc_summary.txt")))
            if not files:
                return None, None
            vals = {}
            for line in open(files[-1]):
                f = line.split()
                if len(f) >= 4 and f[0] == kernel and f[1] in ("FETCH_SIZE", "WRITE_SIZE"):
                    vals[f[1]] = float(f[3])
            if len(vals) != 2:
                return None, None
            return int((vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), os.path.basename(files[-1])

        def roof(k):
            # one launch handles B / launches proofs; average launch duration = per-step sum / launches
            gbps = bytes_per_launch[k] / (kernel_ms[k] * 1e-3) / 1e9 if kernel_ms[k] > 0 else 0.0
            traffic, src = pmc_traffic(kname[k])
            return {"kernel": kname[k], "bound": "hbm", "achieved": round(gbps, 4), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(gbps / HBM_PEAK_GBPS, 6), "traffic": traffic, "traffic_source": src,
                    "avg_launch_ms": round(kernel_ms[k] / launches, 4), "launches_per_step": launches,
                    "algorithmic_bytes_per_launch": bytes_per_launch[k] // launches}

        # analytical multiply-add counts per launch (lane-level v_mad_u64_u32), see DESIGN.md section 6
        # per MSM lane: 32 windows of 4 doublings + one mixed addition per GLV half the lane carries (tables are built
        # ahead); the launcher reports whether a term ran on two lanes (one half each) or on one (both halves)
        n_fix_terms = sum(1 for kind, _ in pl.terms if kind == PL.TERM_VK_BASE)
        msm_fixed = msm_lpt == 3   # fixed-base mode: VK-base terms cost 65 mixed additions and no doubling
        if msm_fixed:
            msm_lpt = 1
        msm_halves = 2 // msm_lpt
        msm_lane = 128 * MAD_DBL + (32 * msm_halves - 1) * MAD_MADD
        tab_point = 4 * MAD_DBL + 3 * MAD_ADD + 48 * MAD_MUL + 14 * MAD_SQR    # window tables of one point: [1..8]P, normalised, and x beta
        mads = {
            "g1_msm": (B * (T - n_fix_terms) * msm_lane + B * n_fix_terms * 65 * MAD_MADD + B * (T - 1) * MAD_ADD + B * 3 * MAD_MUL) if msm_fixed
                      else B * T * msm_lpt * msm_lane + B * (msm_lpt * T - 1) * MAD_ADD + B * 3 * MAD_MUL,
            "pairing": B * 32 * (35 * (6 * 196 + 196) + 63 * (4 * 196 + 196) + 315 * (2 * 196 + 196) + 136 * (3 * 196 + 196)),   # coop program: MUL / SQR / CSQR / LINE
            "g1_decompress": B * slots * (377 * MAD_SQR + 86 * MAD_MUL + 126 * MAD_DBL + 10 * MAD_ADD + tab_point),
            "transcript_combiner": B * sum(128 for ins in pl.instrs if ins[0] == PL.OP_MUL),
        }

        def int_roof(k):
            tops = mads[k] / (kernel_ms[k] * 1e-3) / 1e12 if kernel_ms[k] > 0 else 0.0
            return {"kernel": kname[k], "bound": "int-mad issue (measured v_mad_u64_u32 ceiling)", "achieved": round(tops, 3),
                    "peak": round(IMAD_PEAK_TOPS, 2) if IMAD_PEAK_TOPS else None, "peak_source": IMAD_PEAK_SOURCE,
                    "unit": "T lane-mad/s", "frac": round(tops / IMAD_PEAK_TOPS, 4) if IMAD_PEAK_TOPS else None,
                    "mads_per_launch": mads[k] // launches}

        dominant = max(kernel_ms, key=kernel_ms.get)
        result = {
            "metric": "halo2_proofs_verified_per_sec",
            "value": round(B * world * args.steps / elapsed, 2),
            "unit": "proofs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": label, "proofs_per_gpu": B, "proof_bytes": pl.proof_len, "msm_terms_per_proof": T,
                       "g1_points_per_proof": slots, "public_inputs": pl.n_pi, "plan_instructions": len(pl.instrs),
                       "parallelism": "independent proofs sharded per GPU; accept gather over RCCL" if world > 1 else "1 GPU"},
            "roofline": roof(dominant),
            "msm_roofline": roof("g1_msm"),
            "int_roofline": int_roof(dominant),
            "msm_int_roofline": int_roof("g1_msm"),
            "kernel_ms": {kname[k]: round(v, 4) for k, v in kernel_ms.items()},
            "pipelines_per_step": launches, "msm_lanes_per_term": msm_lpt,
            "all_accepted": ok_all,
            "reject_dataset": reject_check,
            "forge_seconds": round(t_forge, 2),
        }
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
