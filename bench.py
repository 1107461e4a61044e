#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): Gauss-Newton iterations/s + converged systems/s on a batch of
100 000 independent 32-constraint f64 sketches ("ring16", cfg3) per GPU.

    python bench.py --gpus N --steps K --warmup W

One *step* = one pass of the hot path over the HBM-resident batch: `fx_system_solve_device`
(== fiksi `System::solve(SolvingOptions::DEFAULT)` for every sketch: RMS scaling, LCG perturbation,
Jacobian assembly, damped Gauss-Newton / Levenberg-Marquardt to convergence, write-back, residual
check), every step starting from the same start values. Inputs are uploaded before the timed region.

N > 1: one process per GPU (torchrun / torch.distributed, backend nccl == RCCL); each rank solves
its own 100k-system shard (weak scaling, no data-path collective); RCCL is used only for the
barrier, the max-over-ranks time and the sum of the throughput counters.
`--scaling strong` is BASELINE configs[3] (cfg4) as written instead: ONE 100k-system batch, rank r solves its
contiguous shard of 100k / N systems (workloads.shard). At N = 1 the line also carries `cfg4_prediction`: the
measured time of the 1/2, 1/4 and 1/8 shards on this GPU, i.e. the strong-scaling curve before an 8-GPU node
shows up.

Prints ONE JSON line (rank 0). Besides the contract fields it carries
  roofline      — the Jacobian-assembly kernel (K1, `eval_rows_kernel<true>`), the HBM-bound kernel the
                  north-star prices against the 8 TB/s roofline, ON A 500k-SYSTEM BATCH whose footprint (0.97 GB
                  moved per launch) is 3.8x the 256 MiB Infinity Cache. `achieved` / `frac` = the bytes a launch
                  MOVES (a one-structure batch reads its structure from the first System: 1920 B per ring16
                  evaluation, confirmed by `traffic`, the PMC counters) / the median launch time from HIP events on
                  the launch stream. SURVEY §8d's 2560 B figure over the same time is the labelled side key
                  `by_survey_8d_bytes` (rounds 1-3 quoted it; it counts bytes that no longer move);
                  `mixed_structure` = a 500k batch of two interleaved structures, which streams all 2560 B;
                  `in_cache` = the same kernel on the timed 100k batch (under the Infinity Cache: not an HBM rate)
  solve_kernel  — the fused per-system solve kernel that the timed region consists of (latency /
                  f64-VALU bound by construction; its HBM traffic is ~1.5 KB per system)
  step_solvers  — what the other LM step solvers cost on the same resident batch: FX_STEP_CHOLESKY_REFINED and
                  FX_STEP_QR (the reference's numerics, bit-identical paths)
  large_systems — Systems beyond one wavefront (fx_sparse_team.h): cfg2 (one 5 000-point sketch) as a resident
                  solve, the reference's 64-triangle sketch as a batch of 256
  reference_bench_group — the reference's own criterion group (fiksi_bench.rs:46-73: hinged triangles, sizes
                  1 / 4 / 16 / 64) as batches and as single System::solve latency, each beside the oracle
  host_path     — fx_system_solve_batch end to end (host analysis + upload + solve + download)
  cpu_baseline  — the CPU oracle (C++ restatement of the reference algorithm: COO->CSC, COLAMD,
                  sparse Householder QR LM) on the same systems, on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SYSTEMS_PER_GPU = 100_000
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
F64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X f64 vector (non-matrix) peak
FLOPS_PER_TRIAL = 32 ** 3 / 3 + 2 * 32 ** 2 + 2 * 408 + 60 * 32  # ring16 component, see solve_kernel below


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--systems", type=int, default=SYSTEMS_PER_GPU, help="systems per GPU (default: the BASELINE config)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --systems per GPU (default); strong: --systems in total, split over the GPUs (cfg4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick", action="store_true", help="the headline + roofline only (no side workloads)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="systems in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--host-threads", type=int, default=0,
                    help="host threads of the sparse path for batches of several large Systems (0 = library default 8; the "
                         "profiled runs of tools/collect_profiles.sh use 1: rocprofv3 --kernel-trace crashes under concurrent launches)")
    return ap.parse_args()


def relaunch_under_torchrun(args) -> int:
    """--gpus N > 1 without a torchrun environment: start the ranks as child processes (never exec
    from a process that may touch the GPU)."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main() -> int:
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        return relaunch_under_torchrun(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np

    if world > 1 or os.environ.get("FIKSI_BENCH_FORCE_DIST") == "1":
        import torch  # noqa: F401  (before fiksi_amd: one shared HIP runtime per process, see fiksi_amd/_lib.py)

    import __graft_entry__ as graft

    graft.build()
    import fiksi_amd
    from fiksi_amd import abi, distributed, workloads

    dist = None
    torch = None
    reduce_device = None
    # FIKSI_BENCH_FORCE_DIST=1 (rehearsal): bring up the process group even for one rank, so that the RCCL
    # code path (init, barrier, all-reduce on device tensors) can be exercised on a 1-GPU box
    force_dist = os.environ.get("FIKSI_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # Rehearsal knobs (not used by the driver): FIKSI_BENCH_BACKEND=gloo runs the N>1 code path
        # with CPU reductions, FIKSI_BENCH_DEVICE=0 puts every rank on one GPU (a 1-GPU box).
        backend = os.environ.get("FIKSI_BENCH_BACKEND", "nccl")
        if "FIKSI_BENCH_DEVICE" in os.environ:
            local_rank = int(os.environ["FIKSI_BENCH_DEVICE"])
        # a launcher that masks devices per rank (HIP_VISIBLE_DEVICES) leaves each rank one device, index 0
        n_visible = torch.cuda.device_count()
        if n_visible and local_rank >= n_visible:
            local_rank = local_rank % n_visible
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
        reduce_device = "cuda" if backend == "nccl" else None

    # ---- inputs: this rank's shard. weak scaling: a full cfg3 batch per GPU, distinct seeds; strong scaling
    # (cfg4): the one batch of --systems sketches, rank r takes its contiguous shard
    if args.scaling == "strong":
        batch = workloads.shard(workloads.ring16(args.systems, seed0=1000), rank, world)
        n_sys = len(batch["var_off"]) - 1
    else:
        n_sys = args.systems
        batch = workloads.ring16(n_sys, seed0=distributed.rank_seed(1000, rank, n_sys))
    ctx = fiksi_amd.Context(local_rank)
    if args.host_threads:
        ctx.set_host_threads(args.host_threads)
    db = ctx.upload(batch)
    opts = abi.solving_opts()  # SolvingOptions::DEFAULT + the reference LM constants

    def barrier():
        ctx.synchronize()
        if dist is not None:
            if reduce_device == "cuda":
                torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        db.system_solve(opts)
    barrier()

    # ---- timed region: exactly K steps -----------------------------------------------------
    t0 = time.perf_counter()
    ctx.timer_begin()
    for _ in range(args.steps):
        db.system_solve(opts)
    kernel_ms = ctx.timer_end()  # HIP events on the launch stream; synchronizes it
    barrier()
    elapsed = time.perf_counter() - t0

    res = db.get_results()
    converged = int(np.count_nonzero(res["sse_unscaled"] < 1e-4))  # fiksi_bench.rs:65-72
    accepted = int(res["accepted"].sum())
    trials = int(res["trials"].sum())
    trials_per_step = trials  # this rank's launch (the sums below are over all ranks)

    per_rank_s = distributed.gather_times(dist, elapsed, device=reduce_device)
    per_rank_kernel_ms = distributed.gather_times(dist, kernel_ms / args.steps, device=reduce_device)
    per_rank_systems = distributed.gather_times(dist, float(n_sys), device=reduce_device)
    elapsed, (converged, accepted, trials, total_sys) = distributed.reduce_throughput(
        dist, elapsed, [converged, accepted, trials, n_sys], device=reduce_device)

    # ---- K1 (Jacobian assembly) on the same resident batch, HIP-event timed ----------------
    k1_ms, _, k1_launches = _time_k1(ctx, db)
    k1_bytes = workloads.k1_algorithmic_bytes(batch, db.nnz, one_structure=True)  # 1920 B x systems for ring16 (one structure)
    k1_gbs = k1_bytes / (k1_ms * 1e-3) / 1e9

    out = None
    if rank == 0:
        steps = args.steps
        solve_ms = kernel_ms / steps
        # fused kernel: algorithmic HBM bytes per system (SURVEY §8d B_solve) and flop estimate
        nv_tot = int(batch["var_off"][-1])
        ne_tot = int(batch["expr_off"][-1])
        b_solve = 8 * nv_tot + 28 * ne_tot + nv_tot + 8 * nv_tot + 32 * n_sys
        out = {
            "metric": "Gauss-Newton iters/sec + converged systems/sec, 100k×32-constraint f64 batch",
            "value": converged * steps / elapsed,
            "unit": "converged systems/s",
            "gn_iters_per_sec": accepted * steps / elapsed,
            "lm_trials_per_sec": trials * steps / elapsed,
            "systems_per_sec": total_sys * steps / elapsed,
            "converged_fraction": converged / total_sys,
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / steps,
            # every rank's own clock around the K steps (the value above divides by the slowest), its solve kernel's mean
            # launch time by HIP events, and its share of the Systems — both --scaling modes
            "per_rank_ms": [round(s * 1e3 / steps, 4) for s in per_rank_s],
            "per_rank_kernel_ms": [round(m, 4) for m in per_rank_kernel_ms],
            "per_rank_systems": [int(x) for x in per_rank_systems],
            "slowest_rank": int(max(range(len(per_rank_s)), key=lambda r: per_rank_s[r])),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "cfg3 ring16: independent sketches of 16 points / 32 variables / 32 expressions "
                            "(16 ring + 8 chord distances, 8 three-point angles, 144 Jacobian non-zeros), "
                            "System::solve(SolvingOptions::DEFAULT), inputs resident in HBM",
                "systems_per_gpu": n_sys,
                "global_systems": total_sys,
                "parallelism": f"dp{world} (independent systems sharded, no data-path collective)",
                "device": ctx.name(),
            },
            "roofline": None,  # filled below: the HBM figure needs the 500k-System batch (weak scaling, N = 1) or falls back
            "solve_kernel": {
                "kernel": ("lm_solve_grouped_c_kernel (fused scale+perturb+assembly+LM+write-back, four Systems per wavefront: one per "
                           "DPP row; the build for batches of one structure — lists shared by the wavefront, Jt J by its pattern, "
                           "two wavefronts per SIMD: fx_grouped_c.hip)") if db.grouped_build() == 1 else
                          ("lm_solve_grouped_kernel<2 columns per lane, f64> (fused scale+perturb+assembly+LM+write-back, four "
                           "Systems per wavefront: one per DPP row, fx_grouped.hip)") if db.solve_route() == 1 else
                          "lm_solve_kernel<32> (fused scale+perturb+assembly+LM+write-back, one wavefront per system)",
                "avg_launch_ms": solve_ms,
                "algorithmic_hbm_bytes_per_launch": b_solve,
                "achieved_GBps": b_solve / (solve_ms * 1e-3) / 1e9,
                "frac_of_hbm_peak": b_solve / (solve_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                # useful f64 flops of one LM trial on a 32 x 32 component: Cholesky n^3/3, two triangular solves 2 n^2,
                # JtJ over the triangle's 408 products, ~60 per expression row — against the f64 vector peak
                "algorithmic_flops_per_launch": FLOPS_PER_TRIAL * trials_per_step,
                "achieved_TFLOPs": FLOPS_PER_TRIAL * trials_per_step / (solve_ms * 1e-3) / 1e12,
                "frac_of_f64_vector_peak": FLOPS_PER_TRIAL * trials_per_step / (solve_ms * 1e-3) / 1e12 / F64_VECTOR_PEAK_TFLOPS,
                "includes": "the scout pass + chunk ranking of fx_ctx_set_presort (most-work-first hand-out, two launches, ~0.035 ms; part of every step)",
                "note": "latency/f64-VALU bound by construction (~14 kflop per LM trial on a serial "
                        "Cholesky dependency chain); HBM is not its roof (SURVEY.md §7, §8d). "
                        "FIKSI_AMD_GROUPED=0 times the one-System-per-wavefront kernel instead, FIKSI_AMD_GROUPED_C=0 the "
                        "grouped kernel's general build (one wavefront per SIMD)",
            },
        }
        in_cache = {
            "systems": n_sys, "achieved": k1_gbs, "frac": k1_gbs / HBM_PEAK_GBS, "avg_launch_ms": k1_ms, "launches": k1_launches,
            "algorithmic_bytes_per_launch": k1_bytes, "traffic": pmc_traffic("eval_rows_kernel<true", n_sys),
            "note": "194 MB per launch: under the 256 MiB Infinity Cache, so this is not an HBM rate",
        }
        in_cache["frac_by_counter_bytes"] = None if in_cache["traffic"] is None else in_cache["traffic"] / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        roof = {
            "kernel": "eval_rows_kernel<true> (K1 Jacobian assembly: residuals + CSR J values; "
                      "fx_eval_residual_jacobian_device on a resident batch)",
            "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "traffic_source": "profiles/round5_pmc_traffic*.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                              "passes; read bytes = 2 x FETCH_SIZE KiB per the gfx950 correction, checked on a "
                              "known-byte kernel with K1's access widths: tools/probes/fetch_calib.hip)",
        }
        if world == 1 and args.scaling == "weak":
            roof.update(k1_past_l3(ctx, workloads))
        else:  # (N > 1 or the strong-scaling mode: the timed batch's own figure, flagged)
            roof.update({"achieved": k1_gbs, "frac": k1_gbs / HBM_PEAK_GBS, "traffic": in_cache["traffic"], "workload_systems": n_sys,
                         "avg_launch_ms": k1_ms, "algorithmic_bytes_per_launch": k1_bytes,
                         "note": "in-cache figure (the 500k-System HBM measurement runs at N = 1, weak scaling)"})
        roof["in_cache"] = in_cache
        out["roofline"] = roof
        out["solve_kernel"].update(sq_counters("lm_solve_grouped", FLOPS_PER_TRIAL * trials_per_step))
        if world == 1 and not args.quick:
            out["host_path"] = host_path(ctx, batch, np)
            out["step_solvers"] = step_solvers(ctx, db, abi, np, n_sys, solve_ms)
            out["cfg4_prediction"] = cfg4_prediction(ctx, abi, workloads, batch, solve_ms)
            out["other_workloads"] = other_workloads(ctx, abi, workloads, np, n_sys)
            out["large_systems"] = large_systems(ctx, abi, workloads, np)
            out["reference_bench_group"] = reference_bench_group(ctx, abi, workloads, np)
            out["decomposers_single_triangle"] = decomposers_single_triangle(ctx)
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(batch, args.cpu_sample)
        print(json.dumps(out), flush=True)

    db.free()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0



def _time_k1(ctx, db, groups: int = 10, per_group: int = 6):
    """K1 launch time by HIP events on the launch stream: `groups` timed runs of `per_group` back-to-back launches after a
    warm-up (no cold launch in any of them). Returns (median of the runs' per-launch times, their mean, launches timed)."""
    import numpy as np

    for _ in range(3):
        db.eval_residual_jacobian(0)
    ctx.synchronize()
    ms = []
    for _ in range(groups):
        ctx.timer_begin()
        for _ in range(per_group):
            db.eval_residual_jacobian(0)
        ms.append(ctx.timer_end() / per_group)
    return float(np.median(ms)), float(np.mean(ms)), groups * per_group


def k1_past_l3(ctx, workloads, n_big: int = 500_000):
    """K1 on a batch whose per-launch traffic (500k ring16 sketches: 0.97 GB moved) is 3.8x the 256 MiB Infinity Cache:
    nothing one launch reads can still be on the die from the launch before. This is the line's `roofline`.
    `achieved` / `frac` count the bytes that MOVE: the headline batch is one sketch with many parameter sets, and such a
    batch reads kinds and fields from its first System (fx_eval.hip) — 1920 B per System, not SURVEY 8d's 2560 (round 3
    quoted the 2560-byte figure, which the kernel no longer moves: a fraction above what HBM delivers). The 8d figure
    stays as the labelled side key `by_survey_8d_bytes`; `mixed_structure` is the same measurement on a batch of two
    interleaved structures, which does stream all 2560 B."""
    b = workloads.ring16(n_big, seed0=5_000_000)
    db = ctx.upload(b)
    ms, ms_mean, launches = _time_k1(ctx, db)
    nbytes = workloads.k1_algorithmic_bytes(b, db.nnz, one_structure=True)
    nbytes_8d = workloads.k1_algorithmic_bytes(b, db.nnz)
    gbs = nbytes / (ms * 1e-3) / 1e9
    traffic = pmc_traffic("eval_rows_kernel<true", n_big)
    db.free()
    out = {
        "achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "workload_systems": n_big, "avg_launch_ms": ms,
        "avg_launch_ms_mean": ms_mean, "launches": launches,
        "timing": "median over 10 runs of 6 back-to-back launches each, HIP events on the launch stream, after 3 warm-up launches",
        "algorithmic_bytes_per_launch": nbytes,
        "algorithmic_bytes_per_system": nbytes // n_big,
        "bytes_model": "one-structure batch: 8 nv (x) + 8 m (parameters) + 8 m (r) + 8 nnz (J values); kinds and fields come from the "
                       "first System and stay in L1/L2",
        "frac_by_counter_bytes": None if traffic is None else traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "frac_of_streaming_copy_rate": gbs / 6290.0,  # what a plain copy kernel reaches from HBM on this part (round-2 calibration)
        "by_survey_8d_bytes": {"bytes_per_launch": nbytes_8d, "bytes_per_system": nbytes_8d // n_big,
                               "achieved": nbytes_8d / (ms * 1e-3) / 1e9, "frac": nbytes_8d / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "note": "SURVEY 8d's 2560 B per System over the same time: 25 % of these bytes are not moved by a "
                                       "one-structure batch — kept for comparison with rounds 1-3, not a rate"},
        "note": "500k Systems, 3.8x the Infinity Cache; the in-cache figure of the timed 100k batch is under in_cache",
    }
    bm = workloads.ring16_two_structures(n_big, seed0=5_000_000)
    dbm = ctx.upload(bm)
    msm, msm_mean, _ = _time_k1(ctx, dbm)
    nb_m = workloads.k1_algorithmic_bytes(bm, dbm.nnz)
    tr_m = pmc_traffic("eval_rows_kernel<true", n_big, mixed=True)
    out["mixed_structure"] = {
        "workload": "500k ring16 sketches of two structures, interleaved (workloads.ring16_two_structures): every System's kinds and "
                    "fields stream from HBM",
        "avg_launch_ms": msm, "avg_launch_ms_mean": msm_mean, "algorithmic_bytes_per_launch": nb_m,
        "algorithmic_bytes_per_system": nb_m // n_big, "achieved": nb_m / (msm * 1e-3) / 1e9,
        "frac": nb_m / (msm * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": tr_m,
        "frac_by_counter_bytes": None if tr_m is None else tr_m / (msm * 1e-3) / 1e9 / HBM_PEAK_GBS,
    }
    dbm.free()
    return out


def _time_solves(ctx, db, opts, reps=3):
    db.system_solve(opts)  # warm-up (plans of the mode are built here)
    ctx.synchronize()
    ctx.timer_begin()
    for _ in range(reps):
        db.system_solve(opts)
    return ctx.timer_end() / reps


def step_solvers(ctx, db, abi, np, n_sys, default_ms):
    """The same resident batch with the other LM step solvers (include/fiksi_amd.h: fx_step_solver)."""
    out = {"cholesky": {"ms_per_step": default_ms, "note": "the headline: plain normal equations, grouped kernel"}}
    # NOT the headline: the same solve with the Systems handed out longest-first, by the trial counts of the previous
    # solve of this resident batch (fx_batch_schedule_by_last_solve; results bit-identical)
    db.schedule_by_last_solve(True)
    out["cholesky"]["ms_per_step_history_scheduled"] = _time_solves(ctx, db, abi.solving_opts(), reps=5)
    db.schedule_by_last_solve(False)
    for name, solver in (("cholesky_refined", 1), ("qr_reference_numerics", 2)):
        ms = _time_solves(ctx, db, abi.solving_opts(solver=solver), reps=2)
        res = db.get_results()
        conv = int(np.count_nonzero(res["sse_unscaled"] < 1e-4))
        out[name] = {"ms_per_step": ms, "converged_systems_per_sec": conv / (ms * 1e-3), "converged_fraction": conv / n_sys,
                     "lm_trials": int(res["trials"].sum()), "cost_vs_headline": ms / default_ms}
    out["qr_reference_numerics"]["note"] = ("solvi's sparse Householder QR replayed operation by operation (COLAMD order from the host), "
                                            "correctly rounded atan2: every iterate bit-identical to the reference algorithm")
    return out


def cfg4_prediction(ctx, abi, workloads, batch, full_ms):
    """Strong scaling (BASELINE configs[3]) predicted from one GPU: EVERY 1/N shard of the same batch is timed here, one
    after the other — an N-GPU run takes as long as its slowest shard; efficiency = t(1) / (N max_r t(shard r)).
    (Rounds 2-3 timed shard 0 only, which is not the slowest: `ms_shard0` keeps that figure for comparison.)"""
    out = {"systems_total": len(batch["var_off"]) - 1, "ms_1_gpu": full_ms, "shards": {}}
    for n in (2, 4, 8):
        ts = []
        for r in range(n):
            sh = workloads.shard(batch, r, n)
            db = ctx.upload(sh)
            ts.append(_time_solves(ctx, db, abi.solving_opts(), reps=5))
            db.free()
        ms = max(ts)
        out["shards"][str(n)] = {"systems": len(batch["var_off"]) // n, "ms_per_step": ms, "ms_shard0": ts[0], "ms_all_shards": ts,
                                 "predicted_speedup": full_ms / ms, "predicted_efficiency": full_ms / (n * ms)}
    return out


def _build_name(db):
    """Which kernel a resident batch's default solve runs (fx_debug_grouped_build / fx_debug_solve_route)."""
    b = db.grouped_build()
    if b == 2:
        return "lm_solve_grouped_s_kernel (one structure, sparse factor: level-scheduled Cholesky over tables in LDS, fx_grouped_s.hip)"
    if b == 4:
        return "lm_solve_tiny_kernel (one structure of at most 8 variables: eight lanes per System, eight Systems per wavefront, fx_grouped_tiny.hip)"
    if b == 1:
        return "lm_solve_grouped_c*_kernel (one structure: lists shared per wavefront, Jt J by its pattern, fx_grouped_c.hip)"
    if b == 0:
        return "lm_solve_grouped_kernel (four Systems per wavefront, fx_grouped.hip)"
    return "one wavefront per System / wide / team kernels (by size)"


def reference_bench_group(ctx, abi, workloads, np):
    """fiksi/benches/fiksi_bench.rs:46-73: `solve/hinged_triangles`, sizes 1, 4, 16, 64 (6 / 18 / 66 / 258 variables).
    The reference measures the latency of ONE System::solve per size; here each size is reported (a) as that
    latency through the builder API (System::solve on a fresh upload: host flattening + analysis + upload + solve +
    download), next to the oracle's single-thread time for the same solve, and (b) as a resident batch."""
    import time as _t

    import fiksi_amd
    from oracle import oracle as O

    out = {}
    for n_tri, n_batch in ((1, 100_000), (4, 100_000), (16, 20_000), (64, 256)):
        entry = {"variables": 2 + 4 * n_tri, "constraints": 3 * n_tri}
        # (a) one System::solve through the builder API, values reset between solves as the reference bench does
        s = fiksi_amd.System()
        hinge = fiksi_amd.elements.Point.create(s, 0., 0.)
        pts = [(hinge, 0., 0.)]
        for t in range(n_tri):
            p1 = fiksi_amd.elements.Point.create(s, -1., float(t))
            p2 = fiksi_amd.elements.Point.create(s, 1., float(t))
            pts += [(p1, -1., float(t)), (p2, 1., float(t))]
            fiksi_amd.constraints.PointPointDistance.create(s, hinge, p1, 2.)
            fiksi_amd.constraints.PointPointDistance.create(s, hinge, p2, 2.)
            fiksi_amd.constraints.PointPointDistance.create(s, p1, p2, 3.)
        reps = 30 if n_tri < 64 else 8
        s.solve(ctx=ctx)
        dt = 0.0
        for _ in range(reps):
            for h, x, y in pts:
                h.update_value(s, x, y)
            t0 = _t.perf_counter()
            s.solve(ctx=ctx)
            dt += _t.perf_counter() - t0
        r = s.constraint_residuals(ctx=ctx)
        entry["single_solve_ms"] = dt / reps * 1e3
        entry["single_solve_sse"] = float((r * r).sum())  # fiksi_bench.rs:65-72 asserts < 1e-4
        one = workloads.hinged_triangles(1, n_tri)
        O.solve_batch(one, mode=3)
        t0 = _t.perf_counter()
        for _ in range(reps):
            O.solve_batch(one, mode=3)
        entry["single_solve_ms_oracle_1_thread"] = (_t.perf_counter() - t0) / reps * 1e3
        # (b) a resident batch of the same sketch
        b = workloads.hinged_triangles(n_batch, n_tri)
        db = ctx.upload(b)
        ms = _time_solves(ctx, db, abi.solving_opts(), reps=2)
        res = db.get_results()
        conv = int(np.count_nonzero(res["sse_unscaled"] < 1e-4))
        entry["batch"] = {"systems": n_batch, "ms_per_step": ms, "kernel": _build_name(db), "converged_systems_per_sec": conv / (ms * 1e-3),
                          "converged_fraction": conv / n_batch, "triangles_per_sec": conv * n_tri / (ms * 1e-3)}
        db.free()
        out[f"hinged_triangles_{n_tri}"] = entry
    return out


def decomposers_single_triangle(ctx):
    """The reference's `single_triangle` test (fiksi/src/tests/triangles.rs:10-37) as a latency figure: one System::solve
    through the builder under each Decomposer; RMS of the constraint residuals as the test asserts it (< 1e-4)."""
    import time as _t

    import fiksi_amd as F

    out = {}
    for dec in (F.Decomposer.NONE, F.Decomposer.SinglePass, F.Decomposer.RecursiveAssembly):
        s = F.System()
        pts = [F.elements.Point.create(s, x, y) for x, y in ((0., 0.), (1., .5), (2., 1.))]
        for i, j in ((0, 1), (0, 2), (1, 2)):
            F.constraints.PointPointDistance.create(s, pts[i], pts[j], 1.)
        opts = F.SolvingOptions(decomposer=dec)
        s.solve(opts, ctx)
        dt, reps = 0.0, 10
        for _ in range(reps):
            for h, (x, y) in zip(pts, ((0., 0.), (1., .5), (2., 1.))):
                h.update_value(s, x, y)
            t0 = _t.perf_counter()
            s.solve(opts, ctx)
            dt += _t.perf_counter() - t0
        r = s.constraint_residuals(ctx)
        out[dec.name] = {"solve_ms": dt / reps * 1e3, "rms_residual": float((r * r).mean() ** 0.5),
                         "device_solves": int(s.last_result["ncomp"])}
    return out


def host_path(ctx, batch, np):
    """fx_system_solve_batch on host buffers: host analysis + upload + solve + download (PCIe inclusive; never `value`).
    The C entry point itself, on buffers that stay where they are (the variables are refilled outside the timed region):
    the Python mirror's own copy of the variables — 26 MB from a fresh mapping per call — is not the library's time."""
    import ctypes as C
    import time as _t

    from fiksi_amd import abi
    from fiksi_amd._lib import check, lib

    n = len(batch["var_off"]) - 1
    a = abi.normalize_batch(batch)
    start = a["vars"].copy()
    a["vars"] = start.copy()
    res = np.zeros(n, dtype=abi.RESULT_DTYPE)
    o = abi.solving_opts()
    st = abi.as_struct(a)

    def run():
        times = []
        for k in range(8):  # the first call is the warm-up; median of seven
            a["vars"][:] = start
            t0 = _t.perf_counter()
            check(lib.fx_system_solve_batch(ctx.handle, C.byref(st), C.byref(o), res.ctypes.data), "fx_system_solve_batch")
            if k:
                times.append(_t.perf_counter() - t0)
        dt = sorted(times)[len(times) // 2]
        conv = int(np.count_nonzero(res["sse_unscaled"] < 1e-4))
        return {"ms_per_call": dt * 1e3, "ms_per_call_min": min(times) * 1e3, "ms_per_call_max": max(times) * 1e3, "calls": len(times),
                "converged_systems_per_sec": conv / dt}

    out = {"entry_point": "fx_system_solve_batch", "systems": n}
    out.update(run())  # pageable buffers, no hint: what a caller gets who does nothing
    plain_bits = (a["vars"].copy(), res.copy())
    # ... and with the caller's help (fx_host_register on the arrays that travel, FX_HINT_ONE_STRUCTURE): the hint is verified
    # against every System beside the device's work (fx_analyze.cpp: verify_one_structure), the results must be the same bits
    ctx.host_register(a["vars"], a["expr_param"], res)
    ctx.set_batch_hints(one_structure=True)
    try:
        helped = run()
    finally:
        ctx.set_batch_hints(one_structure=False)
        ctx.host_unregister(a["vars"], a["expr_param"], res)
    helped["same_bits_as_the_plain_call"] = bool(np.array_equal(plain_bits[0].view(np.uint64), a["vars"].view(np.uint64)) and
                                                 plain_bits[1].tobytes() == res.tobytes())
    out["registered_buffers_and_one_structure_hint"] = helped
    return out


def other_workloads(ctx, abi, workloads, np, n_sys: int):
    """Reported beside the headline, never part of `value`: the reference's own bench generator
    (`add_hinged_triangles(n = 11)`, fiksi_bench.rs:15-40: 33 distance constraints, 46 variables) as a
    batch of the same size, with Decomposer::None and Decomposer::SinglePass, and cfg5's per-GPU share in
    f32; N = 1 runs only."""
    out = {}
    b = workloads.hinged_triangles(n_sys, 11)
    db = ctx.upload(b)
    for label, opts in (("none", abi.solving_opts()), ("single_pass", abi.solving_opts(decomposer=1))):
        db.system_solve(opts)  # warm-up (SinglePass: builds the block plan)
        ctx.synchronize()
        reps = 3
        ctx.timer_begin()
        for _ in range(reps):
            db.system_solve(opts)
        ms = ctx.timer_end() / reps
        res = db.get_results()
        conv = int(np.count_nonzero(res["sse_unscaled"] < 1e-4))
        out[f"hinged_triangles_11_{label}"] = {
            "systems": n_sys, "ms_per_step": ms, "converged_systems_per_sec": conv / (ms * 1e-3),
            "converged_fraction": conv / n_sys, "gn_iters_per_sec": int(res["accepted"].sum()) / (ms * 1e-3),
        }
    db.free()
    # cfg5's per-GPU share: 125 000 inconsistent (over-constrained least-squares) ring sketches, f32
    b5 = workloads.ring16(125_000, inconsistent=True)
    db = ctx.upload(b5)
    o32 = abi.solving_opts(f32=True)
    db.system_solve(o32)
    ctx.synchronize()
    ctx.timer_begin()
    for _ in range(3):
        db.system_solve(o32)
    ms = ctx.timer_end() / 3
    res = db.get_results()
    settled = int(np.count_nonzero(res["exit"] <= 2))  # SSE / step / ftol exits (cfg5 "converged")
    out["cfg5_share_f32"] = {
        "systems": 125_000, "dtype": "f32", "ms_per_step": ms, "settled_systems_per_sec": settled / (ms * 1e-3),
        "settled_fraction": settled / 125_000, "gn_iters_per_sec": int(res["accepted"].sum()) / (ms * 1e-3),
        "lm_trials": int(res["trials"].sum()),
    }
    db.schedule_by_last_solve(True)
    out["cfg5_share_f32"]["ms_per_step_history_scheduled"] = _time_solves(ctx, db, o32)
    db.schedule_by_last_solve(False)
    ms64 = _time_solves(ctx, db, abi.solving_opts())
    res = db.get_results()
    out["cfg5_share_f64_same_batch"] = {"systems": 125_000, "dtype": "f64", "ms_per_step": ms64, "lm_trials": int(res["trials"].sum()),
                                        "settled_fraction": int(np.count_nonzero(res["exit"] <= 2)) / 125_000}
    db.free()
    # the headline's sketches in two structures, interleaved (a batch of a few sketches with many parameter sets each): the big
    # structure classes in one launch of the one-structure build
    b2 = workloads.ring16_two_structures(n_sys)
    db = ctx.upload(b2)
    ms2 = _time_solves(ctx, db, abi.solving_opts())
    res = db.get_results()
    conv = int(np.count_nonzero(res["sse_unscaled"] < 1e-4))
    out["ring16_two_structures"] = {"systems": n_sys, "ms_per_step": ms2, "converged_systems_per_sec": conv / (ms2 * 1e-3),
                                    "converged_fraction": conv / n_sys, "grouped_build": db.grouped_build()}
    # ... and with the reference's own numerics (FX_STEP_QR): the grouped QR build once per structure class (fx_solve.cpp: launch_class_qr)
    out["ring16_two_structures"]["ms_per_step_qr_reference_numerics"] = _time_solves(ctx, db, abi.solving_opts(solver=2), reps=2)
    db.free()
    # ... and with every System's structure its own (random chords and angle sites): no one-structure program, no structure class —
    # the general build of the grouped kernel, the rate a batch of unrelated sketches gets
    b3 = workloads.ring16_all_different(n_sys)
    db = ctx.upload(b3)
    ms3 = _time_solves(ctx, db, abi.solving_opts())
    res = db.get_results()
    conv = int(np.count_nonzero(res["sse_unscaled"] < 1e-4))
    out["ring16_all_different"] = {"systems": n_sys, "ms_per_step": ms3, "converged_systems_per_sec": conv / (ms3 * 1e-3),
                                   "converged_fraction": conv / n_sys, "kernel": _build_name(db), "lm_trials": int(res["trials"].sum())}
    db.free()
    return out


def pmc_traffic(kernel_substr: str, n_sys: int, mixed: bool = False):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC summaries (collected in separate
    --pmc passes on the same workload; one file per batch size); None when no summary covers this size."""
    names = ("round5_pmc_traffic_500k_mixed.json", "round4_pmc_traffic_500k_mixed.json") if mixed else (
        "round5_pmc_traffic.json", "round5_pmc_traffic_500k.json", "round4_pmc_traffic.json", "round4_pmc_traffic_500k.json", "round3_pmc_traffic.json", "round3_pmc_traffic_500k.json",
        "round2_pmc_traffic.json", "round2_pmc_traffic_500k.json", "round1_pmc_traffic.json")
    for name in names:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            if int(d.get("n_systems", -1)) != int(n_sys):
                continue
            for kname, v in d["kernels"].items():
                if kernel_substr in kname:
                    return int(v["hbm_bytes_per_launch"])
        except Exception:
            continue
    return None


def sq_counters(kernel_substr: str, useful_flops: float):
    """VALU instructions the solve kernel issues per useful full-width FMA, from the committed SQ counter summary
    (rocprofv3 --pmc SQ_INSTS_VALU ..., its own pass on the same 100k batch): a wave64 f64 FMA is 64 lanes x 2 flop."""
    for name in ("round5_pmc_sq.json", "round4_pmc_sq.json", "round3_pmc_sq.json", "round2_pmc_sq.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            for kname, v in d["kernels"].items():
                if kernel_substr in kname:
                    useful = useful_flops / 128.0
                    return {"valu_insts_per_launch": v["SQ_INSTS_VALU"], "useful_full_width_fmas_per_launch": useful,
                            "valu_insts_per_useful_fma": v["SQ_INSTS_VALU"] / useful,
                            "valu_active_fraction_of_wave_lifetime": v.get("valu_active_fraction_of_wave_lifetime"),
                            "sq_counter_source": "profiles/" + name}
        except Exception:
            continue
    return {}


def large_systems(ctx, abi, workloads, np):
    """Systems beyond one wavefront, on the workgroup kernels of the sparse path."""
    out = {}
    b = workloads.large_sketch(5000)
    db = ctx.upload(b)
    for name, solver in (("cholesky", 0), ("cholesky_refined", 1)):
        ms = _time_solves(ctx, db, abi.solving_opts(solver=solver), reps=5)
        res = db.get_results()
        out.setdefault("cfg2_one_5000_point_sketch", {})[name] = {
            "ms_per_solve": ms, "accepted": int(res["accepted"][0]), "trials": int(res["trials"][0]), "exit": int(res["exit"][0]),
            "sse": float(res["sse"][0])}
    db.free()
    out["cfg2_one_5000_point_sketch"]["note"] = ("resident batch, plan kept; the oracle takes 16 accepted steps / 89 trials on this sketch "
                                                 "(tests/golden/cfg2_oracle.json) and ~85 s")
    b = workloads.hinged_triangles(256, 64)
    db = ctx.upload(b)
    ms = _time_solves(ctx, db, abi.solving_opts(), reps=5)
    res = db.get_results()
    conv = int(np.count_nonzero(res["sse_unscaled"] < 1e-4))
    out["hinged_triangles_64_x_256"] = {"ms_per_step": ms, "converged_systems_per_sec": conv / (ms * 1e-3), "converged_fraction": conv / 256,
                                        "kernel": "mf_lm_solo_kernel (fx_front.h): one workgroup per System, the whole LM loop in one launch, the factor by fronts of up to 15 columns in DPP rows"}
    db.free()
    return out


def cpu_baseline(batch, sample: int):
    """The oracle (reference algorithm restated in C++) on a bounded sample of the same systems."""
    import numpy as np

    from fiksi_amd import workloads
    from oracle import oracle as O

    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # a 1-GPU box's CPU share is 16 cores; FIKSI_CPU_THREADS overrides
    cores = int(os.environ.get("FIKSI_CPU_THREADS", min(cores, 16)))
    n_total = len(batch["var_off"]) - 1
    n = sample if sample > 0 else n_total  # 100k systems ~ 5-10 CPU-seconds of reference-algorithm work
    sub = workloads.shard(batch, 0, max(1, n_total // n)) if n < n_total else batch
    n = len(sub["var_off"]) - 1
    t = time.perf_counter()
    v, res = O.solve_batch(sub, mode=3, nthreads=cores)
    dt = time.perf_counter() - t
    r = O.residuals_batch(sub, v).reshape(n, -1)
    conv = int(np.count_nonzero((r * r).sum(1) < 1e-4))
    # single-thread figure on a smaller slice
    n1 = max(1, min(n, 8000))
    sub1 = workloads.shard(sub, 0, max(1, n // n1))
    t = time.perf_counter()
    O.solve_batch(sub1, mode=3, nthreads=1)
    dt1 = time.perf_counter() - t
    return {
        "value": conv / dt,
        "unit": "converged systems/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {n} systems of the same ring16 batch, {cores} threads over disjoint system ranges "
                  f"({dt:.2f} s wall); C++ restatement of the reference algorithm, not the Rust binary",
        "gn_iters_per_sec": float(res["accepted"].sum()) / dt,
        "single_thread_systems_per_sec": (len(sub1["var_off"]) - 1) / dt1,
    }


if __name__ == "__main__":
    sys.exit(main())
