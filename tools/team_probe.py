"""Diagnostic: the workgroup ("team") kernels of the sparse path (fx_sparse_team.h) — the reference's 64-triangle bench
sketch (258 variables) as a resident batch and as single solves, chains of mid-size sketches against the oracle, cfg2."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import fiksi_amd
from fiksi_amd import abi, workloads
from oracle import oracle as O

ctx = fiksi_amd.Context(0)
what = sys.argv[1:] or ["hinged", "mid", "cfg2"]


def timed(db, opts, reps=3):
    db.system_solve(opts)
    ctx.synchronize()
    ctx.timer_begin()
    for _ in range(reps):
        db.system_solve(opts)
    return ctx.timer_end() / reps


if "hinged" in what:
    for n_tri, n_batch in ((64, 256), (64, 1), (32, 256), (128, 64)):
        b = workloads.hinged_triangles(n_batch, n_tri)
        db = ctx.upload(b)
        for solver in (0, 1):
            ms = timed(db, abi.solving_opts(solver=solver))
            res = db.get_results()
            conv = int(np.count_nonzero(res["sse_unscaled"] < 1e-4))
            print(f"hinged_triangles({n_tri}) x {n_batch} solver {solver}: {ms:.3f} ms per solve of the batch, {conv} converged, "
                  f"accepted {res['accepted'][:4].tolist()} trials {res['trials'][:4].tolist()} -> {conv / (ms * 1e-3):.0f} Systems/s", flush=True)
        v = db.get_vars()
        db.free()
        one = workloads.hinged_triangles(1, n_tri)
        v_o, res_o = O.solve_batch(one, mode=3)
        nv = len(v_o)
        print(f"   oracle: accepted {int(res_o['accepted'][0])} trials {int(res_o['trials'][0])} sse {float(res_o['sse'][0]):.3e}; "
              f"device sse {float(res['sse'][0]):.3e}; max |dv| {np.max(np.abs(v[:nv] - v_o)):.2e}", flush=True)

if "mid" in what:
    for n_pts in (40, 100, 300, 700):
        batches = [workloads.large_sketch(n_pts, seed=7 + k) for k in range(4)]
        b = workloads.concat(batches)
        t0 = time.time()
        v, res = ctx.system_solve_batch(b)
        dt = time.time() - t0
        t0 = time.time()
        v, res = ctx.system_solve_batch(b)
        dt2 = time.time() - t0
        v_o, res_o = O.solve_batch(b, mode=3)
        print(f"4 chain sketches of {n_pts} points: first call {dt * 1e3:.2f} ms, second {dt2 * 1e3:.2f} ms; accepted {res['accepted'].tolist()} vs "
              f"{res_o['accepted'].tolist()}; trials {res['trials'].tolist()} vs {res_o['trials'].tolist()}; "
              f"sse rel diff {np.max(np.abs(res['sse'] - res_o['sse']) / np.maximum(res_o['sse'], 1e-30)):.2e}", flush=True)

if "cfg2" in what:
    b = workloads.large_sketch(5000)
    for rep in range(3):
        t0 = time.time()
        v, res = ctx.system_solve_batch(b)
        dt = time.time() - t0
        print(f"cfg2 one-shot call {rep}: {dt * 1e3:.2f} ms, accepted {int(res['accepted'][0])} trials {int(res['trials'][0])} "
              f"exit {int(res['exit'][0])} sse {float(res['sse'][0]):.6e}", flush=True)
    db = ctx.upload(b)
    for solver in (0, 1):
        ms = timed(db, abi.solving_opts(solver=solver), reps=5)
        res = db.get_results()
        print(f"cfg2 resident solver {solver}: {ms:.3f} ms per solve (device timer), accepted {int(res['accepted'][0])} trials {int(res['trials'][0])}", flush=True)
    db.free()
    # a group of eight cfg2-size sketches of one structure: one workgroup each
    b8 = workloads.concat([workloads.large_sketch(2000, seed=7) for _ in range(8)])
    db = ctx.upload(b8)
    ms = timed(db, abi.solving_opts(), reps=2)
    res = db.get_results()
    print(f"8 x 2000-point sketches of one structure: {ms:.3f} ms per solve, trials {res['trials'].tolist()}", flush=True)
    db.free()
ctx.close()
