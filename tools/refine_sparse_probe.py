"""The refined step on the sparse path: agreement with the oracle on large ill-conditioned sketches, and its cost on cfg2."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
from oracle import oracle as O
from helpers import random_big_sketch
ctx = fiksi_amd.Context(0)
b = workloads.concat([random_big_sketch(1000 * 90 + s, 90).flatten() for s in range(24)])
v_o, r_o = O.solve_batch(b, mode=3, trial_cap=4096, nthreads=8)
for solver in (0, 1):
    v, r = ctx.system_solve_batch(b, abi.solving_opts(solver=solver))
    ok = np.isfinite(r["sse"]) & np.isfinite(r_o["sse"])
    same = (r["accepted"] == r_o["accepted"]) & (r["trials"] == r_o["trials"])
    d = np.abs(r["sse"] - r_o["sse"]) / np.maximum(np.abs(r_o["sse"]), 1e-12)
    print(f"solver {solver}: same path {same[ok].mean():.3f}  rel SSE diff median {np.median(d[ok]):.2e} p90 {np.quantile(d[ok], 0.9):.2e} max {d[ok].max():.2e}", flush=True)
c2 = workloads.large_sketch(5000)
for solver in (0, 1):
    ctx.system_solve_batch(c2, abi.solving_opts(solver=solver))
    t0 = time.time(); v, r = ctx.system_solve_batch(c2, abi.solving_opts(solver=solver)); t1 = time.time()
    print(f"cfg2 solver {solver}: {t1 - t0:.3f} s  accepted {r['accepted'][0]} trials {r['trials'][0]} exit {r['exit'][0]} sse {r['sse'][0]:.12e}", flush=True)
