"""Diagnostic: how much closer FX_STEP_CHOLESKY_REFINED follows the oracle's QR-based LM on hard sketches."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
from oracle import oracle as O
from helpers import mixed_sketch, random_sketch
ctx = fiksi_amd.Context(0)
sets = {
    "mixed": workloads.concat([mixed_sketch(100 + s, fix_some=s % 3 == 0).flatten() for s in range(200)]),
    "random": workloads.concat([random_sketch(s).flatten() for s in range(400)]),
    "ring16": workloads.ring16(4000),
    "hinged20 (wide)": workloads.hinged_triangles(200, 20),
}
for name, b in sets.items():
    v_o, r_o = O.solve_batch(b, mode=3, trial_cap=4096, nthreads=8)
    for solver in (0, 1):
        v, r = ctx.system_solve_batch(b, abi.solving_opts(solver=solver))
        ok = ~(np.isnan(r_o["sse"]) | np.isnan(r["sse"]))
        same = (r["accepted"] == r_o["accepted"]) & (r["trials"] == r_o["trials"]) & ok
        d = np.abs(r["sse"] - r_o["sse"]) / np.maximum(np.abs(r_o["sse"]), 1e-12)
        print(f"{name:16s} solver={solver}: same step counts {same.mean():.3f}; SSE rel diff median {np.median(d[ok]):.2e} "
              f"p90 {np.quantile(d[ok], 0.9):.2e} (same-count only p99 {np.quantile(d[same], 0.99):.2e})")
