"""Outcome-level sweep of a step solver against the oracle on many random small sketches (None and SinglePass):
fraction of Systems with the oracle's accepted / trial counts, with the same verdict (sum r^2 < 1e-4 on the solved
variables, fiksi_bench.rs:65-72), and the distribution of |SSE - SSE_oracle| on the Systems that share the path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
from oracle import oracle
from helpers import random_sketch, mixed_sketch

solver, seed0, count, chunk = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 1000
ctx = fiksi_amd.Context(0)
for dec in (0, 1):
    n = same = verdict = 0
    rel = []
    for lo in range(seed0, seed0 + count, chunk):
        flats = [random_sketch(s).flatten() for s in range(lo, lo + chunk)] + [mixed_sketch(s, fix_some=bool(s & 1)).flatten() for s in range(lo, lo + chunk // 10)]
        b = workloads.concat(flats)
        v, r = ctx.system_solve_batch(b, abi.solving_opts(solver=solver, decomposer=dec))
        v_o, r_o = (oracle.solve_batch(b, mode=3, trial_cap=4096, nthreads=16) if dec == 0 else oracle.solve_single_pass_batch(b, trial_cap=4096, nthreads=16))
        res, res_o = oracle.residuals_batch(b, v), oracle.residuals_batch(b, v_o)
        for s in range(len(flats)):
            e0, e1 = int(b["expr_off"][s]), int(b["expr_off"][s + 1])
            if int(r_o["trials"][s]) >= 4000 or r["exit"][s] >= 4:
                continue  # non-finite steps: stopped differently on the two sides
            n += 1
            sp = r["accepted"][s] == r_o["accepted"][s] and r["trials"][s] == r_o["trials"][s]
            same += sp
            sq, sq_o = float((res[e0:e1] ** 2).sum()), float((res_o[e0:e1] ** 2).sum())
            verdict += (sq < 1e-4) == (sq_o < 1e-4)
            if sp and np.isfinite(r_o["sse"][s]):
                rel.append(abs(r["sse"][s] - r_o["sse"][s]) / (1e-10 + 1e-6 * abs(r_o["sse"][s])))
    rel = np.array(rel)
    print(f"solver {solver} decomposer {dec}: {n} Systems, same path {same / n:.4f}, same verdict {verdict / n:.4f}, "
          f"|dSSE| / (1e-10 + 1e-6 SSE) on the same path: median {np.median(rel):.3g}, p99 {np.percentile(rel, 99):.3g}, max {rel.max():.3g}", flush=True)
