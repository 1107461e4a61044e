"""Device-side fuzz: valid structures with hostile VALUES (NaN, +-Inf, 1e308, 0, denormals, coincident points) through every
solve mode — each call must return (exit codes FX_EXIT_NAN / TRIAL_CAP where the reference would loop for ever), fixed variables
must keep their bits, and nothing may hang (run under `timeout`)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
from helpers import random_sketch, mixed_sketch, Lcg

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = fiksi_amd.Context(0)
g = Lcg(777)
bad_vals = [np.nan, np.inf, -np.inf, 1e308, -1e308, 0.0, 5e-324, 1e-300]
base = [workloads.ring16(40), workloads.hinged_triangles(20, 11), workloads.hinged_triangles(6, 16), workloads.hinged_triangles(3, 40),
        workloads.concat([random_sketch(s).flatten() for s in range(40)]), workloads.concat([mixed_sketch(s, fix_some=bool(s & 1)).flatten() for s in range(20)]),
        # round 5: the tiny build (eight lanes per System, with its hand-over of stragglers) and the multifrontal build
        workloads.hinged_triangles(70, 1), workloads.concat([workloads.quadrilateral(bool(s & 1)) for s in range(2)] * 0 + [workloads.quadrilateral(True)] * 67),
        workloads.large_sketch(300, seed=3), workloads.hinged_triangles(2, 64)]
modes = [dict(), dict(decomposer=1), dict(solver=1), dict(solver=2), dict(f32=True), dict(optimizer=1), dict(perturb=False)]
exits = {}
for it in range(n_iter):
    b = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in base[it % len(base)].items()}
    for _ in range(1 + int(g.u(0, 5.99))):
        arr = b["vars"] if g.u(0, 1) < 0.7 else b["expr_param"]
        i = int(g.u(0, len(arr) - 1e-9))
        if g.u(0, 1) < 0.3 and arr is b["vars"] and i + 2 < len(arr):
            arr[i + 2] = arr[i]  # coincident coordinates
        else:
            arr[i] = bad_vals[int(g.u(0, len(bad_vals) - 1e-9))]
    kw = modes[it % len(modes)]
    v, r = ctx.system_solve_batch(b, abi.solving_opts(**kw))
    fx = b["var_fixed"] == 1
    assert np.array_equal(v[fx].view(np.uint64), b["vars"][fx].view(np.uint64)), it
    for e in r["exit"]:
        exits[int(e)] = exits.get(int(e), 0) + 1
    if it % 50 == 0:
        print("iteration", it, exits, flush=True)
print("done", exits)
