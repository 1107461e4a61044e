"""Lists the Systems of a random batch whose FX_STEP_QR solve differs from the oracle's in any bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
from oracle import oracle as O
from helpers import random_sketch

dec = int(sys.argv[1]) if len(sys.argv) > 1 else 0
angles = len(sys.argv) > 2 and sys.argv[2] == "angles"
ctx = fiksi_amd.Context(0)
flats = [random_sketch(9000 + s, angles=angles).flatten() for s in range(500)]
b = workloads.concat(flats)
v, res = ctx.system_solve_batch(b, abi.solving_opts(solver=2, decomposer=dec))
if dec:
    v_o, res_o = O.solve_single_pass_batch(b, trial_cap=4096, nthreads=8)
else:
    v_o, res_o = O.solve_batch(b, mode=3, trial_cap=4096, nthreads=8)
nbad = 0
for s in range(len(res)):
    v0, v1 = int(b["var_off"][s]), int(b["var_off"][s + 1])
    same_v = np.array_equal(v[v0:v1].view(np.uint64), v_o[v0:v1].view(np.uint64))
    same_c = all(res[k][s] == res_o[k][s] for k in ("accepted", "trials", "exit", "ncomp"))
    if not (same_v and same_c):
        nbad += 1
        e0, e1 = int(b["expr_off"][s]), int(b["expr_off"][s + 1])
        print(s, "vars same" if same_v else f"vars differ max {np.nanmax(np.abs(v[v0:v1]-v_o[v0:v1])):.3e}", "gpu", res[s], "ref", res_o[s],
              "tags", b["expr_tag"][e0:e1].tolist(), "nvars", v1 - v0, "fixed", int(b["var_fixed"][v0:v1].sum()), flush=True)
print("differing systems:", nbad, "of", len(res))
