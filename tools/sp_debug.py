import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
from oracle import oracle as O
from helpers import random_big_sketch
n_points = 80
flats = [random_big_sketch(500 * n_points + s, n_points).flatten() for s in range(10)]
b = workloads.concat(flats)
ctx = fiksi_amd.Context(0)
v, res = ctx.system_solve_batch(b, abi.solving_opts(decomposer=1))
v_o, res_o = O.solve_single_pass_batch(b, trial_cap=4096, nthreads=8)
r = O.residuals_batch(b, v); r_o = O.residuals_batch(b, v_o)
for s in range(len(res)):
    e0, e1 = int(b["expr_off"][s]), int(b["expr_off"][s + 1]); v0, v1 = int(b["var_off"][s]), int(b["var_off"][s + 1])
    dr = np.abs(np.abs(r[e0:e1]) - np.abs(r_o[e0:e1]))
    print(s, "gpu", res[s], "ref", res_o[s], "max|dr| %.3e at %d  max|r_o| %.3e  max|dv| %.3e" % (dr.max(), dr.argmax(), np.abs(r_o[e0:e1]).max(), np.abs(v[v0:v1]-v_o[v0:v1]).max()))
s = 1
e0, e1 = int(b["expr_off"][s]), int(b["expr_off"][s + 1])
blocks = abi.single_pass_blocks(b, s)
print("blocks of system 1:", len(blocks), "sizes", [(len(r_), len(v_)) for _, r_, v_ in blocks][:60])
dr = np.abs(np.abs(r[e0:e1]) - np.abs(r_o[e0:e1]))
for i in np.argsort(-dr)[:8]:
    print(" expr", i, "tag", b["expr_tag"][e0 + i], "r gpu %.6e ref %.6e" % (r[e0 + i], r_o[e0 + i]))
