import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from fiksi_amd import abi, workloads
ctx = abi.Context(0)
b = workloads.ring16(100000)
db = ctx.upload(b)
o = abi.solving_opts(solver=2)
for route in (-1,):
    ctx.set_routing(route)
    db.system_solve(o); ctx.synchronize()
    ts = []
    for _ in range(5):
        ctx.timer_begin(); db.system_solve(o); ts.append(ctx.timer_end())
    print("qr routing", route, "route", db.solve_route(o), "ms", round(float(np.median(ts)), 3))
