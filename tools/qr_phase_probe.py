import sys, json
sys.path.insert(0, '/root/repo')
import numpy as np
from fiksi_amd import abi, workloads
ctx = abi.Context(0)
ctx.set_routing(0, 1)
b = workloads.ring16(4096)
db = ctx.upload(b)
for solver in (0, 2):
    o = abi.solving_opts(solver=solver)
    db.system_solve(o); ctx.synchronize()
    ctx.timer_begin(); db.system_solve(o); ms = ctx.timer_end()
    r = db.get_results()
    ph = db.phase_cycles(o)
    tot = sum(ph.values())
    print(json.dumps({"solver": solver, "ms": ms, "trials": int(r["trials"].sum()), "phase_share": {k: round(v / tot, 3) for k, v in ph.items()}, "ticks_per_trial": tot / float(r["trials"].sum())}))
