"""Diagnostic: a large sketch kept on the device and solved repeatedly (plans cached after the first solve)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
ctx = fiksi_amd.Context(0)
b = workloads.large_sketch(5000)
db = ctx.upload(b)
for label, o in (("None", abi.solving_opts()), ("L-BFGS", abi.solving_opts(optimizer=1))):
    for rep in range(3):
        t = time.time(); db.system_solve(o); ctx.synchronize(); dt = time.time() - t
        r = db.get_results()[0]
        print(f"cfg2 resident, {label}, solve {rep}: {dt*1e3:.1f} ms, accepted {r['accepted']}, trials {r['trials']}, sse {r['sse']:.6g}")
db.free()
