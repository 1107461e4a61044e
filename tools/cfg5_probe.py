import sys; sys.path.insert(0,'.')
import numpy as np, fiksi_amd
from fiksi_amd import abi, workloads
ctx = fiksi_amd.Context(0)
b = workloads.ring16(125000, inconsistent=True)
for f32 in (True, False):
    v, res = ctx.system_solve_batch(b, abi.solving_opts(f32=f32))
    print("f32" if f32 else "f64", np.bincount(res["exit"], minlength=6), res["accepted"].max(), res["trials"].max())
    bad = np.where(res["exit"] > 2)[0][:5]; print(res[bad])
