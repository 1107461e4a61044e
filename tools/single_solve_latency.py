import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from fiksi_amd import abi, workloads
import ctypes as C
from fiksi_amd._lib import check, lib
ctx = abi.Context(0)
for n_tri in (1, 4):
    b = workloads.hinged_triangles(1, n_tri)
    a = abi.normalize_batch(b)
    start = a["vars"].copy(); a["vars"] = start.copy()
    res = np.zeros(1, dtype=abi.RESULT_DTYPE); o = abi.solving_opts(); st = abi.as_struct(a)
    ts = []
    for k in range(200):
        a["vars"][:] = start
        t0 = time.perf_counter()
        check(lib.fx_system_solve_batch(ctx.handle, C.byref(st), C.byref(o), res.ctypes.data), "x")
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[20:])
    print(n_tri, "median us", ts[len(ts)//2]*1e6, "min", ts[0]*1e6, "trials", int(res["trials"][0]))
