"""fx_system_solve_batch on ONE sketch of n hinged triangles (66 / 258 variables at 16 / 64): latency per call; under rocprofv3
(--hip-trace --kernel-trace --memory-copy-trace) tools/hp_timeline.py shows the last call.  python tools/probes/oneshot_large.py [n_tri]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from fiksi_amd import abi, workloads
from fiksi_amd._lib import check, lib

ctx = abi.Context(0)
n_tri = int(sys.argv[1]) if len(sys.argv) > 1 else 16
a = abi.normalize_batch(workloads.hinged_triangles(1, n_tri))
start = a["vars"].copy()
a["vars"] = start.copy()
res = np.zeros(1, dtype=abi.RESULT_DTYPE)
o = abi.solving_opts()
st = abi.as_struct(a)
ts = []
for k in range(60):
    a["vars"][:] = start
    t0 = time.perf_counter()
    check(lib.fx_system_solve_batch(ctx.handle, C.byref(st), C.byref(o), res.ctypes.data), "fx_system_solve_batch")
    ts.append(time.perf_counter() - t0)
ts = sorted(ts[10:])
print(n_tri, "triangles: median us", round(ts[len(ts) // 2] * 1e6, 1), "min", round(ts[0] * 1e6, 1), "trials", int(res["trials"][0]))
