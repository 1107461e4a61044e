"""Decomposer::RecursiveAssembly over a batch of Systems of one structure (the reference's `single_triangle` sketch,
fiksi/src/tests/triangles.rs:10-37, N times with jittered start points): wall time of one fxs_systems_solve call.
    python3 tools/ra_batch.py [n_systems] [reps]
Prints one JSON line."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ctypes as C

import fiksi_amd as F
from fiksi_amd import abi
from fiksi_amd._lib import check, lib
from fiksi_amd.system import solve_systems


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    ctx = F.abi.Context(0)
    rng = np.random.default_rng(3)
    start = np.array([(0., 0.), (1., .5), (2., 1.)])
    systems, handles, jit = [], [], rng.uniform(-0.05, 0.05, size=(n, 3, 2))
    t0 = time.perf_counter()
    for k in range(n):
        s = F.System()
        pts = [F.elements.Point.create(s, *(start[i] + jit[k, i])) for i in range(3)]
        for i, j in ((0, 1), (0, 2), (1, 2)):
            F.constraints.PointPointDistance.create(s, pts[i], pts[j], 1.)
        systems.append(s)
        handles.append(pts)
    build_s = time.perf_counter() - t0
    out = {"workload": f"{n} x single_triangle (triangles.rs:10-37), jittered starts", "build_s": build_s}
    for dec in (F.Decomposer.RecursiveAssembly, F.Decomposer.NONE):
        opts = F.SolvingOptions(decomposer=dec)
        res = solve_systems(systems, opts, ctx=ctx)
        times, wrapped = [], []
        arr = (C.c_void_p * n)(*[s._h for s in systems])   # what fiksi_amd.system.solve_systems builds per call
        o = opts._to_abi()
        for rep in range(2 * reps):
            for k in range(n):
                for i in range(3):
                    handles[k][i].update_value(systems[k], *(start[i] + jit[k, i]))
            ctx.synchronize()
            t0 = time.perf_counter()
            if rep % 2:
                res = solve_systems(systems, opts, ctx=ctx)
                wrapped.append((time.perf_counter() - t0) * 1e3)
            else:
                res = np.zeros(n, dtype=abi.RESULT_DTYPE)
                check(lib.fxs_systems_solve(arr, n, ctx.handle, C.byref(o), res.ctypes.data), "fxs_systems_solve")
                times.append((time.perf_counter() - t0) * 1e3)
        out[dec.name] = {"solve_ms_min": min(times), "solve_ms_median": sorted(times)[len(times) // 2],
                         "through_python_wrapper_ms_median": sorted(wrapped)[len(wrapped) // 2],
                         "max_sse_unscaled": float(res["sse_unscaled"].max()),
                         "device_solves_per_system": int(res["ncomp"][0]), "trials_mean": float(res["trials"].mean())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
