"""One-shot solves of large sketches, repeated (System::solve on a sketch that is being dragged): the context keeps the
sparse path's plan of a structure it has seen, so only the first call pays for host planning."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
ctx = fiksi_amd.Context(0)
for name, b in (("cfg2 (5000 points)", workloads.large_sketch(5000)), ("hinged(64), 258 variables", workloads.hinged_triangles(1, 64)),
                ("16 x hinged(64)", workloads.hinged_triangles(16, 64))):
    for dec in (0, 1):
        ts, ref = [], None
        for rep in range(5):
            t0 = time.perf_counter()
            v, r = ctx.system_solve_batch(b, abi.solving_opts(decomposer=dec))
            ts.append((time.perf_counter() - t0) * 1e3)
            if ref is None: ref = (v.copy(), r.copy())
            assert np.array_equal(v.view(np.uint64), ref[0].view(np.uint64)) and r.tobytes() == ref[1].tobytes()
        print(f"{name}, decomposer {dec}: " + " ".join(f"{t:.2f}" for t in ts) + " ms", flush=True)
