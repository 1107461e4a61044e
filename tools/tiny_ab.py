"""The tiny one-structure build (fx_grouped_tiny.hip, eight lanes per System) against the 16-column build (fx_grouped_c.hip) on
batches of small sketches: time per resident solve, and every bit of the variables and results compared.
    python tools/tiny_ab.py [n_systems]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import fiksi_amd
from fiksi_amd import abi, workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = fiksi_amd.Context(0)
out = {}
cases = [("hinged_triangles_1", workloads.hinged_triangles(n, 1))]
try:
    from helpers import tiny_sketch_batches
    cases += tiny_sketch_batches(n // 10)
except ImportError:
    pass
for name, b in cases:
    row = {}
    keep = {}
    for tiny in (False, True):
        ctx.set_one_structure_builds(True, tiny=tiny)
        db = ctx.upload(b)
        row["build_tiny" if tiny else "build_c1"] = db.grouped_build(abi.solving_opts()) if hasattr(db, "grouped_build") else None
        for opts_name, opts in (("", abi.solving_opts()), ("_noperturb", abi.solving_opts(perturb=False))):
            db.system_solve(opts)
            ctx.synchronize()
            ctx.timer_begin()
            for _ in range(5):
                db.system_solve(opts)
            ms = ctx.timer_end() / 5
            keep[(tiny, opts_name)] = (db.get_vars().copy(), db.get_results().copy())
            row[("tiny" if tiny else "c1") + opts_name + "_ms"] = round(ms, 4)
        db.free()
    for opts_name in ("", "_noperturb"):
        (v0, r0), (v1, r1) = keep[(False, opts_name)], keep[(True, opts_name)]
        row["same_bits" + opts_name] = bool(np.array_equal(v0.view(np.uint64), v1.view(np.uint64)) and r0.tobytes() == r1.tobytes())
        if not row["same_bits" + opts_name]:
            bad = np.nonzero((r0["trials"] != r1["trials"]) | (r0["sse"] != r1["sse"]) | (r0["accepted"] != r1["accepted"]))[0]
            row["differing_results" + opts_name] = int(len(bad))
            row["first" + opts_name] = [str(r0[bad[0]]), str(r1[bad[0]])] if len(bad) else "variables only"
    row["trials_mean"] = float(keep[(True, "")][1]["trials"].mean())
    row["converged"] = int((keep[(True, "")][1]["sse_unscaled"] < 1e-4).sum())
    out[name] = row
    print(name, row, flush=True)
ctx.set_one_structure_builds(True)
print(json.dumps(out))
