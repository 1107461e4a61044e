"""A / B of the multifrontal build (fx_front.h) against the column walkers (fx_sparse_team.h) on Systems beyond one wavefront:
resident solves, plan warm; counters and variables compared. python tools/fronts_ab.py [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import fiksi_amd
from fiksi_amd import abi, workloads

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = fiksi_amd.Context(0)
ctx.set_one_structure_builds(False)  # (the grouped kernel's sparse build would take the batches of small factors)
opts = abi.solving_opts()
out = {}
cases = [("cfg2 (5 000 points, one System)", workloads.large_sketch(5000)),
         ("hinged_64 x 1", workloads.hinged_triangles(1, 64)),
         ("hinged_64 x 256", workloads.hinged_triangles(256, 64)),
         ("hinged_16 x 1", workloads.hinged_triangles(1, 16)),
         ("large_sketch(300) x 1", workloads.large_sketch(300, seed=3)),
         ("large_sketch(300) x 64", workloads.concat([workloads.large_sketch(300, seed=3 + k) for k in range(64)])),
         ("large_sketch(1500) x 1", workloads.large_sketch(1500, seed=5))]
for name, b in cases:
    row = {}
    keep = {}
    for fronts in (False, True):
        ctx.set_sparse_fronts(fronts)
        db = ctx.upload(b)
        db.system_solve(opts)
        ctx.synchronize()
        ctx.timer_begin()
        for _ in range(reps):
            db.system_solve(opts)
        ms = ctx.timer_end() / reps
        v, r = db.get_vars().copy(), db.get_results().copy()
        db.free()
        keep[fronts] = (v, r)
        row["fronts_ms" if fronts else "walkers_ms"] = round(ms, 4)
    (v0, r0), (v1, r1) = keep[False], keep[True]
    row["same_counters"] = bool(np.array_equal(r0["accepted"], r1["accepted"]) and np.array_equal(r0["trials"], r1["trials"]) and
                                np.array_equal(r0["exit"], r1["exit"]))
    row["accepted_trials"] = [int(r1["accepted"][0]), int(r1["trials"][0])]
    row["walkers_accepted_trials"] = [int(r0["accepted"][0]), int(r0["trials"][0])]
    row["max_dvar_over_scale"] = float(np.max(np.abs(v0 - v1)) / max(1.0, float(r0["scale"].max())))
    row["max_rel_dsse"] = float(np.max(np.abs(r0["sse"] - r1["sse"]) / (1e-300 + np.abs(r0["sse"]))))
    out[name] = row
    print(name, row, flush=True)
ctx.set_sparse_fronts(True)
print(json.dumps(out))
