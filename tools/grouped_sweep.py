"""Sweep: the grouped kernel against the one-System-per-wavefront kernel on uniform batches of many shapes — same bits
expected for every variable and result (f64 and f32, None and SinglePass, with and without the scheduling switches; where the
grouped kernel's sparse one-structure build takes the batch: the same counters, variables to 1e-9)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
ctx = fiksi_amd.Context(0)
bad = 0
shapes = [("hinged%d" % t, lambda t=t: workloads.hinged_triangles(3000, t)) for t in range(1, 12)]
shapes += [("ring16", lambda: workloads.ring16(3000)), ("ring16_gauge", lambda: workloads.ring16(3000, fix_gauge=True)),
           ("ring16_inconsistent", lambda: workloads.ring16(3000, inconsistent=True))]
for name, make in shapes:
    b = make()
    for kw in ({}, {"f32": True}, {"decomposer": 1}, {"perturb": False}):
        out = []
        for route, hold, presort in ((0, 2, True), (1, 2, True), (1, 0, False), (1, 5, True)):
            ctx.set_routing(route); ctx.set_hold_passes(hold); ctx.set_presort(presort, 1024)
            out.append(ctx.system_solve_batch(b, abi.solving_opts(**kw)))
        ctx.set_routing(1)
        db = ctx.upload(b)
        sparse = db.grouped_build(abi.solving_opts(**kw)) == 2  # (fx_grouped_s.hip: another elimination order — counters equal, variables to round-off)
        db.free()
        for k in range(1, len(out)):
            if sparse:
                same = all(np.array_equal(out[0][1][f], out[k][1][f]) for f in ("accepted", "trials", "exit")) and float(np.abs(out[0][0] - out[k][0]).max()) < 1e-9
            else:
                same = np.array_equal(out[0][0].view(np.uint64), out[k][0].view(np.uint64)) and out[0][1].tobytes() == out[k][1].tobytes()
            if not same:
                bad += 1
                print("DIFFERENT", name, kw, "variant", k, flush=True)
    print(name, "done", flush=True)
ctx.set_routing(-1); ctx.set_hold_passes(2); ctx.set_presort(True, 8192)
print("TOTAL different", bad)
