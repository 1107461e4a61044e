"""Most-work-first hand-out from the scout pass (fx_ctx_set_presort) against index order and against the history schedule."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
ctx = fiksi_amd.Context(0)
batches = {"ring16 100k f64": (workloads.ring16(100000), False), "ring16 50k f64": (workloads.ring16(50000), False),
           "ring16 25k f64": (workloads.ring16(25000), False), "cfg5 125k f32": (workloads.ring16(125000, inconsistent=True), True),
           "cfg5 125k f64": (workloads.ring16(125000, inconsistent=True), False), "hinged11 100k f64": (workloads.hinged_triangles(100000, 11), False),
           "ring16 500k f64": (workloads.ring16(500000), False)}
for name, (b, f32) in batches.items():
    db = ctx.upload(b)
    o = abi.solving_opts(f32=f32)
    line = f"{name:20s}"
    ref = None
    for mode in ("index", "presort", "history"):
        ctx.set_presort(mode == "presort")
        db.schedule_by_last_solve(False)
        if mode == "history":
            db.system_solve(o); db.schedule_by_last_solve(True)
        db.system_solve(o); ctx.synchronize()
        ctx.timer_begin()
        for _ in range(5): db.system_solve(o)
        ms = ctx.timer_end() / 5
        v, r = db.get_vars(), db.get_results()
        if ref is None: ref = (v.copy(), r.copy())
        same = np.array_equal(v.view(np.uint64), ref[0].view(np.uint64)) and r.tobytes() == ref[1].tobytes()
        line += f"  {mode} {ms:7.3f} ms{'' if same else ' DIFFERENT!'}"
    print(line, flush=True)
    db.free()
