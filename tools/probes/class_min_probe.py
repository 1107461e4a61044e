"""Batches of a few structures with fewer than 2 048 Systems each: the one-structure build per class (FIKSI_AMD_CLASS_MIN lowered) against
the general build. python tools/probes/class_min_probe.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
ctx = fiksi_amd.Context(0)
out = {}
for per in (256, 512, 1000, 1500):
    parts = [workloads.ring16(per), workloads.ring16(per, fix_gauge=True), workloads.ring_chords(per, 16, 5), workloads.hinged_triangles(per, 7),
             workloads.ring16(per, inconsistent=True, seed0=5000), workloads.hinged_triangles(per, 5)]
    # interleave the structures System by System
    b = workloads.concat([workloads.shard(p, k, per) for k in range(per) for p in parts])
    db = ctx.upload(b)
    build = db.grouped_build()
    db.system_solve(); ctx.synchronize()
    ts = []
    for _ in range(7):
        ctx.timer_begin(); db.system_solve(); ts.append(ctx.timer_end())
    r = db.get_results()
    out[per] = {"systems": 6 * per, "build": build, "ms": round(sorted(ts)[3], 4), "trials": int(r["trials"].sum())}
    db.free()
print(json.dumps(out))
