"""The relay of the one-structure build's stragglers (fx_ctx_set_relay: sixteen lambdas per round in a workgroup of four wavefronts once
the queue is empty) against the wavefront's own ladder: the headline batch and every 1/8 shard of it (cfg4 as an 8-GPU run would see
it), time per resident solve and every bit of variables and results compared.  python tools/relay_ab.py [n_systems] [min_trials ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import fiksi_amd
from fiksi_amd import abi, workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
mins = [int(x) for x in sys.argv[2:]] or [16]
ctx = fiksi_amd.Context(0)
opts = abi.solving_opts()
full = workloads.ring16(n)


def timed(db, reps=7):
    db.system_solve(opts)
    ctx.synchronize()
    ts = []
    for _ in range(reps):
        ctx.timer_begin()
        db.system_solve(opts)
        ts.append(ctx.timer_end())
    return sorted(ts)[len(ts) // 2]


out = {"systems": n, "batches": {}}
for name, b in [("whole", full)] + [(f"shard_{r}_of_8", workloads.shard(full, r, 8)) for r in range(8)]:
    row = {}
    db = ctx.upload(b)
    ctx.set_relay(False)
    row["off_ms"] = round(timed(db), 4)
    v0, r0 = db.get_vars().copy(), db.get_results().copy()
    for m in mins:
        ctx.set_relay(True, m)
        row[f"relay_{m}_ms"] = round(timed(db), 4)
        v1, r1 = db.get_vars().copy(), db.get_results().copy()
        row[f"relay_{m}_same_bits"] = bool(np.array_equal(v0.view(np.uint64), v1.view(np.uint64)) and r0.tobytes() == r1.tobytes())
        if not row[f"relay_{m}_same_bits"]:
            bad = np.nonzero((r0["trials"] != r1["trials"]) | (r0["sse"] != r1["sse"]) | (r0["accepted"] != r1["accepted"]) | (r0["exit"] != r1["exit"]))[0]
            row[f"relay_{m}_differing"] = int(len(bad))
            if len(bad):
                row[f"relay_{m}_first"] = [str(r0[bad[0]]), str(r1[bad[0]])]
    row["max_trials"] = int(r0["trials"].max())
    db.free()
    out["batches"][name] = row
    print(name, row, flush=True)
ctx.set_relay(True)
shards = [out["batches"][f"shard_{r}_of_8"] for r in range(8)]
for key in ["off_ms"] + [f"relay_{m}_ms" for m in mins]:
    slow = max(s[key] for s in shards)
    out[f"slowest_shard_{key}"] = slow
    out[f"predicted_efficiency_8_{key}"] = round(out["batches"]["whole"][key] / 8.0 / slow, 3)
print(json.dumps(out))
