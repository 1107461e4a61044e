"""cfg4 as an N-GPU run would see it: the 100 000-System ring16 batch cut into N contiguous shards, EVERY shard timed on this
one GPU — the run takes as long as its slowest shard. Prints one JSON line: per N the times of all shards, their maximum,
and the efficiency predicted from it, t(1) / (N max_r t(shard r)).
    python3 tools/shard_times.py [n_systems] [tail:k ...]"""
import json
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from fiksi_amd import abi, workloads


def timed(ctx, db, reps=5):
    db.system_solve()
    ctx.synchronize()
    ts = []
    for _ in range(reps):
        ctx.synchronize()
        ctx.timer_begin()
        db.system_solve()
        ts.append(ctx.timer_end())
    return float(np.median(ts))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100_000
    variants = [a for a in sys.argv[1:] if ":" in a] or ["default"]
    ctx = abi.Context(0)
    full = workloads.ring16(n)
    out = {"systems": n, "variants": {}}
    for v in variants:
        if v == "default":
            ctx.set_ladder()
        elif v == "off:0":
            ctx.set_ladder(False)
        else:
            tail, k = (int(x) for x in v.split(":"))
            ctx.set_ladder(True, tail, k, True)
        row = {}
        t1 = None
        for parts in (1, 2, 4, 8):
            ts = []
            for r in range(parts):
                db = ctx.upload(workloads.shard(full, r, parts) if parts > 1 else full)
                ts.append(round(timed(ctx, db), 4))
                db.free()
            if parts == 1:
                t1 = ts[0]
            row[str(parts)] = {"ms_per_shard": ts, "ms_slowest": max(ts), "predicted_efficiency": round(t1 / (parts * max(ts)), 3)}
        out["variants"][v] = row
    ctx.set_ladder()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
