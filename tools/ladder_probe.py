"""The lambda ladder of the grouped kernel (fx_ctx_set_ladder) on the headline batch and on its 1/2, 1/4 and 1/8 shards (cfg4):
time per solve with the ladder off / on (rows help once the queue is empty) / on with a wavefront that holds a straggler
leaving the queue early, the first round of tickets dealt one per wavefront or not — and that every variant returns the
bits of the ladder-free solve.
    python3 tools/ladder_probe.py [n_systems] [quick|full] [tail:k,tail:k,...]
Prints one JSON line."""
import json
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from fiksi_amd import abi, workloads


def timed(ctx, db, reps=7):
    db.system_solve()
    ctx.synchronize()
    ts = []
    for _ in range(reps):
        ctx.synchronize()
        ctx.timer_begin()
        db.system_solve()
        ts.append(ctx.timer_end())
    return float(np.median(ts)), float(min(ts))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    quick = len(sys.argv) > 2 and sys.argv[2] == "quick"
    ctx = abi.Context(0)
    full = workloads.ring16(n)
    variants = [("off", dict(enable=False)),
                ("tail_only_no_spread", dict(enable=True, tail_systems=0, spread=False)),
                ("tail_only", dict(enable=True, tail_systems=0, spread=True))]
    if len(sys.argv) > 3:  # "tail:k,tail:k,..."
        for tk in sys.argv[3].split(","):
            tail, k, sp = (int(x) for x in (tk + ":1").split(":")[:3])
            variants.append(("park_tail%d_k%d_s%d" % (tail, k, sp), dict(enable=True, tail_systems=tail, min_trials=k, spread=sp)))
    elif not quick:
        for tail in (2048, 4096, 8192, 16384):
            for k in (8, 16, 24):
                variants.append(("park_tail%d_k%d" % (tail, k), dict(enable=True, tail_systems=tail, min_trials=k, spread=True)))
    out = {"systems": n, "shards": {}}
    for parts in (1, 2, 4, 8):
        b = workloads.shard(full, 0, parts) if parts > 1 else full
        db = ctx.upload(b)
        ref = None
        row = {}
        for name, kw in variants:
            ctx.set_ladder(**kw)
            med, best = timed(ctx, db)
            v, r = db.get_vars().copy(), db.get_results().copy()
            if ref is None:
                ref = (v, r)
                same = True
            else:
                same = bool(np.array_equal(v.view(np.uint64), ref[0].view(np.uint64)) and r.tobytes() == ref[1].tobytes())
            row[name] = {"ms_median": round(med, 4), "ms_min": round(best, 4), "same_bits_as_off": same}
        out["shards"][str(parts)] = {"systems": len(b["var_off"]) - 1, "max_trials": int(ref[1]["trials"].max()), "variants": row}
        db.free()
    ctx.set_ladder()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
