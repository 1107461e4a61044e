"""The slow end of a ring16 batch: the Systems that run the most trials (the 1.5 % with the most trials), solved as a batch of their own
by the grouped kernel (four Systems per wavefront, one DPP row each) and by the one-System-per-wavefront kernel — the time
of such a batch is the per-trial latency of one solve times its trial count.
    python3 tools/straggler_probe.py [n_systems]
Prints one JSON line."""
import json
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from fiksi_amd import abi, workloads


def take(batch, ids):
    parts = []
    n = len(batch["var_off"]) - 1
    for s in ids:
        parts.append(workloads.shard(batch, int(s), n))
    return workloads.concat(parts)


def timed(ctx, db, reps=5):
    db.system_solve()
    ctx.synchronize()
    best = 1e9
    for _ in range(reps):
        ctx.synchronize()
        ctx.timer_begin()
        db.system_solve()
        best = min(best, ctx.timer_end())
    return best


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
    ctx = abi.Context(0)
    b = workloads.ring16(n)
    v, res = ctx.system_solve_batch(b)
    cut = max(int(np.percentile(res["trials"], 98.5)), 30)
    slow = np.nonzero(res["trials"] >= cut)[0]
    out = {"systems": n, "slow_means_trials_of": cut, "slow_systems": int(len(slow)), "max_trials": int(res["trials"].max()),
           "median_trials": float(np.median(res["trials"])), "trials_p90_p99": [float(np.percentile(res["trials"], q)) for q in (90, 99)]}
    db = ctx.upload(b)
    out["whole_batch_ms"] = timed(ctx, db)
    db.free()
    sb = take(b, slow)
    rest = take(b, np.nonzero(res["trials"] < cut)[0][:4000])
    for name, grouped in (("grouped_kernel", 1), ("one_per_wavefront_kernel", 0)):
        ctx.set_routing(grouped, 1)
        db = ctx.upload(sb)
        ms = timed(ctx, db)
        r = db.get_results()
        db.free()
        out[name] = {"slow_only_ms": ms, "us_per_trial_of_the_longest": ms * 1e3 / float(r["trials"].max())}
        db = ctx.upload(rest)
        out[name]["first_4000_fast_ones_ms"] = timed(ctx, db)
        db.free()
    # where one slow System's cycles go, in either kernel (the diagnostic builds with phase stamps; s_memtime ticks of 100 MHz)
    one = take(b, slow[np.argsort(res["trials"][slow])[-1:]])
    for name, grouped in (("grouped_kernel", 1), ("one_per_wavefront_kernel", 0)):
        ctx.set_routing(grouped, 1)
        db = ctx.upload(one)
        ph = db.phase_cycles()
        db.system_solve()
        r = db.get_results()
        tot = float(sum(ph.values()))
        out[name]["one_slow_system"] = {"trials": int(r["trials"][0]), "accepted": int(r["accepted"][0]),
                                        "phase_share": {k: round(v / tot, 3) for k, v in ph.items()}, "ticks_100MHz": int(tot)}
        db.free()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
