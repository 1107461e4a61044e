"""How the paths for Systems beyond one wavefront scale with the number of Systems in a resident batch (a hidden per-System
host cost shows as a straight line through the origin): the reference's 64-triangle sketch under LM, L-BFGS and SinglePass,
and batches of many different structures.
    python3 tools/large_scaling.py
Prints one JSON line."""
import json
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from fiksi_amd import abi, workloads


def timed(ctx, db, opts, reps=3):
    db.system_solve(opts)
    ctx.synchronize()
    ctx.timer_begin()
    for _ in range(reps):
        db.system_solve(opts)
    return ctx.timer_end() / reps


def main():
    ctx = abi.Context(0)
    out = {}
    for name, opts in (("lm", abi.solving_opts()), ("lbfgs", abi.solving_opts(optimizer=1)), ("single_pass", abi.solving_opts(decomposer=1)),
                       ("lm_refined", abi.solving_opts(solver=1))):
        out[name] = {}
        for n in (1, 16, 64, 256):
            db = ctx.upload(workloads.hinged_triangles(n, 64))
            out[name][str(n)] = round(timed(ctx, db, opts), 4)
            db.free()
    out["different_structures_lm"] = {}
    for n in (1, 8, 32):
        b = workloads.concat([workloads.large_sketch(70 + 3 * k, seed=100 + k) for k in range(n)])
        db = ctx.upload(b)
        out["different_structures_lm"][str(n)] = round(timed(ctx, db, abi.solving_opts()), 4)
        db.free()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
