"""A / B of the grouped kernel's sparse one-structure build (fx_grouped_s.hip) against the paths such batches took before it
(the team kernels / the wide kernel / one wavefront per System): time per solve of a resident batch, and how the results compare —
both are normal-equation Cholesky steps, in different elimination orders, so counters agree and variables agree to round-off.
    python3 tools/grouped_s_ab.py [reps]          # one child process per build (FIKSI_AMD_GROUPED_C=0 / 1)"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [(11, 100000), (13, 20000), (16, 20000), (24, 8192), (31, 8192)]


def one(reps, dump):
    import numpy as np
    import fiksi_amd
    from fiksi_amd import workloads
    ctx = fiksi_amd.Context(0)
    out = {}
    for n_tri, n in CASES:
        b = workloads.hinged_triangles(n, n_tri)
        db = ctx.upload(b)
        build = db.grouped_build()
        db.system_solve()
        ctx.synchronize()
        ctx.timer_begin()
        for _ in range(reps):
            db.system_solve()
        ms = ctx.timer_end() / reps
        res = db.get_results()
        v = db.get_vars()
        np.save(f"{dump}_{n_tri}_vars.npy", v)
        np.save(f"{dump}_{n_tri}_res.npy", res)
        out[f"hinged_{n_tri}_x_{n}"] = {"variables": 2 + 4 * n_tri, "build": build, "ms": round(ms, 4), "trials": int(res["trials"].sum()),
                                        "accepted": int(res["accepted"].sum()), "converged": float((res["sse_unscaled"] < 1e-4).mean())}
        db.free()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        print(json.dumps(one(int(sys.argv[2]), sys.argv[3])))
        sys.exit(0)
    import numpy as np
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    res = {}
    for flag, name in (("0", "before"), ("1", "sparse_build")):
        env = dict(os.environ, FIKSI_AMD_GROUPED_C=flag)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(reps), "/tmp/gs_" + name], env=env, capture_output=True, text=True)
        if r.returncode:
            print(r.stderr[-3000:])
            sys.exit(1)
        res[name] = json.loads(r.stdout.strip().splitlines()[-1])
    cmp_ = {}
    for n_tri, n in CASES:
        va, vb = np.load(f"/tmp/gs_before_{n_tri}_vars.npy"), np.load(f"/tmp/gs_sparse_build_{n_tri}_vars.npy")
        ra, rb = np.load(f"/tmp/gs_before_{n_tri}_res.npy"), np.load(f"/tmp/gs_sparse_build_{n_tri}_res.npy")
        cmp_[f"hinged_{n_tri}"] = {"same_counters": float(((ra["accepted"] == rb["accepted"]) & (ra["trials"] == rb["trials"]) & (ra["exit"] == rb["exit"])).mean()),
                                   "max_abs_diff_of_variables": float(np.abs(va - vb).max())}
    res["compare"] = cmp_
    print(json.dumps(res))
