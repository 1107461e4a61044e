"""A / B of the grouped kernel's one-structure build (fx_grouped_c.hip) against the general build on the same resident
batch — batches of one structure, and batches of a few structures (a launch over their structure classes): time per solve and a
digest of every result record and solved variable (the two must agree bit for bit).
    python3 tools/grouped_c_ab.py [n_systems] [reps]          # both builds, one child process each (FIKSI_AMD_GROUPED_C=0 / 1)"""
import hashlib
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(n, reps):
    import numpy as np
    import fiksi_amd
    from fiksi_amd import abi, workloads
    ctx = fiksi_amd.Context(0)
    out = {}
    for name, b in (("ring16", workloads.ring16(n)), ("ring16_fixed_gauge", workloads.ring16(n // 4, fix_gauge=True)),
                    ("ring16_inconsistent", workloads.ring16(n // 4, inconsistent=True)), ("ring20_chords", workloads.ring_chords(n, 20, 7)),
                    ("hinged_5", workloads.hinged_triangles(n, 5)), ("hinged_1", workloads.hinged_triangles(n, 1)),
                    ("hinged_3", workloads.hinged_triangles(n, 3)), ("ring16_inconsistent_f32", workloads.ring16(n + n // 4, inconsistent=True)),
                    ("ring16_overconstrained", workloads.ring16_overconstrained(n)), ("ring16_overconstrained_f32", workloads.ring16_overconstrained(n)),
                    # batches of SEVERAL structures: one launch over their big structure classes (+ the general build for the rest)
                    ("two_structures", workloads.ring16_two_structures(n)),
                    ("three_classes_and_a_small_one", workloads.concat([workloads.ring16(3 * n // 10), workloads.hinged_triangles(3 * n // 10, 5),
                                                                         workloads.ring16(4 * n // 10, fix_gauge=True), workloads.hinged_triangles(700, 3)]))):
        db = ctx.upload(b)
        o = abi.solving_opts(f32=name.endswith("_f32"))
        db.system_solve(o)
        ctx.synchronize()
        ctx.timer_begin()
        for _ in range(reps):
            db.system_solve(o)
        ms = ctx.timer_end() / reps
        res = db.get_results()
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(res).tobytes())
        h.update(np.ascontiguousarray(db.get_vars()).tobytes())
        out[name] = {"ms": round(ms, 4), "trials": int(res["trials"].sum()), "digest": h.hexdigest()[:16]}
        db.free()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        print(json.dumps(one(int(sys.argv[2]), int(sys.argv[3]))))
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    res = {}
    for flag in ("0", "1"):
        env = dict(os.environ, FIKSI_AMD_GROUPED_C=flag)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n), str(reps)], env=env, capture_output=True, text=True)
        if r.returncode:
            print(r.stderr[-3000:])
            sys.exit(1)
        res["general" if flag == "0" else "one_structure"] = json.loads(r.stdout.strip().splitlines()[-1])
    res["same_bits"] = all(res["general"][k]["digest"] == res["one_structure"][k]["digest"] for k in res["general"])
    print(json.dumps(res))
