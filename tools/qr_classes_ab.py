"""FX_STEP_QR on a batch of SEVERAL structures: the grouped QR build per big structure class (launch_class_qr) against the
one-wavefront QR kernel for everybody (FIKSI_AMD_QR_CLASSES=0, a child process each): time per resident solve and a digest of every
variable and result; the one-structure batch of the same size beside it.   python tools/qr_classes_ab.py [n_systems]"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "child" else 100000

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np

    import fiksi_amd
    from fiksi_amd import abi, workloads

    n = int(sys.argv[2])
    ctx = fiksi_amd.Context(0)
    o = abi.solving_opts(solver=2)
    out = {}
    for name, b in (("two_structures", workloads.ring16_two_structures(n)), ("one_structure", workloads.ring16(n)),
                    ("two_structures_and_a_rest", workloads.concat([workloads.ring16_two_structures(n // 2), workloads.hinged_triangles(300, 5),
                                                                    workloads.ring16(n // 4, fix_gauge=True)]))):
        db = ctx.upload(b)
        db.system_solve(o)
        ctx.synchronize()
        ctx.timer_begin()
        for _ in range(3):
            db.system_solve(o)
        ms = ctx.timer_end() / 3
        v, r = db.get_vars(), db.get_results()
        out[name] = {"ms": round(ms, 3), "digest": hashlib.sha256(v.tobytes() + r.tobytes()).hexdigest()[:16], "trials": int(r["trials"].sum())}
        db.free()
    print(json.dumps(out))
    sys.exit(0)

res = {}
for label, env in (("classes", {}), ("one_wavefront_kernel_for_all", {"FIKSI_AMD_QR_CLASSES": "0"})):
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    res[label] = json.loads(lines[-1]) if lines else {"error": (p.stderr or "")[-400:]}
ok = all("error" not in v for v in res.values())
if ok:
    res["same_bits"] = {k: res["classes"][k]["digest"] == res["one_wavefront_kernel_for_all"][k]["digest"] for k in res["classes"]}
    res["two_structures_over_one_structure"] = round(res["classes"]["two_structures"]["ms"] / res["classes"]["one_structure"]["ms"], 3)
print(json.dumps(res))
