"""One resident hinged_triangles(1, 64) System (258 variables), solved a few times: the launches of a solve of the multifrontal
build's one-workgroup kernel (rocprofv3 --kernel-trace) and, under FIKSI_AMD_TEAM_PROF=1, its phases.
    python tools/hinged64_once.py [n_systems] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fiksi_amd
from fiksi_amd import abi, workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = fiksi_amd.Context(0)
ctx.set_one_structure_builds(False)
db = ctx.upload(workloads.hinged_triangles(n, 64))
opts = abi.solving_opts()
db.system_solve(opts)
ctx.synchronize()
ctx.timer_begin()
for _ in range(reps):
    db.system_solve(opts)
ms = ctx.timer_end() / reps
r = db.get_results()
print({"systems": n, "ms_per_solve": round(ms, 4), "accepted": int(r["accepted"][0]), "trials": int(r["trials"][0])})
