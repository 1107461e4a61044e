import os, sys
sys.path.insert(0, "/root/repo")
import fiksi_amd
from fiksi_amd import abi, workloads
ctx = fiksi_amd.Context(0)
ctx.set_one_structure_builds(False)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
db = ctx.upload(workloads.concat([workloads.large_sketch(300, seed=3 + k) for k in range(n)]))
opts = abi.solving_opts()
db.system_solve(opts)
ctx.synchronize()
ctx.timer_begin()
for _ in range(3):
    db.system_solve(opts)
print("ms", ctx.timer_end() / 3)
