import sys; sys.path.insert(0, '.')
import fiksi_amd
from fiksi_amd import abi, workloads
ctx = fiksi_amd.Context(0)
db = ctx.upload(workloads.ring16(100000))
for solver in (0, 1):
    o = abi.solving_opts(solver=solver)
    db.system_solve(o); ctx.synchronize()
    ctx.timer_begin()
    for _ in range(5): db.system_solve(o)
    print(f"ring16 x100000 solver={solver}: {ctx.timer_end()/5:.3f} ms")
