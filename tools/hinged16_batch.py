"""20 000 sketches of 16 hinged triangles (66 variables: the wide kernel), resident batch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fiksi_amd
from fiksi_amd import workloads
ctx = fiksi_amd.Context(0)
for n, t in ((20000, 16), (2048, 16), (2048, 20), (2048, 31)):
    db = ctx.upload(workloads.hinged_triangles(n, t))
    db.system_solve(); ctx.synchronize()
    ctx.timer_begin()
    for _ in range(3): db.system_solve()
    print(f"{n} x hinged({t}): {ctx.timer_end() / 3:.3f} ms", flush=True)
    db.free()
