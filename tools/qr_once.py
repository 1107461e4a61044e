"""Diagnostic driver for rocprofv3: n ring16 sketches solved `reps` times with FX_STEP_QR (the reference's numerics), nothing
else. Third argument: routing (-1 default, 0 = one System per wavefront)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fiksi_amd
from fiksi_amd import abi, workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = fiksi_amd.Context(0)
if len(sys.argv) > 3:
    ctx.set_routing(int(sys.argv[3]))
db = ctx.upload(workloads.ring16(n))
o = abi.solving_opts(solver=2)
for _ in range(reps):
    db.system_solve(o)
ctx.synchronize()
ctx.timer_begin()
for _ in range(reps):
    db.system_solve(o)
print('{"systems": %d, "ms_per_solve": %.4f}' % (n, ctx.timer_end() / reps))
